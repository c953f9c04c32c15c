export TMPDIR=/tmp
for w in 4 8 16 32; do
for p in lp_bandm lp_agg lp_agg2 lp_bnl1 lp_25fv47; do
EICOS_HYB_W=$w python tools/dev/gpu_sweep.py $p 256 2 2>&1 | tail -3 | head -1 | sed "s/^/W=$w /"
done; done > gpurun_out/r2_hyb.log 2>&1
cat gpurun_out/r2_hyb.log
