export TMPDIR=/tmp
for p in lp_afiro lp_adlittle lp_blend lp_bandm lp_beaconfd lp_agg; do
python tools/dev/gpu_sweep.py $p 256 2 2>&1 | tail -3
done > gpurun_out/r2_small.log 2>&1
cat gpurun_out/r2_small.log
