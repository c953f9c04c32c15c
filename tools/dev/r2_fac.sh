export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r2_fac_t.log
cat gpurun_out/r2_fac_t.log
python tools/dev/gpu_sweep.py MPC02 1024 3 2>&1 | head -1 > gpurun_out/r2_fac.log
python tools/dev/gpu_sweep.py MPC02 512 3 2>&1 | head -3 >> gpurun_out/r2_fac.log
python tools/dev/gpu_sweep.py MPC02 256 2 2>&1 | head -3 >> gpurun_out/r2_fac.log
for p in lp_afiro lp_bandm lp_agg2 lp_25fv47; do python tools/dev/gpu_sweep.py $p 256 2 2>&1 | head -2; done >> gpurun_out/r2_fac.log
cat gpurun_out/r2_fac.log
