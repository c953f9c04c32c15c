export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r2_fac_t.log
cat gpurun_out/r2_fac_t.log
python tools/dev/gpu_sweep.py MPC02 1024 3 2>&1 | head -1 > gpurun_out/r2_fac.log
python tools/dev/gpu_sweep.py MPC02 512 3 2>&1 | head -2 >> gpurun_out/r2_fac.log
python tools/dev/gpu_sweep.py MPC02 4096 2 2>&1 | head -1 >> gpurun_out/r2_fac.log
cat gpurun_out/r2_fac.log
