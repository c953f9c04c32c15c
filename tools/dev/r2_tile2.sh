export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2_tile2t.log
cat gpurun_out/r2_tile2t.log
python tools/dev/gpu_sweep.py dense-front 512 3 > gpurun_out/r2_tile2.log 2>&1
python tools/dev/gpu_sweep.py dense-front 256 2 >> gpurun_out/r2_tile2.log 2>&1
python tools/dev/gpu_sweep.py MPC02 512 3 >> gpurun_out/r2_tile2.log 2>&1
python tools/dev/gpu_sweep.py MPC02 1024 3 >> gpurun_out/r2_tile2.log 2>&1
cat gpurun_out/r2_tile2.log
