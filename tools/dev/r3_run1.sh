export TMPDIR=/tmp
python tools/dev/r3_parity.py > gpurun_out/r3_parity1.log 2>&1
python tools/dev/gpu_sweep.py MPC02 512 3 > gpurun_out/r3_base_sweep.log 2>&1
python tools/dev/gpu_sweep.py MPC02 1024 3 >> gpurun_out/r3_base_sweep.log 2>&1
python tools/dev/gpu_sweep.py MPC02 256 3 >> gpurun_out/r3_base_sweep.log 2>&1
tail -5 gpurun_out/r3_parity1.log; cat gpurun_out/r3_base_sweep.log
