python tests/gpu_sweep.py MPC02 1024 2 2>&1 | tail -1
python tests/gpu_sweep.py MPC02 512 2 2>&1 | tail -1
python tests/gpu_sweep.py MPC02 1 2 2>&1 | tail -1
EICOS_THREADS=1024 python tests/gpu_sweep.py lp_25fv47 256 1 2>&1 | tail -2
