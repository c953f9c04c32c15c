EICOS_THREADS=512 EICOS_NLDS=1 python tests/gpu_sweep.py MPC02 1024 2 2>&1 | tail -1
EICOS_THREADS=256 EICOS_NLDS=1 python tests/gpu_sweep.py MPC02 1024 2 2>&1 | tail -1
EICOS_THREADS=512 EICOS_NLDS=2 python tests/gpu_sweep.py MPC02 1024 2 2>&1 | tail -1
EICOS_THREADS=512 EICOS_NLDS=1 python tests/gpu_sweep.py MPC02 512 2 2>&1 | tail -2
EICOS_THREADS=512 EICOS_NLDS=1 python tests/gpu_sweep.py MPC02 4096 2 2>&1 | tail -1
