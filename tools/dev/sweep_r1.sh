python tools/dev/gpu_sweep.py MPC02 1024 2 2>&1 | head -1 | cut -c1-200
python tools/dev/gpu_sweep.py MPC02 4096 2 2>&1 | head -1 | cut -c1-200
python tools/dev/gpu_sweep.py MPC02 512 2 2>&1 | tail -1 | cut -c1-330
