# quick regression + timing after a kernel change
python -m pytest tests -q -m gpu -x 2>&1 | tail -2
EICOS_NLDS=1 EICOS_THREADS=512 python tools/dev/gpu_sweep.py MPC02 64 2 2>&1 | cut -c1-330
python tools/dev/gpu_sweep.py MPC02 768 2 2>&1 | cut -c1-330
python tools/dev/gpu_sweep.py MPC02 1024 3 | head -1
python bench.py --no-cpu-baseline | tail -1 | cut -c1-200
