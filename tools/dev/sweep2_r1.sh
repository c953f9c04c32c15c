for B in 256 512 768 1024 1536 4096; do
  timeout 200 python tools/dev/gpu_sweep.py MPC02 $B 3 2>&1 | head -1 | cut -c1-170
done
EICOS_THREADS=512 timeout 200 python tools/dev/gpu_sweep.py MPC02 256 3 2>&1 | head -1 | cut -c1-170
EICOS_THREADS=256 timeout 200 python tools/dev/gpu_sweep.py MPC02 256 3 2>&1 | head -1 | cut -c1-170
