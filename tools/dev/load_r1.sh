# per-instance phase times vs machine load (how much of a level step is loaded memory latency)
for B in 64 256 512; do
  for N in 1 2; do
    EICOS_NLDS=$N EICOS_THREADS=512 python tools/dev/gpu_sweep.py MPC02 $B 2 2>&1 | cut -c1-420
  done
done
EICOS_NLDS=2 EICOS_THREADS=1024 python tools/dev/gpu_sweep.py MPC02 256 2 2>&1 | cut -c1-420
EICOS_NLDS=0 EICOS_THREADS=512 python tools/dev/gpu_sweep.py MPC02 64 2 2>&1 | cut -c1-420
