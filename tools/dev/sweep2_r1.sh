for cfg in "512 1 1024" "256 1 1024" "256 1 1536" "256 1 3072" "256 0 2048" "128 1 1536" "128 0 2048" "512 0 2048"; do
  set -- $cfg
  EICOS_THREADS=$1 EICOS_NLDS=$2 timeout 120 python tools/dev/gpu_sweep.py MPC02 $3 2 2>&1 | head -1 | cut -c1-170
done
