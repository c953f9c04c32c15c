for lib in "" build_exp/libexp_nogather.so; do
  for B in 64 512; do
    EICOS_AMD_LIB=$lib EICOS_NLDS=1 EICOS_THREADS=512 timeout 120 python tools/dev/gpu_sweep.py MPC02 $B 1 2>&1 | cut -c1-300
  done
done
