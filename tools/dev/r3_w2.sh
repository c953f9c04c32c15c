# round 3: the two-waves-per-SIMD build of the 256-thread kernel (EICOS_W2) against the default one, same box
export TMPDIR=/tmp EICOS_EXPERIMENT=1
for rep in 1 2; do for e in EICOS_W2=0 EICOS_W2=1; do for p in "MPC02 1024" "MPC02 512" "lp_adlittle 256" "lp_blend 256"; do set -- $p; echo "--- $e $p"; env $e python tools/dev/gpu_sweep.py $1 $2 3 2>&1 | grep -v "^   " | cut -c1-200; done;
  echo "--- $e soc"; env $e python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --soc 2>/dev/null | tail -1 | cut -c1-100; done; done
