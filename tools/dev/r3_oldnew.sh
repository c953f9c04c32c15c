# round 3: the library at the r03_v2 profile state (build_exp/libold.so, commit d828a7e) against the current one, one box, every workload of the bench line
export TMPDIR=/tmp EICOS_EXPERIMENT=1
for rep in 1 2; do for lib in build_exp/libold.so build_exp/libnew.so; do
  export EICOS_AMD_LIB=$PWD/$lib
  echo "=== $lib"
  for p in "MPC02 1024" "MPC02 512" "lp_afiro 256" "lp_bandm 256" "lp_25fv47 256"; do set -- $p; python tools/dev/gpu_sweep.py $1 $2 3 2>&1 | grep -v "^   " | cut -c1-200; done
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-soc --no-configs --pattern dense-front --batch 512 2>/dev/null | tail -1 | cut -c1-120
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --soc 2>/dev/null | tail -1 | cut -c1-120
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-soc --batch 4096 2>/dev/null | tail -1 | cut -c1-120
done; done
