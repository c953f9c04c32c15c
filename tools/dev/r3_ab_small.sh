export TMPDIR=/tmp EICOS_EXPERIMENT=1
for p in lp_afiro lp_adlittle lp_bandm lp_agg lp_25fv47; do for lib in "$@"; do EICOS_AMD_LIB=$PWD/$lib python tools/dev/gpu_sweep.py $p 256 3 2>&1 | head -1 | cut -c1-200; done; done
for lib in "$@"; do EICOS_AMD_LIB=$PWD/$lib python tools/dev/gpu_sweep.py MPC02 1024 3 2>&1 | head -1 | cut -c1-200; done
