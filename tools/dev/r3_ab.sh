# same-box A/B of library variants: r3_ab.sh <pattern> <batch> libA.so libB.so ...
export TMPDIR=/tmp EICOS_EXPERIMENT=1
pat=$1; B=$2; shift 2
for rep in 1 2; do for lib in "$@"; do EICOS_AMD_LIB=$PWD/$lib python tools/dev/gpu_sweep.py $pat $B 3 2>&1 | cut -c1-360; done; done
