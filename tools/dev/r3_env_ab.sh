# same-box A/B of an environment knob: r3_env_ab.sh <pattern> <batch> "ENV=a" "ENV=b" ...
export TMPDIR=/tmp EICOS_EXPERIMENT=1
pat=$1; B=$2; shift 2
for rep in 1 2; do for e in "$@"; do echo "--- $e"; env $e python tools/dev/gpu_sweep.py $pat $B 3 2>&1 | cut -c1-360; done; done
