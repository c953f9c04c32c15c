# A/B of library variants on one box: usage ab_r1.sh <B> lib1 lib2 ... ("" = the in-tree library); two rounds
B=$1; shift
for round in 1 2; do
for lib in "$@"; do
  EICOS_AMD_LIB=$lib timeout 200 python tools/dev/gpu_sweep.py MPC02 $B 3 2>&1 | head -1 | cut -c1-150
done
done
