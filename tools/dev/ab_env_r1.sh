# A/B of an environment switch on one box: usage ab_env_r1.sh <B> VAR=a VAR=b ...
B=$1; shift
for round in 1 2; do
for kv in "$@"; do
  env $kv timeout 200 python tools/dev/gpu_sweep.py MPC02 $B 3 2>&1 | head -1 | cut -c1-150 | sed "s/^/[$kv] /"
done
done
