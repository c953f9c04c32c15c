#!/bin/bash
# Round 6: same-box, interleaved comparison of the round-4 and round-5 FINAL libraries (build_exp/r04, build_exp/r05: `git worktree` at
# 2992e6f / 87ad960 + make) and of the current product, at the large batches (the 168-VGPR build at three workgroups per CU) and with
# two workgroups per CU forced (256-VGPR build + apex + residual head in LDS).
#   usage: bash tools/dev/r6_ab_libs.sh "MPC02 4096 0" "MPC02 2048 0" ...
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
R=${REPS:-2}
{ for rep in $(seq $R); do for w in "$@"; do
for l in r04 r05; do printf "%-12s " $l; EICOS_AMD_LIB=$PWD/build_exp/$l/libeicos_amd.so python tools/dev/r4_phases.py $w | head -1; done
printf "%-12s " product; python tools/dev/r4_phases.py $w | head -1
if [ -n "$BPC2" ]; then
printf "%-12s " r05-bpc2; EICOS_BLOCKS_PER_CU=2 EICOS_AMD_LIB=$PWD/build_exp/r05/libeicos_amd.so python tools/dev/r4_phases.py $w | head -1
printf "%-12s " product-bpc2; EICOS_BLOCKS_PER_CU=2 python tools/dev/r4_phases.py $w | head -1
fi
done; done; } > gpurun_out/r6_ab_libs.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r6_ab_libs.log | cut -c1-230
