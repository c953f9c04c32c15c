#!/bin/bash
# Round 5: same-box comparison of library variants (build_exp/lib<tag>.so, tools/dev/build_variant.sh) against the product, interleaved
#   usage: LIBS="maxilp bias0" bash tools/dev/r5_libs.sh "MPC02 1024 0" "MPC02 512 0" ...
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ for rep in 1 2; do for w in "$@"; do
echo -n "product   "; python tools/dev/r4_phases.py $w | head -1
for l in $LIBS; do printf "%-9s " $l; EICOS_AMD_LIB=$PWD/build_exp/lib$l.so python tools/dev/r4_phases.py $w | head -1; done
done; done; } > gpurun_out/r5_libs.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r5_libs.log | cut -c1-230
