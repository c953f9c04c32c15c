#!/bin/bash
# Round 5: the compact dual solve -- bit-identity test, then same-box A/B against the single right-hand-side path
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -x -q -k "compact_dual" 2>&1 | tail -15
for rep in 1 2; do
for w in "MPC02 1024 0" "MPC02 512 0" "MPC02 4096 0" "MPC02 256 0"; do
for D in 0 2; do echo -n "EICOS_DUAL=$D  "; EICOS_DUAL=$D timeout 300 python tools/dev/r4_phases.py $w | head -1; done
done; done
for D in 0 2; do echo "--- phases EICOS_DUAL=$D"; EICOS_DUAL=$D timeout 300 python tools/dev/r4_phases.py MPC02 512 0; done
} > gpurun_out/r5_compact.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r5_compact.log | cut -c1-300
