#!/bin/bash
# Round 6 regression run: the whole GPU suite (+ measured x parity maxima), then the stage timers of the main workloads.
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out; rm -f gpurun_out/parity_xerr.jsonl
{ timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
cat gpurun_out/parity_xerr.jsonl
if [ -n "$PHASES" ]; then for w in "MPC02 512 0" "MPC02 512 1" "lp_bandm 256 0" "lp_agg 256 0" "lp_adlittle 256 0"; do python tools/dev/r4_phases.py $w; done; fi
} > gpurun_out/r6_suite.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r6_suite.log | cut -c1-400
