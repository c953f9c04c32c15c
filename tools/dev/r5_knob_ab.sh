#!/bin/bash
# Round 5: same-box A/B of ONE library under two settings of an experiment knob, interleaved, on a list of workloads ("pattern batch soc").
#   usage: KNOB=EICOS_TRI_W A=1 B=2 bash tools/dev/r5_knob_ab.sh "MPC02 1024 0" "MPC02 512 0" ...     (TESTK="expr": the tests selected by `-k expr` first)
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
out=gpurun_out/r5_knob_ab_${KNOB}.log
{ if [ -n "$TESTK" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$TESTK" 2>&1 | tail -5; fi
for rep in 1 2; do
for w in "$@"; do
echo -n "$KNOB=$A  "; env $KNOB=$A python tools/dev/r4_phases.py $w | head -1
echo -n "$KNOB=$B  "; env $KNOB=$B python tools/dev/r4_phases.py $w | head -1
done; done
echo "--- phases, $KNOB=$A then $B, first workload"
env $KNOB=$A python tools/dev/r4_phases.py $1
env $KNOB=$B python tools/dev/r4_phases.py $1
} > $out 2>&1
grep -v "Exception ignored\|BrokenPipe" $out | cut -c1-260
