#!/bin/bash
# Round 6, first GPU call on the new bench / ring / pinned-extent code: the new tests, the default bench line (timed), launch shapes.
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -x -q -k "ring_of_events or partly_pinned or contract_json or host_pointer" 2>&1 | tail -8
time python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
} > gpurun_out/r6_first.log 2>&1
cat gpurun_out/r6_first.log | cut -c1-250
