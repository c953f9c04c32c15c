#!/bin/bash
# Round 5 regression run: the whole GPU suite, then the evidence run (tools/dev/profile_r5.sh)
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out; rm -f gpurun_out/parity_xerr.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r5_pytest.log
tail -4 gpurun_out/r5_pytest.log; cat gpurun_out/parity_xerr.jsonl
if [ "$1" = "profile" ]; then bash tools/dev/profile_r5.sh; fi
