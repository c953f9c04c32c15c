#!/bin/bash
# Round 5 campaign (fresh seeds) on the round's final library: host-pointer updateData through the pinned bounce pipeline (every case),
# split tile sweeps (the tile / hybrid cases).  Same classes of cases expected as in profiles/r04_log_fuzz_r4*.log.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python tools/dev/fuzz_gpu.py 8000 620000
FUZZ_SCALE=3 timeout 600 python tools/dev/fuzz_gpu.py 2000 640000
FUZZ_DYNREG=1 timeout 300 python tools/dev/fuzz_gpu.py 2000 660000
} > gpurun_out/fuzz_r5.log 2>&1
grep -c "ORDERING-DEPENDENT" gpurun_out/fuzz_r5.log; grep -v "ORDERING-DEPENDENT" gpurun_out/fuzz_r5.log | cut -c1-400 | tail -40
