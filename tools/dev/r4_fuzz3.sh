#!/bin/bash
# third campaign (fresh seeds) on the library whose stage functions save no callee-saved registers: same classes of cases expected as in r04_log_fuzz_r4b.log
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python tools/dev/fuzz_gpu.py 8000 520000
FUZZ_SCALE=3 timeout 600 python tools/dev/fuzz_gpu.py 2000 540000
FUZZ_DYNREG=1 timeout 300 python tools/dev/fuzz_gpu.py 2000 560000
} > gpurun_out/fuzz_r4c.log 2>&1
grep -c "ORDERING-DEPENDENT" gpurun_out/fuzz_r4c.log; grep -v "ORDERING-DEPENDENT" gpurun_out/fuzz_r4c.log | cut -c1-400
