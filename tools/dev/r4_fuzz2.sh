#!/bin/bash
# second, larger campaign on the final round-4 kernels (fresh seeds)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 1200 python tools/dev/fuzz_gpu.py 12000 420000
FUZZ_SCALE=3 timeout 900 python tools/dev/fuzz_gpu.py 3000 440000
FUZZ_SCALE=6 timeout 700 python tools/dev/fuzz_gpu.py 600 450000
FUZZ_DYNREG=1 timeout 400 python tools/dev/fuzz_gpu.py 2400 460000
} > gpurun_out/fuzz_r4b.log 2>&1
grep -c "ORDERING-DEPENDENT" gpurun_out/fuzz_r4b.log; grep -v "ORDERING-DEPENDENT" gpurun_out/fuzz_r4b.log | cut -c1-400
