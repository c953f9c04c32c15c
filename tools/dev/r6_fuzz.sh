#!/bin/bash
# Round 6 campaign (fresh seeds) on the round's final library: the U-in-LDS build (small batches run one workgroup per CU: most cases take it),
# half of the cases through the one-call fused updateData + solve (pageable arrays: staged while the kernel runs).
export TMPDIR=/tmp FUZZ_FUSED=1 EICOS_FUSED_STAGED=1
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python tools/dev/fuzz_gpu.py 8000 700000
FUZZ_SCALE=3 timeout 600 python tools/dev/fuzz_gpu.py 2000 720000
FUZZ_DYNREG=1 timeout 300 python tools/dev/fuzz_gpu.py 2000 740000
} > gpurun_out/fuzz_r6.log 2>&1
grep -c "ORDERING-DEPENDENT" gpurun_out/fuzz_r6.log; grep -v "ORDERING-DEPENDENT" gpurun_out/fuzz_r6.log | cut -c1-400 | tail -40
