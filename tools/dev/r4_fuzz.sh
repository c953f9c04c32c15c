#!/bin/bash
# randomised differential campaign on the round-4 kernels (tiny / wave cone paths, amalgamated tile partitions, every older variant); fresh seeds
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python tools/dev/fuzz_gpu.py ${N1:-6000} 320000
FUZZ_SCALE=3 timeout 700 python tools/dev/fuzz_gpu.py ${N2:-1500} 340000
FUZZ_SCALE=6 timeout 500 python tools/dev/fuzz_gpu.py ${N3:-300} 350000
FUZZ_DYNREG=1 timeout 400 python tools/dev/fuzz_gpu.py ${N4:-1200} 360000
} > gpurun_out/fuzz_r4.log 2>&1
tail -60 gpurun_out/fuzz_r4.log | cut -c1-400
