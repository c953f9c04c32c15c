#!/bin/bash
# randomized differential campaign on the round-3 kernels (level-0 fast path, trip loops, cone-aware ordering, every older variant)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 900 python tools/dev/fuzz_gpu.py ${N1:-6000} 120000
FUZZ_SCALE=3 timeout 600 python tools/dev/fuzz_gpu.py ${N2:-1500} 140000
FUZZ_SCALE=6 timeout 400 python tools/dev/fuzz_gpu.py ${N3:-300} 150000
FUZZ_DYNREG=1 timeout 400 python tools/dev/fuzz_gpu.py ${N4:-1200} 160000
} > gpurun_out/fuzz_r3.log 2>&1
tail -40 gpurun_out/fuzz_r3.log
