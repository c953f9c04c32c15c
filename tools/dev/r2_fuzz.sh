#!/bin/bash
# randomized differential campaign on the round-2 paths (tiles / hybrid / LDS-resident / dual right-hand sides)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
timeout 1500 python tools/dev/fuzz_gpu.py ${N1:-10000} 20000
FUZZ_SCALE=3 timeout 1500 python tools/dev/fuzz_gpu.py ${N2:-2500} 40000
FUZZ_SCALE=6 timeout 1500 python tools/dev/fuzz_gpu.py ${N3:-400} 50000
FUZZ_DYNREG=1 timeout 900 python tools/dev/fuzz_gpu.py ${N4:-2000} 60000
} > gpurun_out/fuzz_r2.log 2>&1
tail -60 gpurun_out/fuzz_r2.log
