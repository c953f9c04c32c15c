#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
python tools/dev/gpu_sweep.py dense-front 512 3
python tools/dev/gpu_sweep.py dense-front 256 3
python tools/dev/gpu_sweep.py lp_25fv47 256 3
python tools/dev/gpu_sweep.py lp_agg2 256 3
FUZZ_DYNREG=1 timeout 900 python tools/dev/fuzz_gpu.py 1000 60000
timeout 900 python tools/dev/fuzz_gpu.py 2000 70000
} > gpurun_out/cases.log 2>&1
cat gpurun_out/cases.log | cut -c1-420
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
