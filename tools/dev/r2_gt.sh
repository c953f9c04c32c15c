#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for r in 1 2; do
EICOS_GTILES=0 python tools/dev/gpu_sweep.py dense-front 512 3
python tools/dev/gpu_sweep.py dense-front 512 3
done
EICOS_GTILES=0 python tools/dev/gpu_sweep.py dense-front 256 3
python tools/dev/gpu_sweep.py dense-front 256 3
} > gpurun_out/gt.log 2>&1
grep -v "factor us per call" gpurun_out/gt.log | cut -c1-330
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6
