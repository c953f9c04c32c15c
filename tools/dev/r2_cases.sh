#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
python tools/dev/r2_case.py 80945 1 0 | grep -v "inst [012] |Ax" | head -2
python tools/dev/r2_case.py 81421 1 0 | grep -v "inst [012] |Ax" | head -2
python tools/dev/r2_case.py 23028 1 0 EICOS_THREADS=512 EICOS_IDX16=0 | grep -v "inst [012] |Ax" | head -2
for r in 1 2; do
for p in "dense-front 512" "lp_agg2 256" "lp_25fv47 256" "lp_bandm 256"; do
  EICOS_AMD_LIB=$PWD/build_exp/libold.so python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -1 | cut -c1-170
  python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -1 | cut -c1-170
done; done
timeout 900 python tools/dev/fuzz_gpu.py 3000 80000 2>&1 | tail -8
} > gpurun_out/cases.log 2>&1
cat gpurun_out/cases.log | cut -c1-420 | sed 's/resident.*: ms=/ ms=/; s/pcost0.*//'
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
