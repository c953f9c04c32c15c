#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for r in 1 2; do
for p in "dense-front 512" "dense-front 256" "lp_agg2 256" "lp_25fv47 256" "lp_bnl1 256"; do
  for v in "" pf4 pf6; do
    if [ -z "$v" ]; then python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -1 | cut -c1-170; else EICOS_AMD_LIB=$PWD/build_exp/lib$v.so python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -1 | cut -c1-170; fi
  done
done; done
python tools/dev/gpu_sweep.py dense-front 256 3 | sed -n 2p
EICOS_AMD_LIB=$PWD/build_exp/libpf4.so python tools/dev/gpu_sweep.py dense-front 256 3 | sed -n 2p
} > gpurun_out/ab4.log 2>&1
cat gpurun_out/ab4.log | sed 's/resident.*: ms=/ ms=/; s/pcost0.*//'
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4

