#!/bin/bash
# same-box A/B: default library vs build_exp/lib$1.so on the headline, SOC, batch-512 and two Netlib patterns; then the GPU parity suite on the variant
cd "$(dirname "$0")/../.."
V=${1:-fma2}
mkdir -p gpurun_out
{
for r in 1 2; do
for p in "MPC02 1024" "MPC02 512" "MPC02 4096" "lp_bandm 256" "lp_agg2 256" "dense-front 512"; do
  python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -1 | cut -c1-170
  EICOS_AMD_LIB=$PWD/build_exp/lib$V.so python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -1 | cut -c1-170
done
done
} > gpurun_out/ab3.log 2>&1
cat gpurun_out/ab3.log | sed 's/resident.*: ms=/ ms=/; s/pcost0.*//'
EICOS_AMD_LIB=$PWD/build_exp/lib$V.so timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -8
