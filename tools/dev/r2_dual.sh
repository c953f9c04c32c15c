#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for p in "lp_bandm 256" "lp_agg 256" "lp_agg2 256" "lp_25fv47 256" "lp_bnl1 256" "lp_beaconfd 256" "lp_blend 256" "MPC02 256" "MPC02 1"; do
  for d in 0 1; do EICOS_DUAL=$d python tools/dev/gpu_sweep.py $p 4 2>/dev/null | head -2 | cut -c1-260; done
done
} > gpurun_out/dual.log 2>&1
cat gpurun_out/dual.log | sed 's/resident.*: ms=/ ms=/; s/pcost0.*//'
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
