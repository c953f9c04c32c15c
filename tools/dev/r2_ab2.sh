#!/bin/bash
# same-box A/B: build_exp/libold.so (previous commit) vs the current library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for r in 1 2; do
for p in "dense-front 512" "dense-front 256" "lp_agg2 256" "lp_25fv47 256" "lp_bandm 256" "lp_agg 256"; do
  EICOS_AMD_LIB=$PWD/build_exp/libold.so python tools/dev/gpu_sweep.py $p 4 | head -1 | cut -c1-200
  python tools/dev/gpu_sweep.py $p 4 | head -1 | cut -c1-200
done
done
} > gpurun_out/ab2.log 2>&1
cat gpurun_out/ab2.log
