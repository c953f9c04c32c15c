#!/bin/bash
# small Netlib patterns (config 3 shape: batch 256): factor paths x LDS residency
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
for p in lp_afiro lp_adlittle lp_blend; do
  for B in 256; do
    python tools/dev/gpu_sweep.py $p $B 3
    EICOS_TILES=0 python tools/dev/gpu_sweep.py $p $B 3
    # (an EICOS_THREADS=64 line stood here: the library never had a 64-thread kernel, the run re-measured the default)
    EICOS_TILES=1 python tools/dev/gpu_sweep.py $p $B 3
    EICOS_TILES=1 EICOS_THREADS=256 python tools/dev/gpu_sweep.py $p $B 3
  done
done
} > gpurun_out/small.log 2>&1
grep -v "factor us per call" gpurun_out/small.log | tail -60
