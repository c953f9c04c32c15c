#!/bin/bash
# Round 5: what does a dual right-hand-side solve buy at EQUAL occupancy?  MPC02 at one workgroup per CU (where two 48 KB vectors fit), dual off / on,
# 256 and 512 threads, with the per-stage timers.
export TMPDIR=/tmp EICOS_EXPERIMENT=1 EICOS_TRI_W=1 EICOS_BLOCKS_PER_CU=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ for T in 256 512; do for B in 256 1024; do for D in 0 1; do
echo "--- T=$T B=$B dual=$D"; EICOS_THREADS=$T EICOS_DUAL=$D python tools/dev/r4_phases.py MPC02 $B 0
done; done; done; } > gpurun_out/r5_dual_probe.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r5_dual_probe.log | cut -c1-260
