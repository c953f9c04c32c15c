#!/bin/bash
# Round 5: per-stage timers of the small Netlib patterns at batch 256 (one workgroup per CU)
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ for p in lp_afiro lp_adlittle lp_blend lp_beaconfd lp_bandm; do python tools/dev/r4_phases.py $p 256 0; done
echo "--- afiro, scalar path forced for adlittle / blend"; for p in lp_adlittle lp_blend; do EICOS_TILES=0 python tools/dev/r4_phases.py $p 256 0; done
} > gpurun_out/r5_small_phases.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r5_small_phases.log | cut -c1-260
