#!/bin/bash
# Round 6: the U-in-LDS build (kernels_ubl*.hip) -- tests, then same-box A/B against EICOS_UBL=0 on the Netlib patterns at batch 256
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ timeout 1200 python -m pytest tests -m gpu -x -q -k "u_in_lds or kernel_build or every_kernel_variant or fixture_matches" 2>&1 | tail -8
for rep in 1 2; do for w in ${PATTERNS:-lp_bandm lp_agg lp_adlittle lp_blend lp_beaconfd lp_agg2 lp_agg3 lp_bnl1}; do
printf "UBL=0 "; EICOS_UBL=0 python tools/dev/r4_phases.py $w 256 0 | head -${LINES:-1}
printf "UBL=1 "; EICOS_UBL=1 python tools/dev/r4_phases.py $w 256 0 | head -${LINES:-1}
done; done; } > gpurun_out/r6_ubl.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r6_ubl.log | cut -c1-260
