#!/bin/bash
# Round 5: cheap launch-shape probes on the headline batch
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ for rep in 1 2; do
echo -n "default        "; python tools/dev/r4_phases.py MPC02 1024 0 | head -1
echo -n "3 per CU       "; EICOS_BLOCKS_PER_CU=3 python tools/dev/r4_phases.py MPC02 1024 0 | head -1
echo -n "2 per CU, 168  "; EICOS_W2=0 python tools/dev/r4_phases.py MPC02 1024 0 | head -1
echo -n "b1536 default  "; python tools/dev/r4_phases.py MPC02 1536 0 | head -1
echo -n "b768 3 per CU  "; EICOS_BLOCKS_PER_CU=3 python tools/dev/r4_phases.py MPC02 768 0 | head -1
echo -n "b768 default   "; python tools/dev/r4_phases.py MPC02 768 0 | head -1
done; } > gpurun_out/r5_knobs.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r5_knobs.log | cut -c1-200
