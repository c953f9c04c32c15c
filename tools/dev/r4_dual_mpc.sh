# round 4: dual right-hand-side solves on the headline pattern at batches that do NOT fit one workgroup per CU (the default turns them off there:
# two 48 KB vectors per workgroup leave room for one workgroup per CU only)
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for rep in 1 2; do
for B in 1024 512; do
python tools/dev/r4_phases.py MPC02 $B 0 | head -1
EICOS_DUAL=1 python tools/dev/r4_phases.py MPC02 $B 0 | head -1
EICOS_DUAL=1 EICOS_THREADS=512 python tools/dev/r4_phases.py MPC02 $B 0 | head -1
done; done
python tools/dev/r4_phases.py MPC02 256 0
EICOS_DUAL=0 python tools/dev/r4_phases.py MPC02 256 0
} > gpurun_out/r4_dual_mpc.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r4_dual_mpc.log | cut -c1-250
