# round 4, first GPU contact: stage timers LP vs SOC, dense-front, small patterns (baseline before any change)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python tools/dev/r4_phases.py MPC02 512 0
python tools/dev/r4_phases.py MPC02 512 1
python tools/dev/r4_phases.py MPC02 1024 0
python tools/dev/r4_phases.py MPC02 1024 1
python tools/dev/r4_phases.py dense-front 256 0
python tools/dev/r4_phases.py lp_afiro 256 0
python tools/dev/r4_phases.py lp_bandm 256 0
} > gpurun_out/r4_diag1.log 2>&1
tail -40 gpurun_out/r4_diag1.log
