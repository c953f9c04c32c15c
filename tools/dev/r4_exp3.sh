export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ for rep in 1 2; do
python tools/dev/r4_phases.py MPC02 1024 1 | head -1
EICOS_AMD_LIB=$PWD/build_exp/libbase.so python tools/dev/r4_phases.py MPC02 1024 1 | head -1
done
python tools/dev/r4_tile256.py 1500 24 64 512
python tools/dev/r4_tile256.py 1000 16 64 512
python tools/dev/r4_tile256.py 2000 32 64 512
} > gpurun_out/r4_exp3.log 2>&1
cat gpurun_out/r4_exp3.log | cut -c1-300
