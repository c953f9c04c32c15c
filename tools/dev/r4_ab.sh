# same-box A/B: current library against build_exp/$BASE (default libbase.so) on a list of workloads ("pattern batch soc"), two interleaved repetitions
#   usage: bash tools/dev/r4_ab.sh "dense-front 512 0" "MPC02 1024 1" ...      (env BASE=libhead.so, TESTS="-k expr" to run a test subset first)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
base=${BASE:-libbase.so}
{ if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests -m gpu -x -q $TESTS 2>&1 | tail -5; fi
for rep in 1 2; do
for w in "$@"; do
python tools/dev/r4_phases.py $w | head -1
EICOS_AMD_LIB=$PWD/build_exp/$base python tools/dev/r4_phases.py $w | head -1
done; done
python tools/dev/r4_phases.py $1
} > gpurun_out/r4_ab.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r4_ab.log | cut -c1-250
