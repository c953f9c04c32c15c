export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/dev/r4_fatal_seeds.py gpurun_out/ordering_fatal_seeds.json > gpurun_out/fatal_regen.log 2>&1; tail -2 gpurun_out/fatal_regen.log | cut -c1-200
bash tools/dev/profile_r4.sh 2>&1 | tail -5
