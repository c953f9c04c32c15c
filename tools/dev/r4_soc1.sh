export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -x -q -k "multi_gpu or device_list or soc or random_socp or big_cone or fatal_seeds or dense_front or every_kernel" 2>&1 | tail -15
python tools/dev/r4_phases.py MPC02 512 1; python tools/dev/r4_phases.py MPC02 1024 1; python tools/dev/r4_phases.py MPC02 1024 0; python tools/dev/hash_outputs.py gpurun_out/r4_hash_tiny.json > /dev/null 2>&1; echo hash done; } > gpurun_out/r4_soc1.log 2>&1
cat gpurun_out/r4_soc1.log | cut -c1-300
