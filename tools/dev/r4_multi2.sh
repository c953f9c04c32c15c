export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ python tools/dev/r4_multi_dbg.py; timeout 600 python -m pytest tests -m gpu -x -q -k "multi_gpu or device_list" 2>&1 | tail -40; python tools/dev/r4_fatal_seeds.py gpurun_out/ordering_fatal_seeds.json | tail -3; 
python tools/dev/r4_phases.py dense-front 256 0; python tools/dev/r4_phases.py dense-front 512 0; } > gpurun_out/r4_multi2.log 2>&1
cat gpurun_out/r4_multi2.log | cut -c1-400
