# round 4: the multi-GPU layer on the one leased GPU: tests, the C++ demo, bench --multi 0,0 against the plain run
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -m gpu -x -q -k "multi_gpu or device_list or bench_emits" 2>&1 | tail -5
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-soc --no-configs | cut -c1-400
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --multi 0,0 | cut -c1-700
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --multi 0 | cut -c1-400
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --multi 0,0 --total 2048 | cut -c1-400
} > gpurun_out/r4_multi.log 2>&1
cat gpurun_out/r4_multi.log
