export TMPDIR=/tmp
for round in 1 2; do
python tools/dev/gpu_sweep.py MPC02 512 4 2>&1 | head -1 | cut -c1-200
EICOS_ORDER_ALWAYS=1 python tools/dev/gpu_sweep.py MPC02 512 4 2>&1 | head -1 | cut -c1-200
done > gpurun_out/r2_ab.log 2>&1
cat gpurun_out/r2_ab.log
