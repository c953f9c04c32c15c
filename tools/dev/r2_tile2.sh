export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "dense_front or big_cone or ldl_factor or random_socp or config4" 2>&1 | tail -8 > gpurun_out/r2_tile2t.log
cat gpurun_out/r2_tile2t.log
python tools/dev/gpu_sweep.py dense-front 64 2 > gpurun_out/r2_tile2.log 2>&1
python tools/dev/gpu_sweep.py dense-front 512 2 >> gpurun_out/r2_tile2.log 2>&1
cat gpurun_out/r2_tile2.log
