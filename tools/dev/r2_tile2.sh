export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "dense_front or big_cone or ldl_factor or random_socp or config4 or variant or dynamic" 2>&1 | tail -8 > gpurun_out/r2_tile2t.log
cat gpurun_out/r2_tile2t.log
python tools/dev/gpu_sweep.py dense-front 512 3 > gpurun_out/r2_tile2.log 2>&1
python tools/dev/gpu_sweep.py dense-front 256 2 >> gpurun_out/r2_tile2.log 2>&1
for p in lp_agg2 lp_25fv47; do python tools/dev/gpu_sweep.py $p 256 2 2>&1 | head -2; done >> gpurun_out/r2_tile2.log
cat gpurun_out/r2_tile2.log
