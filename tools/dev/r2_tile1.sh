export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "dense_front or big_cone or ldl_factor or random_socp" 2>&1 | tail -15 > gpurun_out/r2_tile1.log
cat gpurun_out/r2_tile1.log
python -m pytest tests -m gpu -q -k "config4" 2>&1 | tail -15 > gpurun_out/r2_tile1b.log
cat gpurun_out/r2_tile1b.log
python bench.py --pattern dense-front --batch 512 --steps 2 --warmup 1 2>&1 | tail -1 > gpurun_out/r2_tile1_bench.json
cat gpurun_out/r2_tile1_bench.json
