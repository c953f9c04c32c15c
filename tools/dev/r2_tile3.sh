export TMPDIR=/tmp
python bench.py --pattern dense-front --batch 512 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['roofline']['kernel_ms'], d['config']['update_kernel_ms'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x/tile_stats -- python3 bench.py --pattern dense-front --batch 512 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rocprof', d['value'], d['roofline']['kernel_ms'])"
head -3 gpurun_out/prof_x/tile_stats/*/*kernel_stats.csv | cut -c1-200
python bench.py --pattern dense-front --batch 512 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['roofline']['kernel_ms'])"
