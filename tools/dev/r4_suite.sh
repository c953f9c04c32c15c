# round 4: GPU test suite + default bench line (driver-shaped), outputs under gpurun_out/
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4_gputest.log
timeout 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench rc=$?"
wc -c gpurun_out/r4_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'])
for k,v in d['config']['summary'].items(): print(k, v)
PY
