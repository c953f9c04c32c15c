export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r2_pytest1.log
cat gpurun_out/r2_pytest1.log
for p in lp_afiro lp_adlittle lp_blend lp_bandm lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_bnl1 lp_25fv47; do
  python bench.py --pattern $p --batch 256 --perturb --steps 3 --warmup 1 2>&1 | tail -1
done > gpurun_out/r2_configs3.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r2_configs3.jsonl'):
    d=json.loads(l); c=d['config']; cb=d['cpu_baseline']
    print(c['workload'].split(',')[1].strip()[:22], 'lev',c['levels'],'nnzL',c['nnzL'],'T',c['threads_per_block'],'opt',c['optimal'],'gpu %.0f cpu %.0f ratio %.2f frac %.3f match %s maxdiff %d' % (d['value'], cb['value'], d['value']/cb['value'], d['roofline']['frac'], cb['iters_match_gpu'], cb['iters_max_abs_diff_vs_gpu']))
PY
