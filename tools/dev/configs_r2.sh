# round 2: every BASELINE.json config beyond the headline line (one bench line each)
export TMPDIR=/tmp
for p in lp_afiro lp_adlittle lp_blend lp_bandm lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_bnl1 lp_25fv47; do
  python bench.py --pattern $p --batch 256 --perturb --steps 3 --warmup 1 2>&1 | tail -1
done > gpurun_out/configs_r2.jsonl
python bench.py --pattern dense-front --batch 512 --steps 3 --warmup 1 2>&1 | tail -1 >> gpurun_out/configs_r2.jsonl
python bench.py --batch 512 --steps 5 --warmup 1 2>&1 | tail -1 >> gpurun_out/configs_r2.jsonl
python bench.py --batch 4096 --steps 3 --warmup 1 --no-soc 2>&1 | tail -1 >> gpurun_out/configs_r2.jsonl
wc -l gpurun_out/configs_r2.jsonl
