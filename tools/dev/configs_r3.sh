# round 3: one bench line per BASELINE.json config beyond the default line's own `configs` object (all ten LPnetlib patterns,
# MPC02 at batch 4096 / 768); run AFTER profiles/r03_*_pmc.json exist so that roofline.traffic is filled where a summary matches
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for p in lp_afiro lp_adlittle lp_blend lp_bandm lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_bnl1 lp_25fv47; do
  python bench.py --pattern $p --batch 256 --perturb --steps 3 --warmup 1 2>&1 | tail -1
done > gpurun_out/configs_r3.jsonl
python bench.py --pattern dense-front --batch 512 --steps 3 --warmup 1 2>&1 | tail -1 >> gpurun_out/configs_r3.jsonl
python bench.py --batch 512 --steps 5 --warmup 1 --no-soc --no-configs 2>&1 | tail -1 >> gpurun_out/configs_r3.jsonl
python bench.py --batch 4096 --steps 3 --warmup 1 --no-soc --no-configs 2>&1 | tail -1 >> gpurun_out/configs_r3.jsonl
python bench.py --steps 10 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_r3.json
wc -l gpurun_out/configs_r3.jsonl; cut -c1-150 gpurun_out/bench_r3.json
