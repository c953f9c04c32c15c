# round 6: one bench line per LPnetlib pattern of BASELINE configs[3] (all ten, batch 256, perturbed c / h) on the final library
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for p in lp_afiro lp_adlittle lp_blend lp_bandm lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_bnl1 lp_25fv47; do
  python bench.py --pattern $p --batch 256 --perturb --steps 20 --warmup 3 2>/dev/null | tail -1
done > gpurun_out/r6_configs.jsonl
wc -l gpurun_out/r6_configs.jsonl
