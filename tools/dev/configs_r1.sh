# BASELINE.json configs beyond the headline one: LPnetlib batch 256 (perturbed), dense-front batch 512, MPC batch 4096
for p in lp_afiro lp_adlittle lp_blend lp_bandm lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_bnl1 lp_25fv47; do
  python bench.py --pattern $p --batch 256 --perturb --steps 3 --warmup 1 2>&1 | tail -1
done
python bench.py --pattern dense-front --batch 512 --steps 2 --warmup 1 2>&1 | tail -1
python bench.py --batch 4096 --steps 3 --warmup 1 2>&1 | tail -1
