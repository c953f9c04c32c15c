# round 3: deferred-L on every Netlib pattern (same box), with the plan statistics a rule could use
export TMPDIR=/tmp EICOS_EXPERIMENT=1
python - <<'PY'
import os, eicos_amd
for n in ["MPC02","lp_afiro","lp_adlittle","lp_blend","lp_bandm","lp_beaconfd","lp_agg","lp_agg2","lp_agg3","lp_bnl1","lp_25fv47"]:
    pat,_=eicos_amd.read_problem(f"tests/golden/{n}.epb"); g=eicos_amd.BatchSolver(pat,1); d=g.dims()
    print(n,{k:d[k] for k in ("nnzL","nlevels","factor_pairs","factor_path","threads_per_block")}, "pairs/level/T %.2f" % (d["factor_pairs"]/d["nlevels"]/d["threads_per_block"]), flush=True); g.close()
PY
for p in lp_adlittle lp_blend lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_bnl1 lp_25fv47; do for rep in 1 2; do for e in EICOS_FAC_DEFER=0 EICOS_FAC_DEFER=1; do echo "--- $e $p"; env $e python tools/dev/gpu_sweep.py $p 256 3 2>&1 | cut -c1-330; done; done; done
