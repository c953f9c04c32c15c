# round 4: LDS bank-conflict share of the triangular sweeps ALONE (tools/dev/sweep_bench: the library's tri_sweep on the real MPC02 plan, one
# factor copy per workgroup, 512 workgroups = two per CU), next to the whole k_solve launch's counters in profiles/r04_v3_pmc.md
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep_pmc
build_exp/sweep_bench tests/golden/MPC02.epb 512 200 256 2>&1 | tail -6
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/sweep_pmc -- build_exp/sweep_bench tests/golden/MPC02.epb 512 200 256 > gpurun_out/sweep_pmc/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/sweep_pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for c in sorted(v): print("   ", c, "%.4g" % v[c])
    if v.get("SQ_LDS_IDX_ACTIVE", 0) > 0: print("    bank conflict / LDS active = %.3f ; LDS active per CU / busy cycles per SE = %.3f ; wait fraction %.3f" % (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], v["SQ_LDS_IDX_ACTIVE"] / 256 / (v["SQ_BUSY_CYCLES"] / 32), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
find gpurun_out/sweep_pmc -name "*agent_info.csv" -delete
