# round 3: HBM-side traffic of k_solve on a workload: r3_pmc2.sh <tag> <bench args...>
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/pmc_$tag
# (a TCC_EA0_* group made rocprofv3 abort and hang until its timeout on this pool: not collected)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$tag/$t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-soc --no-configs "$@" > gpurun_out/pmc_${tag}_$t.log 2>&1
  echo "$t rc=$?"
done
python3 - $tag <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
tot = collections.defaultdict(float); nd = collections.defaultdict(set)
for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot): print(f"{k}: {tot[k]/max(1,len(nd[k])):.6g}  (launches {len(nd[k])})")
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    rd = 2 * tot["FETCH_SIZE"] / len(nd["FETCH_SIZE"]) * 1024; wr = tot["WRITE_SIZE"] / len(nd["WRITE_SIZE"]) * 1024
    print(f"traffic per launch: read {rd:.4g} + write {wr:.4g} = {rd+wr:.4g} bytes")
PY
