# how much of FETCH_SIZE is the shared index arrays?  same workload with 16-bit and with 32-bit gather indices
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  rm -rf gpurun_out/pmc_idx$v
  EICOS_IDX16=$v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_idx$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-soc --no-configs > gpurun_out/pmc_idx$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]; tot = 0; ids = set()
for f in glob.glob(f"gpurun_out/pmc_idx{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve" in r["Kernel_Name"]: tot += float(r["Counter_Value"]); ids.add(r["Dispatch_Id"])
print(f"IDX16={v}: FETCH_SIZE per launch {tot/len(ids):.5g} KiB -> read bytes (x2 x1024) {2*1024*tot/len(ids):.4g}")
PY
done
