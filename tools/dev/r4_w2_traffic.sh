# round 4: is the extra traffic of the three-per-CU launches a property of the 168-VGPR build or of three workgroups sharing a CU's L2 share?
# the same batch-1024 launch (two per CU) with the 256-VGPR build (default) and with the 168-VGPR build (EICOS_W2=0)
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd $GRAFT_REPO_ROOT
out=gpurun_out/w2_traffic; rm -rf $out; mkdir -p $out
args="--steps 2 --warmup 1 --no-cpu-baseline --no-soc --no-configs"
for w in 1 0; do
  export EICOS_W2=$w
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/w2${w}_$c -- python3 bench.py $args > $out/w2${w}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
for w in (1, 0):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(float); dur = []
        for f in glob.glob(f"gpurun_out/w2_traffic/w2{w}_{c}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_solve" in r["Kernel_Name"] and r["Counter_Name"] == c: vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
        for f in glob.glob(f"gpurun_out/w2_traffic/w2{w}_{c}/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "k_solve" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6); name = r["Kernel_Name"][:40]
        tot[c] = sum(vals.values()) / max(1, len(vals))
    print(f"EICOS_W2={w} ({name}): read {2*tot['FETCH_SIZE']*1024/1e9:.2f} GB write {tot['WRITE_SIZE']*1024/1e9:.2f} GB total {(2*tot['FETCH_SIZE']+tot['WRITE_SIZE'])*1024/1e9:.2f} GB per launch, kernel {sum(dur)/len(dur):.2f} ms")
PY
find $out -name "*agent_info.csv" -delete
