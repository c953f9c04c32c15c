# round 4: HBM-side traffic of ONE stage under the real mixed load: a variant library that runs the (idempotent) stage R times per pass against
# the product library, FETCH_SIZE / WRITE_SIZE of k_solve per launch; the difference / (R - 1) / passes = the stage's own traffic
#   usage: bash tools/dev/r4_stage_traffic.sh libfacrep5.so 5 [bench args]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
var=$1; R=$2; shift 2
out=gpurun_out/stage_traffic; rm -rf $out; mkdir -p $out
args="--steps 2 --warmup 1 --no-cpu-baseline --no-soc --no-configs $*"
for lib in base $var; do
  if [ $lib = base ]; then unset EICOS_AMD_LIB; else export EICOS_AMD_LIB=$PWD/build_exp/$lib; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${lib}_$c -- python3 bench.py $args > $out/${lib}_$c.log 2>&1
    echo "$lib $c rc=$?"
  done
done
python3 - "$var" "$R" <<'PY'
import csv, glob, sys, collections
var, R = sys.argv[1], int(sys.argv[2])
res = {}
for lib in ("base", var):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = collections.defaultdict(float); dur = []
        for f in glob.glob(f"gpurun_out/stage_traffic/{lib}_{c}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_solve" in r["Kernel_Name"] and r["Counter_Name"] == c: vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
        for f in glob.glob(f"gpurun_out/stage_traffic/{lib}_{c}/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "k_solve" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        res[(lib, c)] = (sum(vals.values()) / max(1, len(vals)), sum(dur) / max(1, len(dur)), len(vals))
for lib in ("base", var):
    rd = 2 * res[(lib, "FETCH_SIZE")][0] * 1024; wr = res[(lib, "WRITE_SIZE")][0] * 1024
    print(f"{lib:16s} read {rd/1e9:8.2f} GB  write {wr/1e9:8.2f} GB  total {(rd+wr)/1e9:8.2f} GB per launch; kernel {res[(lib,'FETCH_SIZE')][1]:.2f} ms ({res[(lib,'FETCH_SIZE')][2]} launches)")
b = 2 * res[("base", "FETCH_SIZE")][0] * 1024 + res[("base", "WRITE_SIZE")][0] * 1024
v = 2 * res[(var, "FETCH_SIZE")][0] * 1024 + res[(var, "WRITE_SIZE")][0] * 1024
print(f"stage traffic = ({v/1e9:.2f} - {b/1e9:.2f}) / {R-1} = {(v-b)/(R-1)/1e9:.2f} GB per launch = {100*(v-b)/(R-1)/b:.1f} % of the launch; time: {(res[(var,'FETCH_SIZE')][1]-res[('base','FETCH_SIZE')][1])/(R-1):.2f} ms per launch")
PY
find $out -name "*agent_info.csv" -delete
