# round 4: what bounds the tiny-pattern launches (lp_afiro, batch 256: one 128-thread workgroup per CU, LDS-resident slabs)?
# instruction fetch (kkt_solve alone is 140 KB of code against a 64 KB instruction cache) or data loads of the shared plan arrays (L2)?
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/afiro_pmc; rm -rf $out; mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1
grep -o -i -E "\b(SQC?_[A-Z_0-9]*(ICACHE|IFETCH|INST_LEVEL|WAIT_INST|DCACHE)[A-Z_0-9]*)" $out/counters.txt | sort -u | tr '\n' ' '; echo
args="--steps 2 --warmup 1 --no-cpu-baseline --no-soc --no-configs --pattern lp_afiro --batch 256 --perturb"
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
         "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -- python3 bench.py $args > $out/p$i.log 2>&1
  echo "pass $i rc=$? ($c)"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("gpurun_out/afiro_pmc/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_solve" not in r["Kernel_Name"]: continue
        agg["k_solve"][r["Counter_Name"]] += float(r["Counter_Value"]); cnt["k_solve"][r["Counter_Name"]] += 1
for k, v in agg.items():
    for c in sorted(v): print("%-32s %.5g per launch (%d launches)" % (c, v[c] / cnt[k][c], cnt[k][c]))
PY
find $out -name "*agent_info.csv" -delete
