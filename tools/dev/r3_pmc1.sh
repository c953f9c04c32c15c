# round 3: where does a CU saturate?  instruction mix + unit busy counters of k_solve on the headline workload
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_r3
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" \
         "TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
         "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r3/g$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-soc > gpurun_out/pmc_r3_g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_r3/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
disp = collections.Counter()
for f in glob.glob("gpurun_out/pmc_r3/g1/**/*counter_collection.csv", recursive=True):
    ids = set(r["Dispatch_Id"] for r in csv.DictReader(open(f)) if "k_solve" in r["Kernel_Name"])
    nd = len(ids)
print("dispatches per group:", nd)
for k in sorted(tot): print(f"{k}: {tot[k]/max(1,nd):.5g}")
PY
