# memory-system counters at one and two workgroups per CU (separate --pmc passes, kernel-trace only)
export TMPDIR=/tmp
for B in ${PMC_B:-512}; do
i=0
for c in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
         "FETCH_SIZE WRITE_SIZE" \
         "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc2/B${B}_$i -- python3 tools/dev/gpu_sweep.py MPC02 $B 1 > gpurun_out/pmc2_B${B}_$i.log 2>&1
  tail -2 gpurun_out/pmc2_B${B}_$i.log | head -1 | cut -c1-150
done
done
