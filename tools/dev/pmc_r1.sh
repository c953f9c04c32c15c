export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r1/$tag -- python3 tools/dev/gpu_sweep.py MPC02 1024 1 > gpurun_out/pmc_r1_$tag.log 2>&1
  tail -1 gpurun_out/pmc_r1_$tag.log
done
find gpurun_out/pmc_r1 -name "*counter_collection.csv" | head
