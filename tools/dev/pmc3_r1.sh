# instruction-cache counters (separate --pmc passes, kernel-trace only)
export TMPDIR=/tmp
for B in 256 512; do
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc3/B${B}_$i -- python3 tools/dev/gpu_sweep.py MPC02 $B 1 > gpurun_out/pmc3_B${B}_$i.log 2>&1
  grep -h "iter/s" gpurun_out/pmc3_B${B}_$i.log | cut -c1-120
done
done
