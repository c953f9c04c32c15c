# rocprofv3 summaries for profiles/: kernel stats of the bench command + HBM traffic PMC passes (separate runs)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1_bench.log 2>&1
tail -1 gpurun_out/prof_r1_bench.log
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/prof_r1/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r1_pmc_$tag.log 2>&1
done
find gpurun_out/prof_r1 -name "*.csv" | head -20
