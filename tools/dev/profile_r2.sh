# round 2: rocprofv3 summaries for profiles/ -- kernel stats of the bench command + PMC passes (separate runs, kernel-trace only)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r2
python3 bench.py --steps 10 --warmup 2 > gpurun_out/prof_r2_bench_plain.json 2> gpurun_out/prof_r2_bench_plain.err
tail -1 gpurun_out/prof_r2_bench_plain.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-soc > gpurun_out/prof_r2_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/soc_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --soc > gpurun_out/prof_r2_bench_soc.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/prof_r2/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-soc > gpurun_out/prof_r2_pmc_$tag.log 2>&1
done
# the SOC leg of the headline line: traffic counters
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/prof_r2/soc_pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --soc > gpurun_out/prof_r2_soc_pmc_$c.log 2>&1
done
# dense-front (tile / MFMA path): kernel stats + MFMA and traffic counters
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/tile_stats -- python3 bench.py --pattern dense-front --batch 512 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r2_tile_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/prof_r2/tile_pmc_$tag -- python3 bench.py --pattern dense-front --batch 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r2_tile_pmc_$tag.log 2>&1
done
find gpurun_out/prof_r2 -name "*.csv" | wc -l
# FETCH_SIZE / WRITE_SIZE calibration incl. the gather case (profiles/r02_fetch_calibration.md)
if [ -x build_exp/calib_fetch ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/prof_r2/calib_$c -- build_exp/calib_fetch > gpurun_out/prof_r2_calib_$c.log 2>&1
  done
fi
