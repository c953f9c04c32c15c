# round 6: rocprofv3 summaries for profiles/ -- kernel stats of the bench command + PMC passes (separate runs, kernel-trace only)
# for EVERY workload of the default bench line (headline, its SOC leg, and the configs).  Every pass is bounded.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r6
rm -rf $out; mkdir -p $out
python3 bench.py --steps 10 --warmup 2 > $out/bench_plain.json 2> $out/bench_plain.err
tail -1 $out/bench_plain.json | cut -c1-200
# rehearsal of the N > 1 measurement path on this one GPU: eight shards of 512 through eicos_multi_* 
python3 bench.py --multi 0,0,0,0,0,0,0,0 --total 4096 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_multi8.json 2> $out/bench_multi8.err
run() { # prefix, full-counters?, bench args...
  pre=$1; full=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${pre}stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-soc --no-configs "$@" > $out/${pre}stats.log 2>&1
  groups=("FETCH_SIZE" "WRITE_SIZE")
  if [ "$full" = 1 ]; then
    groups+=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
  fi
  for c in "${groups[@]}"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-24)
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${pre}pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-soc --no-configs "$@" > $out/${pre}pmc_$tag.log 2>&1
    echo "$pre $tag rc=$?"
  done
}
run ""       1
run soc_     1 --soc
run tile_    1 --pattern dense-front --batch 512
run b512_    0 --batch 512
run b4096_   0 --batch 4096
run afiro_   0 --pattern lp_afiro --batch 256 --perturb
run bandm_   0 --pattern lp_bandm --batch 256 --perturb
run fv47_    0 --pattern lp_25fv47 --batch 256 --perturb
find $out -name "*.csv" | wc -l
# keep the merge small: only the summaries' inputs travel back
find $out -name "*agent_info.csv" -delete
du -sh $out
