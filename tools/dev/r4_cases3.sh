# the class-(b) cases of the third round-4 campaign (profiles/r04_log_fuzz_r4c.log) in detail (GPU variants vs oracle)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ python tools/dev/r2_case.py 541473 3 0 EICOS_THREADS=128 EICOS_IDX16=0 EICOS_DUAL=0 EICOS_CONE_ORDER=1
python tools/dev/r2_case.py 541692 3 0 EICOS_TILES=1 EICOS_DUAL=0
python tools/dev/r2_case.py 541815 3 0 EICOS_THREADS=256 EICOS_NLDS=1 EICOS_IDX16=0 EICOS_TILES=0 EICOS_DUAL=0 EICOS_W2=0 EICOS_CONE_ORDER=0
python tools/dev/r2_case.py 541900 3 0 EICOS_THREADS=128 EICOS_NLDS=0 EICOS_FAC_DEFER=1
} > gpurun_out/r4_cases3.log 2>&1
cut -c1-420 gpurun_out/r4_cases3.log
