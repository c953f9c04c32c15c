# the class-(b) cases of the round-4 campaign in detail (GPU variants vs oracle)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ python tools/dev/r2_case.py 350007 6 0 EICOS_NLDS=2 EICOS_TILES=2 EICOS_CONE_ORDER=0
python tools/dev/r2_case.py 350271 6 0 EICOS_DUAL=0
python tools/dev/r2_case.py 350256 6 0 EICOS_THREADS=512 EICOS_NLDS=1 EICOS_TILES=2 EICOS_W2=0 EICOS_CONE_ORDER=1
python tools/dev/r2_case.py 322370 1 0 EICOS_THREADS=256 EICOS_TILES=1 EICOS_LDSRES=0 EICOS_DUAL=0 EICOS_FAC_DEFER=0
} > gpurun_out/r4_cases.log 2>&1
cut -c1-420 gpurun_out/r4_cases.log
