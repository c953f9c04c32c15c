export TMPDIR=/tmp EICOS_EXPERIMENT=1
for p in lp_bandm lp_bnl1 lp_agg; do for d in 0 1; do echo "--- $p dual=$d"; EICOS_TILES=0 EICOS_THREADS=256 EICOS_DUAL=$d python tools/dev/gpu_sweep.py $p 1024 3 2>&1 | cut -c1-330 | head -1; done; done
