export TMPDIR=/tmp EICOS_EXPERIMENT=1
for p in lp_afiro lp_blend lp_adlittle; do
  for B in 2048 8192; do
    python tools/dev/gpu_sweep.py $p $B 3 2>&1 | head -1 | cut -c1-220
    EICOS_KI=2 EICOS_TILES=0 python tools/dev/gpu_sweep.py $p $B 3 2>&1 | head -1 | cut -c1-220
    EICOS_TILES=0 python tools/dev/gpu_sweep.py $p $B 3 2>&1 | head -1 | cut -c1-220
  done
done
