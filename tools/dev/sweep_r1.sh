for p in lp_afiro lp_bandm lp_bnl1; do for T in 128 256 512; do
  EICOS_THREADS=$T python tools/dev/gpu_sweep.py $p 256 2 2>&1 | head -1 | cut -c1-200
done; done
for T in 256 512 1024; do EICOS_THREADS=$T python tools/dev/gpu_sweep.py lp_25fv47 256 1 2>&1 | head -1 | cut -c1-200; done
