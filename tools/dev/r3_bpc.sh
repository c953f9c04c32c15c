export TMPDIR=/tmp EICOS_EXPERIMENT=1
for bpc in 2 3; do for B in 1024 768 1536; do EICOS_FORCE_BLOCKS_PER_CU=$bpc python tools/dev/gpu_sweep.py MPC02 $B 3 2>&1 | head -1 | cut -c1-200; done; done
