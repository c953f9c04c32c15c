# round 3: batch 1024 on MPC02 -- resident workgroups between 2 and 3 per CU (the dispatcher fills CUs unevenly; the instance queue balances)
export TMPDIR=/tmp EICOS_EXPERIMENT=1
for rep in 1 2; do
  echo "--- default"; python tools/dev/gpu_sweep.py MPC02 1024 3 2>&1 | cut -c1-200
  for g in 576 640 704 768; do echo "--- 3 per CU allowed, grid $g"; EICOS_FORCE_BLOCKS_PER_CU=3 EICOS_GRID=$g python tools/dev/gpu_sweep.py MPC02 1024 3 2>&1 | cut -c1-200; done
done
