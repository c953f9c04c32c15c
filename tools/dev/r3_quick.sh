# round 3: regression + timing after a kernel change
export TMPDIR=/tmp EICOS_EXPERIMENT=1
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for B in 512 1024; do python tools/dev/gpu_sweep.py MPC02 $B 3 2>&1 | grep -v "^   factor" | cut -c1-330; done
