# round 3: head of the refinement residual in LDS -- regression + same-box A/B
export TMPDIR=/tmp EICOS_EXPERIMENT=1
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do for e in EICOS_E_LDS=0 EICOS_E_LDS=1; do for p in "MPC02 1024" "MPC02 512" "lp_bandm 256" "lp_25fv47 256" "lp_bnl1 256" "lp_agg3 256"; do set -- $p; echo "--- $e $p"; env $e python tools/dev/gpu_sweep.py $1 $2 3 2>&1 | grep -v "^   factor" | cut -c1-330; done; done; done
