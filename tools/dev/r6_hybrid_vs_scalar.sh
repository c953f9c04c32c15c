export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for w in "lp_bandm 256 0" "lp_agg 256 0" "lp_beaconfd 256 0" "lp_bnl1 256 0"; do
printf "auto    "; python tools/dev/r4_phases.py $w | head -2
printf "scalar  "; EICOS_TILES=0 python tools/dev/r4_phases.py $w | head -2
done; done 2>&1 | grep -v "Exception\|Broken" | cut -c1-220
