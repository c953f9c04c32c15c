# Round 6: small block systems swept serially by one wavefront (tiles.cpp: serial_max), same box, interleaved: off / 48 / 200 operations
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for w in "lp_bandm 256 0" "lp_agg 256 0" "lp_beaconfd 256 0" "lp_25fv47 256 0" "lp_bnl1 256 0" "lp_agg3 256 0"; do
for m in 0 48 200; do printf "serial_max=%-4s " $m; EICOS_TILE_SERIAL_MAX=$m python tools/dev/r4_phases.py $w | head -1; done
done; done 2>&1 | grep -v "Exception\|Broken" | cut -c1-200
