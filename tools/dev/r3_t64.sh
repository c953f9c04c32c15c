# one-off: needs the 64-thread instantiation of the LDS-resident k_solve, which was built for this measurement and removed again
# (profiles/r03_t64_single_wavefront.log).  Kept as the record of what was run.
export TMPDIR=/tmp EICOS_EXPERIMENT=1
for p in lp_afiro lp_adlittle lp_blend; do for t in 128 64; do echo "--- $p T=$t"; EICOS_TILES=0 EICOS_THREADS=$t python tools/dev/gpu_sweep.py $p 256 3 2>&1 | cut -c1-300 | head -2; done; done
python tools/dev/gpu_sweep.py lp_afiro 256 3 2>&1 | cut -c1-300 | head -2
EICOS_THREADS=64 python tools/dev/gpu_quick.py lp_afiro issue98 infeasible1 update_data feas 2>&1 | cut -c1-200 | head -12
