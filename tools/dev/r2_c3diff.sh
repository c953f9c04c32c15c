export TMPDIR=/tmp
python tools/dev/r2_c3diff.py lp_beaconfd lp_agg lp_agg2 lp_agg3 lp_25fv47 > gpurun_out/r2_c3diff.log 2>&1
EICOS_TILES=0 python tools/dev/r2_c3diff.py lp_beaconfd lp_25fv47 >> gpurun_out/r2_c3diff.log 2>&1
cat gpurun_out/r2_c3diff.log
