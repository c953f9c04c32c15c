export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ timeout 1200 python -m pytest tests -m gpu -x -q -k "dense_front or tile or hybrid or ldl_factor or every_kernel or dynamic_regularisation or config3 or fatal_seeds or random_socp or big_cone" 2>&1 | tail -5
for rep in 1 2; do
python tools/dev/r4_phases.py dense-front 512 0 | head -1
EICOS_AMD_LIB=$PWD/build_exp/libbase.so python tools/dev/r4_phases.py dense-front 512 0 | head -1
python tools/dev/r4_phases.py lp_25fv47 256 0 | head -1
EICOS_AMD_LIB=$PWD/build_exp/libbase.so python tools/dev/r4_phases.py lp_25fv47 256 0 | head -1
done
python tools/dev/r4_phases.py dense-front 256 0
} > gpurun_out/r4_exp4.log 2>&1
cat gpurun_out/r4_exp4.log | cut -c1-300
