export TMPDIR=/tmp EICOS_EXPERIMENT=1
build_exp/sweep_bench tests/golden/MPC02.epb 512 200 > gpurun_out/sb2.log 2>&1
build_exp/sweep_bench tests/golden/MPC02.epb 256 200 >> gpurun_out/sb2.log 2>&1
build_exp/sweep_bench tests/golden/MPC02.epb 768 200 >> gpurun_out/sb2.log 2>&1
cat gpurun_out/sb2.log
# T=512 at two workgroups per CU (128 VGPRs) vs the default
( EICOS_AMD_LIB=$PWD/build_exp/libw512.so EICOS_THREADS=512 EICOS_FORCE_BLOCKS_PER_CU=2 python tools/dev/gpu_sweep.py MPC02 512 3
  EICOS_AMD_LIB=$PWD/build_exp/libw512.so EICOS_THREADS=512 EICOS_FORCE_BLOCKS_PER_CU=2 python tools/dev/gpu_sweep.py MPC02 1024 3
  python tools/dev/gpu_sweep.py MPC02 1024 3 ) 2>&1 | grep -v "^   factor" | cut -c1-400 | tee gpurun_out/w512.log
