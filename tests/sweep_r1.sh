python tests/gpu_quick.py feas issue98 unboundedMaxSqrt lp_afiro lp_agg update_data MPC02 2>&1 | grep -v "^MPC02 batch\|second solve\|inst " 
for w in 2 3 4 6 8; do for lds in 1 0; do for T in 256 512; do
  EICOS_AMD_LIB=$PWD/eicos_amd/libeicos_amd_w$w.so EICOS_WS_LDS=$lds EICOS_THREADS=$T python tests/gpu_sweep.py MPC02 1024 2 2>&1 | tail -1
done; done; done
for w in 4 8; do EICOS_AMD_LIB=$PWD/eicos_amd/libeicos_amd_w$w.so EICOS_WS_LDS=0 python tests/gpu_sweep.py MPC02 4096 2 2>&1 | tail -1; done
