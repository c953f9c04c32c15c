export TMPDIR=/tmp
bash tools/dev/ab_env_r1.sh 1024 EICOS_BLOCKS_PER_CU=2 EICOS_BLOCKS_PER_CU=3 > gpurun_out/r2_ab.log 2>&1
bash tools/dev/ab_env_r1.sh 1536 EICOS_BLOCKS_PER_CU=2 EICOS_BLOCKS_PER_CU=3 >> gpurun_out/r2_ab.log 2>&1
bash tools/dev/ab_env_r1.sh 4096 EICOS_BLOCKS_PER_CU=2 EICOS_BLOCKS_PER_CU=3 >> gpurun_out/r2_ab.log 2>&1
cat gpurun_out/r2_ab.log
