export TMPDIR=/tmp
for round in 1 2; do for lib in "" build_exp/libe3.so build_exp/libe4.so; do
EICOS_AMD_LIB=$lib python tools/dev/gpu_sweep.py dense-front 512 2 2>&1 | head -2 | cut -c1-330
done; done > gpurun_out/r2_ab.log 2>&1
cat gpurun_out/r2_ab.log
