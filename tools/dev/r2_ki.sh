export TMPDIR=/tmp
for ki in 2 1; do for t in 512 256; do
EICOS_KI=$ki EICOS_THREADS=$t python tools/dev/gpu_sweep.py MPC02 1024 2 2>&1 | tail -3
done; done > gpurun_out/r2_ki.log 2>&1
EICOS_KI=2 python tools/dev/gpu_sweep.py MPC02 512 2 >> gpurun_out/r2_ki.log 2>&1
EICOS_KI=1 python tools/dev/gpu_sweep.py MPC02 512 2 >> gpurun_out/r2_ki.log 2>&1
EICOS_KI=1 EICOS_THREADS=512 python tools/dev/gpu_sweep.py MPC02 256 2 >> gpurun_out/r2_ki.log 2>&1
cat gpurun_out/r2_ki.log
