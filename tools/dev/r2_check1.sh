export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2_pytest1.log
cat gpurun_out/r2_pytest1.log
