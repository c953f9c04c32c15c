export TMPDIR=/tmp
python tools/dev/r2_dbg1.py > gpurun_out/r2_dbg1.log 2>&1
tail -5 gpurun_out/r2_dbg1.log
