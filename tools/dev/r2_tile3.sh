export TMPDIR=/tmp
python -m pytest tests -m gpu -q -k "ldl_factor and dense" 2>&1 | grep -E "AssertionError|Error" | cut -c1-300 > gpurun_out/r2_tile3t.log
cat gpurun_out/r2_tile3t.log
