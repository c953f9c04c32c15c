#!/bin/bash
# Round 5, first GPU call: the suite, the x-error measurement, the default bench line (with host_e2e).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r5_pytest.log
tail -5 gpurun_out/r5_pytest.log
timeout 600 python tools/dev/r5_xerr.py > gpurun_out/r5_xerr.log 2>&1; cat gpurun_out/r5_xerr.log
timeout 900 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_bench.json").read().strip().splitlines()[-1])
print("value",d["value"],"frac",d["roofline"]["frac"])
for k,v in d["config"]["summary"].items(): print(k, json.dumps(v))
PY
tail -3 gpurun_out/r5_bench.err | cut -c1-600
