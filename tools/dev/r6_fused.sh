#!/bin/bash
# Round 6: eicos_batch_update_solve (updateData fused into the solve launch) -- tests, then the host round trip of bench.py
export TMPDIR=/tmp EICOS_EXPERIMENT=1
cd "$GRAFT_REPO_ROOT" || exit 1; mkdir -p gpurun_out
{ timeout 1200 python -m pytest tests -m gpu -x -q -k "fused or host_pointer or ring_of or lds_resident or update_kernels or keep_semantics" 2>&1 | tail -8
python - <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench, eicos_amd
pat, sets = eicos_amd.read_problem("tests/golden/MPC02.epb")
for rep in range(2):
    he = bench.host_e2e(pat, sets, 1024, 0, 1.0)
    print(json.dumps({k: ({kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in he.items()}))
PY
for w in "MPC02 1024 0" "MPC02 512 0"; do python tools/dev/r4_phases.py $w | head -1; done
} > gpurun_out/r6_fused.log 2>&1
grep -v "Exception ignored\|BrokenPipe" gpurun_out/r6_fused.log | cut -c1-1200
