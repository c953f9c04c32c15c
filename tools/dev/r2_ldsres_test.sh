#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
python - <<'PY'
import eicos_amd
from eicos_amd import read_epb
for nm in ["lp_afiro","issue98","infeasible1","update_data","feas","unboundedLP1","emptyProblem","lp_adlittle","lp_blend"]:
    pat, sets = read_epb(f"tests/golden/{nm}.epb")
    g = eicos_amd.BatchSolver(pat, 4); d = g.dims(); print(nm, d["dim_K"], d["threads_per_block"], d["lds_resident"], d["factor_path"], d["lds_bytes"]); g.close()
PY
