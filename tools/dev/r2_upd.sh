export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2_upd_t.log
cat gpurun_out/r2_upd_t.log
python - <<'PY' > gpurun_out/r2_upd.log 2>&1
import os, sys, numpy as np
sys.path.insert(0, '.')
import torch, eicos_amd
from eicos_amd.generate import feasible_batch, mpc_soc_variant
pat, sets = eicos_amd.read_problem('tests/golden/MPC02.epb')
for soc in (False, True):
    p_ = mpc_soc_variant(pat) if soc else pat
    B = 1024
    d = feasible_batch(p_, sets[0], 0, 64)
    tile = lambda a: np.tile(a, (B // 64, 1))
    dev = {k: torch.from_numpy(tile(v)).cuda() for k, v in d.items()}
    res = {}
    for mode in ("1", "0"):
        os.environ["EICOS_UPDATE_LDS"] = mode
        g = eicos_amd.BatchSolver(p_, B)
        ms = []
        for _ in range(5):
            g.update_device(*[dev[k].data_ptr() if dev[k].numel() else 0 for k in ("Gpr", "Apr", "c", "h", "b")]); g.sync(); ms.append(g.last_update_ms())
        g.solve(); x = g.solution().copy(); it = g.info_arrays()["iter"].copy()
        # keep semantics: update only c (NULL groups keep), then again everything
        g.update_device(0, 0, dev["c"].data_ptr(), 0, 0); g.solve(); x2 = g.solution().copy()
        res[mode] = (x, it, x2); g.close()
        print("soc", soc, "EICOS_UPDATE_LDS", mode, "update ms", ["%.3f" % m for m in ms])
    print("   bit-identical results:", np.array_equal(res["1"][0], res["0"][0]), np.array_equal(res["1"][1], res["0"][1]), np.array_equal(res["1"][2], res["0"][2]), np.array_equal(res["1"][0], res["1"][2]))
PY
cat gpurun_out/r2_upd.log
