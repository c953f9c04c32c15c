"""Dev (GPU): dense-front pattern of a given size with T = 512 (one workgroup per CU) against T = 256 (two per CU when LDS allows)."""
import os, sys
os.environ["EICOS_EXPERIMENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from eicos_amd import BatchSolver
from eicos_amd.generate import feasible_batch, dense_front_pattern
n, k, d, B = (int(v) for v in sys.argv[1:5])
pat, base = dense_front_pattern(n, k, d)
dd = feasible_batch(pat, base, 0, min(B, 128))
tile = lambda a: np.tile(a, ((B + a.shape[0] - 1) // a.shape[0], 1))[:B]
for T in ("512", "256"):
    os.environ["EICOS_THREADS"] = T
    g = BatchSolver(pat, B)
    g.update(tile(dd["Gpr"]), tile(dd["Apr"]), tile(dd["c"]), tile(dd["h"]), tile(dd["b"]))
    ms = []
    for r in range(3):
        codes = g.solve(); ms.append(g.last_solve_ms())
    ia = g.info_arrays(); dm = g.dims()
    print(f"dense-front n={n} k={k} d={d} B={B} T={dm['threads_per_block']} resident={dm['resident_blocks']} lds={dm['lds_bytes']} dual={dm['dual_rhs']} path={dm['factor_path']}: "
          f"ms={min(ms):.2f} -> {ia['iter'].sum()/min(ms)*1e3:.0f} iter/s ok={(codes==0).sum()}", flush=True)
    g.close()
