"""Dev script: distribution of per-instance solve time inside one launch (batch <= resident workgroups)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from eicos_amd.generate import feasible_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pat, sets = read_epb('tests/golden/MPC02.epb')
d = feasible_batch(pat, sets[0], 0, min(B, 256))
tile = lambda a: np.tile(a, ((B + a.shape[0] - 1) // a.shape[0], 1))[:B]
g = BatchSolver(pat, B)
g.update(tile(d['Gpr']), tile(d['Apr']), tile(d['c']), tile(d['h']), tile(d['b']))
g.solve(); g.solve()
ms = g.last_solve_ms(); ia = g.info_arrays()
tot = np.array([g.debug_trace(i)[-1][6] for i in range(B)])
it = ia['iter']; ns = ia['n_ldlsolve']
print(f"B={B} kernel {ms:.2f} ms; per-instance total us: min {tot.min():.0f} p50 {np.median(tot):.0f} p90 {np.percentile(tot,90):.0f} max {tot.max():.0f}")
print("iters min/mean/max", it.min(), it.mean(), it.max(), " ldlsolves min/mean/max", ns.min(), ns.mean(), ns.max())
per = tot / ns
print("us per ldl solve: min %.1f p50 %.1f p90 %.1f max %.1f" % (per.min(), np.median(per), np.percentile(per, 90), per.max()))
# by block index (placement): mean time of consecutive groups of 64 instances
print("mean total by group of 64:", [int(tot[k:k+64].mean()) for k in range(0, B, 64)])
print("corr(total, ldlsolves) %.3f" % np.corrcoef(tot, ns)[0, 1])
print("slowest 8:", np.argsort(tot)[-8:], tot[np.argsort(tot)[-8:]].astype(int), ns[np.argsort(tot)[-8:]])
