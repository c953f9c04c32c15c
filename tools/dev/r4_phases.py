"""Dev (GPU): per-stage in-kernel timers of instance 0 for the LP and SOC variants of a pattern at one batch size.
usage: python tools/dev/r4_phases.py [pattern] [batch] [soc 0/1]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from eicos_amd.generate import feasible_batch, mpc_soc_variant, dense_front_pattern
name = sys.argv[1] if len(sys.argv) > 1 else "MPC02"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
soc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if name == "dense-front":
    pat, base = dense_front_pattern(2000, 32, 64)
else:
    pat, sets = read_epb(f"tests/golden/{name}.epb"); base = sets[0]
if soc:
    pat = mpc_soc_variant(pat)
d = feasible_batch(pat, base, 0, B)
g = BatchSolver(pat, B)
g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
ms = []
for r in range(3):
    codes = g.solve(); ms.append(g.last_solve_ms())
ia = g.info_arrays(); dm = g.dims()
print(f"{name} soc={soc} B={B} T={dm['threads_per_block']} resident={dm['resident_blocks']} build={g.kernel_build()} levels={dm['nlevels']} nnzL={dm['nnzL']} "
      f"ms={min(ms):.2f} iters={ia['iter'].sum()} -> {ia['iter'].sum()/min(ms)*1e3:.0f} iter/s  ldl/iter={ia['n_ldlsolve'].sum()/ia['iter'].sum():.2f}", flush=True)
if B <= dm["resident_blocks"]:
    rows = []
    for i in range(0, min(B, 64), 8):
        tr = g.debug_trace(i)[-1]; it = max(1, ia["iter"][i] + 1)
        rows.append([tr[0] / it, (tr[1] + tr[5]) / it, tr[2] / it, tr[3] / it, tr[4] / it, tr[6] / it, ia["n_ldlsolve"][i] / it])
    r = np.mean(rows, axis=0)
    print("   us per pass (mean of 8 instances): factor %.0f  ldl-sweeps %.0f  kkt-resid %.0f  kkt-post %.0f  resid-stage %.0f  total %.0f   solves/pass %.2f" % tuple(r))
    print("   per LDL solve: sweeps %.1f us, refinement residual %.1f us" % (r[1] / r[6], r[2] / r[6]))
g.close()
