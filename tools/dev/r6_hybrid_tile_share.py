import os, sys
sys.path.insert(0, ".")
import numpy as np
from eicos_amd import BatchSolver, read_epb
from eicos_amd.generate import perturbed_batch
for name in ("lp_bandm", "lp_agg", "lp_beaconfd"):
    pat, sets = read_epb(f"tests/golden/{name}.epb")
    B = 256
    d = perturbed_batch(pat, sets[0], 0, B)
    g = BatchSolver(pat, B); g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    for r in range(2): g.solve()
    ia = g.info_arrays(); rows = []
    for i in range(0, 64, 8):
        tr = g.debug_trace(i)[-1]; rows.append([tr[8], tr[9], ia["n_sweep"][i], tr[1] + tr[5], tr[0], tr[7]])
    r = np.mean(rows, axis=0)
    print(name, g.kernel_build(), "tile fwd %.2f bwd %.2f us per solve pass; all sweeps %.1f us per pass-over-L; factor total %.0f of which tile part %.0f (per solve-call sums over the solve)" % (r[0] / r[2], r[1] / r[2], r[3] / r[2], r[4], r[5]))
    g.close()
