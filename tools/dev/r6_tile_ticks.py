import os, sys
sys.path.insert(0, ".")
import numpy as np
from eicos_amd import BatchSolver
from eicos_amd.generate import feasible_batch, dense_front_pattern
pat, base = dense_front_pattern(2000, 32, 64)
B = 256
d = feasible_batch(pat, base, 0, B)
g = BatchSolver(pat, B); g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
for r in range(2): g.solve()
ia = g.info_arrays()
rows = []
for i in range(0, 64, 8):
    tr = g.debug_trace(i)[-1]; rows.append([tr[8], tr[9], ia["n_sweep"][i], ia["iter"][i] + 1, tr[1] + tr[5]])
r = np.mean(rows, axis=0)
print("forward sweeps %.0f us, backward %.0f us per solve pass over L (n_sweep %.1f): fwd %.1f bwd %.1f us per pass; ldl total %.0f" % (r[0], r[1], r[2], r[0] / r[2], r[1] / r[2], r[4]))
