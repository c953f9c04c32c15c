"""Dev (GPU; library built with -DEICOS_SOLO_TICKS): time of the single-wavefront tree top / dense apex per LDL solve.  usage: EICOS_AMD_LIB=build_exp/libsoloticks.so python tools/dev/r5_solo_ticks.py [pattern] [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from eicos_amd.generate import feasible_batch
name = sys.argv[1] if len(sys.argv) > 1 else "MPC02"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
pat, sets = read_epb(f"tests/golden/{name}.epb")
d = feasible_batch(pat, sets[0], 0, B)
g = BatchSolver(pat, B)
g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
g.solve(); g.solve()
ia = g.info_arrays()
rows = []
for i in range(0, min(B, 64), 8):
    tr = g.debug_trace(i)[-1]
    rows.append([tr[1] / ia["n_ldlsolve"][i], tr[5] / ia["n_ldlsolve"][i]])
r = np.mean(rows, axis=0)
print(f"{name} B={B} ms={g.last_solve_ms():.2f}: per LDL solve: sweeps outside the single-wavefront part {r[0]:.1f} us, single-wavefront part (tree top / apex) {r[1]:.1f} us")
g.close()
