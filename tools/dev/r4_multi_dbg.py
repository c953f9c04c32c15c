import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, "tests")
import numpy as np, eicos_amd
from conftest import load_fixture
from eicos_amd.generate import feasible_batch
pat, sets = load_fixture("MPC02")
B = 1024
d = feasible_batch(pat, sets[0], 0, B)
def run(mk, upd):
    g = mk(); upd(g); codes = g.solve(); ia = g.info_arrays(); x = g.solution(); g.close(); return codes, ia, x
c1, ia1, x1 = run(lambda: eicos_amd.BatchSolver(pat, B, device=0), lambda g: g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]))
c1b, ia1b, x1b = run(lambda: eicos_amd.BatchSolver(pat, B, device=0), lambda g: g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]))
print("single vs single again: x equal", np.array_equal(x1, x1b))
ca, iaa, xa = run(lambda: eicos_amd.BatchSolver(pat, 512, device=0), lambda g: g.update(*[d[k][:512] for k in ("Gpr", "Apr", "c", "h", "b")]))
print("512-handle vs first half of 1024-handle: x equal", np.array_equal(xa, x1[:512]), "iters equal", np.array_equal(iaa["iter"], ia1["iter"][:512]),
      "ndiff rows", int((np.abs(xa - x1[:512]).max(axis=1) > 0).sum()), "max rel diff", float(np.abs(xa - x1[:512]).max() / np.abs(x1).max()))
cm, iam, xm = run(lambda: eicos_amd.MultiBatchSolver(pat, B, [0, 0]), lambda g: g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]))
print("multi vs single: x equal", np.array_equal(xm, x1), "rows differing", np.flatnonzero(np.abs(xm - x1).max(axis=1) > 0)[:20], "codes", np.array_equal(cm, c1), "iter", np.array_equal(iam["iter"], ia1["iter"]))
print("multi first half vs 512-handle", np.array_equal(xm[:512], xa))
for k in ("n_ldlsolve", "n_factor", "pcost"):
    print(k, np.array_equal(iam[k], ia1[k]))
