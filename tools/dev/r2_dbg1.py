import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../tests")
import numpy as np
np.set_printoptions(linewidth=250, precision=9)
from conftest import load_fixture
import eicos_amd
from oracle.oracle import OracleSolver
pat, sets = load_fixture("unboundedMaxSqrt")
o = OracleSolver(pat, sets[0]); oc = o.solve(); to = o.trace()
g = eicos_amd.BatchSolver(pat, 1)
g.update(*[np.repeat(a[None, :], 1, 0) for a in (sets[0].Gpr, sets[0].Apr, sets[0].c, sets[0].h, sets[0].b)])
gc = g.solve(); gi = g.info()[0]
tg = g.debug_trace(0)
print("oracle", oc, o.info(), "\ngpu", gc, gi)
cols = eicos_amd.BatchSolver.TRACE_COLS
print(cols)
for i in range(20):
    print(i, "G", tg[i])
    if i < len(to):
        print(i, "O", to[i])
        print(i, "rel", np.abs(tg[i] - to[i]) / np.maximum(np.abs(to[i]), 1e-300))
