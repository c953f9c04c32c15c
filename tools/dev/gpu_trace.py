"""Dev script: per-iteration trace, oracle vs GPU (needs a -DEICOS_TRACE build via EICOS_AMD_LIB)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from oracle.oracle import OracleSolver
name = sys.argv[1]
pat, sets = read_epb(f'tests/golden/{name}.epb')
v = sets[0]
o = OracleSolver(pat, v); print("oracle exit", o.solve(), flush=True)
g = BatchSolver(pat, 1)
g.update(v.Gpr[None], v.Apr[None], v.c[None], v.h[None], v.b[None])
print("gpu exit", g.solve(), flush=True)
to, tg = o.trace(), g.debug_trace(0)
np.set_printoptions(linewidth=250, precision=4)
for i in range(max(len(to), 25)):
    if i < len(to): print("o", i, to[i])
    print("g", i, tg[i])
