import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, eicos_amd
from conftest import fuzz_case_r3
from eicos_amd.problem_io import Values
from oracle.oracle import OracleSolver
pat, d = fuzz_case_r3(441378, 3)
g = eicos_amd.BatchSolver(pat, 3); g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve(); ia = g.info_arrays()
tg = g.debug_trace(2)
o = OracleSolver(pat, Values(d["Gpr"][2], d["Apr"][2], d["c"][2], d["h"][2], d["b"][2])); oc = o.solve(); to = o.trace(); oi = o.info()
print("gpu iter", ia["iter"][2], "oracle iter", oi["iter"], "nitref", [ia[k][2] for k in ("nitref1","nitref2","nitref3")], [oi[k] for k in ("nitref1","nitref2","nitref3")])
np.set_printoptions(linewidth=250, precision=4)
cols = ("pcost","dcost","gap","pres","dres","k/t","mu","step","sigma","tau","kap","nitref3")
print(cols)
for it in range(10, 19):
    print(it, "G", tg[it] if it <= ia["iter"][2] else None)
    print(it, "O", to[it] if it < len(to) else None)
