import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
os.environ["EICOS_EXPERIMENT"]="1"
import numpy as np, eicos_amd
from conftest import fuzz_case_r3
from eicos_amd.problem_io import Values
from oracle.oracle import OracleSolver
seed, scale, inst = 800436, 3, 1
pat, d = fuzz_case_r3(seed, scale)
o = OracleSolver(pat, Values(d["Gpr"][inst], d["Apr"][inst], d["c"][inst], d["h"][inst], d["b"][inst])); oc = o.solve(); to = o.trace(); oi = o.info()
np.set_printoptions(linewidth=250, precision=3)
for apex in ("0", "1"):
    os.environ["EICOS_APEX"] = apex
    g = eicos_amd.BatchSolver(pat, 3); g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve(); ia = g.info_arrays()
    tg = g.debug_trace(inst)
    print("apex", apex, "dims", {k: g.dims()[k] for k in ("factor_path","threads_per_block")}, "gpu iter", ia["iter"][inst], "oracle", oi["iter"], "nitref", [ia[k][inst] for k in ("nitref1","nitref2","nitref3")], [oi[k] for k in ("nitref1","nitref2","nitref3")], "ldl", ia["n_ldlsolve"][inst])
    n = min(len(tg), 26)
    print(np.array(tg[:n])[:, 7:15] if np.array(tg).shape[1] > 15 else np.array(tg[:n]))
    g.close()
print("oracle trace"); print(np.array(to)[:24])
