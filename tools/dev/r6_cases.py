"""Dev (GPU): the class-(b) cases of the round-6 campaign again -- as the campaign ran them (U in LDS where it fits, fused / staged call for
the cases that drew it) against the same variant with EICOS_UBL=0 through update + solve: are the new paths bit-identical on them?"""
import os, sys
os.environ["EICOS_EXPERIMENT"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eicos_amd
from eicos_amd.generate import feasible_batch, random_socp_pattern
CASES = {720270: (124, 0, 56, [4, 1, 33, 33], {'EICOS_IDX16': '0', 'EICOS_LDSRES': '0', 'EICOS_DUAL': '0', 'EICOS_FAC_DEFER': '0', 'EICOS_CONE_ORDER': '1'}),
         720536: (143, 37, 80, [7, 12], {'EICOS_THREADS': '512', 'EICOS_NLDS': '0', 'EICOS_FAC_DEFER': '1', 'EICOS_CONE_ORDER': '1'}),
         721107: (141, 4, 79, [2, 33, 2, 2, 3, 40], {'EICOS_NLDS': '1', 'EICOS_TILES': '1', 'EICOS_LDSRES': '0', 'EICOS_FAC_DEFER': '1', 'EICOS_CONE_ORDER': '1'}),
         721433: (141, 44, 46, [2, 2, 7, 33, 7, 4, 12], {'EICOS_THREADS': '128', 'EICOS_NLDS': '2', 'EICOS_TILES': '2', 'EICOS_UBL': '0'}),
         721790: (187, 30, 113, [33, 3, 7, 3, 4], {'EICOS_NLDS': '0', 'EICOS_TILES': '1', 'EICOS_DUAL': '0'})}
for seed, (n, p, l, q, var) in CASES.items():
    pat, base = random_socp_pattern(n, p, l, q, density=0.05 / 3, seed=seed)
    d = feasible_batch(pat, base, 0, 3, seed=seed)
    out = []
    for mode in ("campaign", "fused", "classic-noubl"):
        for k in list(os.environ):
            if k.startswith("EICOS_") and k != "EICOS_EXPERIMENT":
                del os.environ[k]
        os.environ.update(var)
        if mode == "classic-noubl":
            os.environ["EICOS_UBL"] = "0"
        g = eicos_amd.BatchSolver(pat, 3)
        if mode == "fused":
            x = np.zeros((3, pat.n)); codes = g.update_solve(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], x_out=x)
        else:
            g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve(); x = g.solution()
        ia = g.info_arrays()
        out.append((g.kernel_build(), g.last_update_path(), codes.copy(), ia["iter"].copy(), x.copy()))
        g.close()
    same = all(np.array_equal(out[0][k], o[k]) for o in out[1:] for k in (2, 3, 4))
    print(seed, [(o[0], o[1]) for o in out], "codes", out[0][2], "iters", out[0][3], "-> all three bit-identical:", same, flush=True)
