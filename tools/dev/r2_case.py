"""Dev script (GPU): one fuzz case in detail.  usage: python tools/dev/r2_case.py <seed> <scale> <dyn 0|1> KEY=VAL ..."""
import os, sys
os.environ["EICOS_EXPERIMENT"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eicos_amd
from eicos_amd.generate import feasible_batch, random_socp_pattern
from eicos_amd.problem_io import Values
from oracle.oracle import OracleSolver
seed = int(sys.argv[1]); scale = int(sys.argv[2]); dyn = sys.argv[3] == "1"
envv = dict(a.split("=") for a in sys.argv[4:])
rng = np.random.default_rng(seed)
n = int(rng.integers(2, 70 * scale)); p = int(rng.integers(0, max(1, n // 2))); l = int(rng.integers(0, 50 * scale)); nc = int(rng.integers(0, 5 * scale))
q = [int(rng.choice([1, 2, 3, 4, 7, 12, 33, 40, 64])) for _ in range(nc)]
if l + sum(q) == 0: l = 3
dens = float(rng.choice([0.05, 0.15, 0.3, 0.6])) / scale
pat, base = random_socp_pattern(n, p, l, q, density=dens, seed=seed)
d = feasible_batch(pat, base, 0, 3, seed=seed)
print("case", seed, "n", n, "p", p, "l", l, "q", q, "dens", dens)
import scipy.sparse as sp
G = [sp.csc_matrix((d["Gpr"][i], pat.Gir, pat.Gjc), shape=(pat.m, pat.n)) for i in range(3)]
A = [sp.csc_matrix((d["Apr"][i], pat.Air, pat.Ajc), shape=(pat.p, pat.n)) for i in range(3)]
xs = {}
KEYS = ("EICOS_LDSRES", "EICOS_TILES", "EICOS_THREADS", "EICOS_NLDS", "EICOS_IDX16", "EICOS_DUAL", "EICOS_FAC_DEFER", "EICOS_W2", "EICOS_CONE_ORDER")
for env in (envv, {}, {"EICOS_TILES": "0", "EICOS_LDSRES": "0"}):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    g = eicos_amd.BatchSolver(pat, 3)
    if dyn: g.set_dynamic_regularization(2e-7, 1e-13)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve(); ia = g.info_arrays(); dm = g.dims(); x = g.solution(); y, z, s = g.duals()
    print(env, "path", dm["factor_path"], "ldsres", dm["lds_resident"], "codes", [int(c) for c in codes], "iter", [int(v) for v in ia["iter"]], "pcost", ia["pcost"], "pres", ia["pres"], "dres", ia["dres"], "gap", ia["gap"])
    for i in range(3):
        rp = np.abs(A[i] @ x[i] - d["b"][i]).max() if pat.p else 0.0
        rg = np.abs(G[i] @ x[i] + s[i] - d["h"][i]).max()
        print("    inst", i, "|Ax-b|", rp, "|Gx+s-h|", rg, "c'x", d["c"][i] @ x[i])
    xs[str(env)] = x.copy()
    g.close()
for i in range(3):
    o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
    if dyn: o.set_dynamic_regularization(2e-7, 1e-13)
    oc = o.solve(); oi = o.info(); ox = o.x()
    print("oracle", i, oc, oi["iter"], oi["pcost"], oi["pres"], oi["dres"], oi["gap"], " |x-x_gpu| per variant:", {k: float(np.abs(v[i] - ox).max()) for k, v in xs.items()}, "|x|", np.abs(ox).max())
