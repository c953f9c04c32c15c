#!/usr/bin/env python3
"""Round 5 probe: what would an INTERNAL variable order buy?  The rows of the three product plans (columns of [A; G], rows of A, LP rows of G) are
sorted by length -- as a plain equivalent problem handed to the unchanged solver: P_z G P_x', P_y A P_x', P_x c, P_y b, P_z h -- and timed against
the original order.  usage: python tools/dev/r5_perm_probe.py [batch] [soc 0/1]"""
import os, sys
import numpy as np
from scipy.sparse import csc_matrix
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import eicos_amd
from eicos_amd.generate import feasible_batch, mpc_soc_variant
from eicos_amd.problem_io import Pattern

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
soc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
pat, sets = eicos_amd.read_epb(os.path.join(ROOT, "tests", "golden", "MPC02.epb"))
if soc:
    pat = mpc_soc_variant(pat, sets[0])
d = feasible_batch(pat, sets[0], 0, B)
G = csc_matrix((np.arange(1, pat.nnzG + 1, dtype=np.float64), pat.Gir, pat.Gjc), shape=(pat.m, pat.n))  # values = entry id + 1
A = csc_matrix((np.arange(1, pat.nnzA + 1, dtype=np.float64), pat.Air, pat.Ajc), shape=(pat.p, pat.n))
collen = np.diff(pat.Gjc) + np.diff(pat.Ajc)
px = np.argsort(-collen, kind="stable")                        # new column k = old column px[k]
py = np.argsort(-np.diff(A.tocsr().indptr), kind="stable")
glen = np.diff(G.tocsr().indptr)
pz = np.concatenate([np.argsort(-glen[: pat.l], kind="stable"), np.arange(pat.l, pat.m)])  # LP rows sorted, cone rows kept
Gp = G[pz][:, px].tocsc(); Gp.sort_indices()
Ap = A[py][:, px].tocsc(); Ap.sort_indices()
gmap, amap = Gp.data.astype(np.int64) - 1, Ap.data.astype(np.int64) - 1   # new entry -> old entry
ppat = Pattern(pat.n, pat.m, pat.p, pat.l, pat.q, Gp.indptr.astype(np.int32), Gp.indices.astype(np.int32), Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32))
dp = dict(Gpr=np.ascontiguousarray(d["Gpr"][:, gmap]), Apr=np.ascontiguousarray(d["Apr"][:, amap]), c=np.ascontiguousarray(d["c"][:, px]),
          h=np.ascontiguousarray(d["h"][:, pz]), b=np.ascontiguousarray(d["b"][:, py]))


def run(p_, d_, tag):
    g = eicos_amd.BatchSolver(p_, B)
    g.update(d_["Gpr"], d_["Apr"], d_["c"], d_["h"], d_["b"])
    ms = []
    for _ in range(4):
        codes = g.solve(); ms.append(g.last_solve_ms())
    ia = g.info_arrays(); x = g.solution(); dm = g.dims(); g.close()
    print(f"{tag}: nnzL={dm['nnzL']} levels={dm['nlevels']} ms={min(ms):.2f} iters={ia['iter'].sum()} -> {ia['iter'].sum() / min(ms) * 1e3:.0f} iter/s ok={(codes == 0).sum()}", flush=True)
    return ia, x


for rep in range(2):
    ia0, x0 = run(pat, d, "original order ")
    ia1, x1 = run(ppat, dp, "rows by length ")
xb = np.empty_like(x1); xb[:, px] = x1
print("iterations equal on", int((ia0["iter"] == ia1["iter"]).sum()), "of", B, " max |x - x'| rel", float((np.abs(xb - x0).max(axis=1) / np.maximum(1, np.abs(x0).max(axis=1))).max()))
