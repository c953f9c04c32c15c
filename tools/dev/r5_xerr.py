#!/usr/bin/env python3
"""Round 5: how close is x to the oracle's, really?  Max over instances of ||x - x_ref||_inf / max(1, ||x_ref||_inf), split into the
instances whose iteration counts agree and those that differ by one pass.  Run on the GPU box:
    python tools/dev/r5_xerr.py > gpurun_out/r5_xerr.log"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import eicos_amd
from eicos_amd.generate import feasible_batch, mpc_soc_variant, dense_front_pattern
from oracle import oracle as orc

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")


def one(name, pat, base, B):
    d = feasible_batch(pat, base, 0, B)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays(); x = g.solution(); g.close()
    r = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], len(os.sched_getaffinity(0)), want_x=True)
    it_o, it_g = r["iters"].astype(int), ia["iter"].astype(int)
    xs = np.maximum(1.0, np.abs(r["x"]).max(axis=1))
    err = np.abs(x - r["x"]).max(axis=1) / xs
    same = it_o == it_g
    pc = np.abs(ia["pcost"] - r["pcost"]) / np.maximum(1.0, np.abs(r["pcost"]))
    out = {"workload": name, "B": B, "codes_equal": int((codes == r["exitcodes"]).sum()), "iters_equal": int(same.sum()),
           "iters_pm1": int((np.abs(it_o - it_g) == 1).sum()), "iters_beyond": int((np.abs(it_o - it_g) > 1).sum()),
           "xerr_equal_max": float(err[same].max()) if same.any() else None, "xerr_equal_p99": float(np.quantile(err[same], 0.99)) if same.any() else None,
           "xerr_pm1_max": float(err[~same].max()) if (~same).any() else None, "pcost_rel_max": float(pc.max())}
    print(json.dumps(out), flush=True)


pat, sets = eicos_amd.read_epb(os.path.join(ROOT, "tests", "golden", "MPC02.epb"))
one("MPC02 b1024", pat, sets[0], 1024)
one("MPC02 b4096", pat, sets[0], 4096)
spat = mpc_soc_variant(pat, sets[0])
one("MPC02-SOC b1024", spat, sets[0], 1024)
dpat, dbase = dense_front_pattern(2000, 32, 64)
one("dense-front b512", dpat, dbase, 512)
