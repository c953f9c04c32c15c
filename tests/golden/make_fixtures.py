#!/usr/bin/env python3
"""Regenerate tests/golden/*.epb and expected.json from the reference's test DATA.

Run in the build container only (needs /root/reference, which does not exist on the
GPU box):  python tests/golden/make_fixtures.py

What is taken from the reference: the numeric problem data arrays (c, h, b, CSC of G
and A, cone sizes) that its own tests hold as C array initialisers, and the exit code
each test asserts (test/ecostester.cpp:54-72 and the mu_assert line of each header).
No reference source text is stored -- only numbers, re-encoded in the little-endian
"EPB1" container described in eicos_amd/problem_io.py.

What is NOT from the reference: `highs_optimum` -- an independent LP optimum computed
here with scipy.optimize.linprog(method="highs") for every LP-only fixture.
"""
import json
import os
import re
import struct
import sys

import numpy as np

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))

sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from eicos_amd.problem_io import parse_c_initialisers  # noqa: E402  (the package's ECOS data.h parser)


def parse_header(path):
    """Return {name: int | float | np.ndarray | None} for every scalar/array initialiser."""
    return parse_c_initialisers(open(path).read())


def problem(n, m, p, l, q, Gjc, Gir, Ajc, Air, sets):
    q = np.zeros(0, np.int32) if q is None else np.asarray(q, np.int32)
    Gjc = np.zeros(n + 1, np.int32) if Gjc is None else np.asarray(Gjc, np.int32)
    Gir = np.zeros(0, np.int32) if Gir is None else np.asarray(Gir, np.int32)
    Ajc = np.zeros(n + 1, np.int32) if Ajc is None else np.asarray(Ajc, np.int32)
    Air = np.zeros(0, np.int32) if Air is None else np.asarray(Air, np.int32)
    assert len(Gjc) == n + 1 and len(Ajc) == n + 1
    assert l == m - int(q.sum())
    norm = []
    for (Gpr, Apr, c, h, b) in sets:
        f = lambda a, k: np.zeros(k) if a is None else np.asarray(a, np.float64)
        Gpr, Apr, c, h, b = f(Gpr, len(Gir)), f(Apr, len(Air)), f(c, n), f(h, m), f(b, p)
        assert len(Gpr) == len(Gir) == Gjc[-1] and len(Apr) == len(Air) == Ajc[-1]
        assert len(c) == n and len(h) == m and len(b) == p
        norm.append((Gpr, Apr, c, h, b))
    return dict(n=n, m=m, p=p, l=l, q=q, Gjc=Gjc, Gir=Gir, Ajc=Ajc, Air=Air, sets=norm)


def write_epb(path, P):
    with open(path, "wb") as f:
        f.write(b"EPB1")
        f.write(struct.pack("<8i", P["n"], P["m"], P["p"], P["l"], len(P["q"]),
                            len(P["Gir"]), len(P["Air"]), len(P["sets"])))
        for k in ("q", "Gjc", "Gir", "Ajc", "Air"):
            f.write(P[k].astype("<i4").tobytes())
        for s in P["sets"]:
            for a in s:
                f.write(a.astype("<f8").tobytes())


def highs(P, k=0):
    from scipy.optimize import linprog
    from scipy.sparse import csc_matrix
    if len(P["q"]) or P["n"] == 0:
        return None
    Gpr, Apr, c, h, b = P["sets"][k]
    n, m, p = P["n"], P["m"], P["p"]
    G = csc_matrix((Gpr, P["Gir"], P["Gjc"]), shape=(m, n))
    A = csc_matrix((Apr, P["Air"], P["Ajc"]), shape=(p, n)) if p else None
    r = linprog(c, A_ub=G, b_ub=h, A_eq=A, b_eq=b if p else None,
                bounds=[(None, None)] * n, method="highs")
    return dict(status=int(r.status), fun=(float(r.fun) if r.status == 0 else None))


def prefixed(d, pre):
    g = lambda k: d.get(pre + k)
    return problem(g("n"), g("m"), g("p"), g("l"), g("q"), g("Gjc"), g("Gir"), g("Ajc"), g("Air"),
                   [(g("Gpr"), g("Apr"), g("c"), g("h"), g("b"))])


def main():
    expected = {}
    probs = {}

    def add(name, P, codes, cite, optval=None):
        probs[name] = P
        e = dict(exit_codes=codes, cite=cite, n=P["n"], m=P["m"], p=P["p"], l=P["l"],
                 q=[int(v) for v in P["q"]], nnzG=len(P["Gir"]), nnzA=len(P["Air"]),
                 nsets=len(P["sets"]))
        if optval is not None:
            e["reference_optval"] = optval
        hs = [highs(P, k) for k in range(len(P["sets"]))]
        if hs[0] is not None:
            e["highs"] = hs
        expected[name] = e

    # MPC02: test/MPC/MPC02.h:4-18,38 (OPTIMAL or OPTIMAL+INACC)
    d = parse_header(f"{REF}/MPC/MPC02.h")
    add("MPC02", prefixed(d, "MPC02_"), [0, 10], "test/MPC/MPC02.h:4-18,38")

    # update_data: two value sets on one pattern, test/updateData/update_data.h
    d = parse_header(f"{REF}/updateData/update_data.h")
    P = problem(d["udd_n"], d["udd_m"], d["udd_p"], d["udd_l"], None,
                d["udd_Gjc"], d["udd_Gir"], d["udd_Ajc"], d["udd_Air"],
                [(d["udd_G1pr"], d["udd_A1pr"], d["udd_c1"], d["udd_h1"], d["udd_b1"]),
                 (d["udd_G2pr"], d["udd_A2pr"], d["udd_c2"], d["udd_h2"], d["udd_b2"])])
    add("update_data", P, [0, 10], "test/updateData/update_data.h:4-9,1654-1683",
        optval=[d["udd_optval1"], d["udd_optval2"]])

    # LPnetlib: test/LPnetlib/lp_*.h (strict OPTIMAL, line 39 of each)
    for nm in ["25fv47", "adlittle", "afiro", "agg", "agg2", "agg3", "bandm", "beaconfd", "blend", "bnl1"]:
        d = parse_header(f"{REF}/LPnetlib/lp_{nm}.h")
        add(f"lp_{nm}", prefixed(d, f"lp_{nm}_"), [0], f"test/LPnetlib/lp_{nm}.h:3-18,39")

    # small local-variable style headers
    for name, path, codes in [
            ("unboundedLP1", "unboundedProblems/unboundedLP1.h", [2]),
            ("unboundedMaxSqrt", "unboundedProblems/unboundedMaxSqrt.h", [2]),
            ("infeasible1", "infeasibleProblems/infeasible1.h", [1]),
            ("emptyProblem", "emptyProblem/emptyProblem.h", [0])]:
        d = parse_header(f"{REF}/{path}")
        add(name, prefixed(d, ""), codes, f"test/{path}:4-18,33")

    # feas: dims are literals in the ECOS_setup call (test/feasibilityProblems/feas.h:21-25)
    d = parse_header(f"{REF}/feasibilityProblems/feas.h")
    add("feas", problem(1, 2, 0, 2, None, d["feas_Gp"], d["feas_Gi"], None, None,
                        [(d["feas_Gx"], None, d["feas_c"], d["feas_h"], None)]),
        [0], "test/feasibilityProblems/feas.h:4-9,21-25,36")

    # issue98: dims literal in the call (test/cvxpyProblems/githubIssue98.h:26-30)
    d = parse_header(f"{REF}/cvxpyProblems/githubIssue98.h")
    add("issue98", problem(5, 11, 0, 6, d["q"], d["Gp"], d["Gi"], None, None,
                           [(d["Gx"], None, d["c"], d["h"], None)]),
        [0], "test/cvxpyProblems/githubIssue98.h:4-13,26-30,41")

    for name, P in probs.items():
        write_epb(os.path.join(OUT, name + ".epb"), P)
    with open(os.path.join(OUT, "expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)
    tot = sum(os.path.getsize(os.path.join(OUT, n + ".epb")) for n in probs)
    print(f"wrote {len(probs)} fixtures, {tot/1e6:.2f} MB")
    for k, e in expected.items():
        print(k, e["n"], e["m"], e["p"], e["q"], e["exit_codes"], e.get("highs"), e.get("reference_optval"))


if __name__ == "__main__":
    sys.exit(main())
