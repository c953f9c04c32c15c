"""Dev script (not a pytest file): GPU vs oracle on every fixture, prints a table."""
import json, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from eicos_amd.generate import feasible_batch
from oracle.oracle import OracleSolver
exp = json.load(open('tests/golden/expected.json'))
names = sys.argv[1:] or sorted(exp)
for name in names:
    pat, sets = read_epb(f'tests/golden/{name}.epb')
    o = OracleSolver(pat, sets[0]); oc = o.solve(); oi = o.info(); ox = o.x()
    t0 = time.time(); g = BatchSolver(pat, 2); t1 = time.time()
    v = sets[0]
    rep = lambda a: np.stack([a, a])
    g.update(rep(v.Gpr), rep(v.Apr), rep(v.c), rep(v.h), rep(v.b))
    gc = g.solve(); gi = g.info()[0]; gx = g.solution()[0]
    dx = np.abs(gx - ox).max() if pat.n else 0.0
    print(f"{name:18s} oracle exit={oc:3d} it={oi['iter']:3d} pcost={oi['pcost']:+.10e} | gpu exit={gc[0]:3d}/{gc[1]:3d} it={gi['iter']:3d} pcost={gi['pcost']:+.10e} dx={dx:.2e} xmax={np.abs(ox).max() if pat.n else 0:.2e} nsolve o/g={oi['n_ldlsolve']}/{gi['n_ldlsolve']} ms={g.last_solve_ms():.2f} setup={t1-t0:.2f}s", flush=True)
    for k in range(1, len(sets)):
        v = sets[k]
        o.update(v); oc = o.solve(); oi = o.info()
        g.update(rep(v.Gpr), rep(v.Apr), rep(v.c), rep(v.h), rep(v.b)); gc = g.solve(); gi = g.info()[0]
        print(f"   update[{k}] oracle exit={oc} it={oi['iter']} pcost={oi['pcost']:+.10e} | gpu exit={gc[0]} it={gi['iter']} pcost={gi['pcost']:+.10e}")
    g.close()
# batch on MPC02
pat, sets = read_epb('tests/golden/MPC02.epb')
for B in (64, 1024):
    d = feasible_batch(pat, sets[0], 0, B)
    g = BatchSolver(pat, B)
    t0 = time.time(); g.update(d['Gpr'], d['Apr'], d['c'], d['h'], d['b']); t1 = time.time()
    codes = g.solve(); ms = g.last_solve_ms()
    ia = g.info_arrays()
    print(f"MPC02 batch {B}: update {t1-t0:.2f}s (kernel {g.last_update_ms():.2f} ms) solve {ms:.1f} ms  exit codes {np.unique(codes, return_counts=True)} iters mean {ia['iter'].mean():.2f} nsolve mean {ia['n_ldlsolve'].mean():.1f} -> {ia['iter'].sum()/ms*1e3:.0f} iter/s", g.dims(), flush=True)
    codes = g.solve(); ms = g.last_solve_ms()
    print(f"   second solve {ms:.1f} ms -> {ia['iter'].sum()/ms*1e3:.0f} iter/s")
    if B == 64:
        for i in range(4):
            from eicos_amd.problem_io import Values
            o = OracleSolver(pat, Values(d['Gpr'][i], d['Apr'][i], d['c'][i], d['h'][i], d['b'][i])); oc = o.solve(); oi = o.info()
            print(f"   inst {i}: oracle exit={oc} it={oi['iter']} pcost={oi['pcost']:+.10e} | gpu exit={codes[i]} it={ia['iter'][i]} pcost={ia['pcost'][i]:+.10e} dx={np.abs(o.x()-g.solution()[i]).max():.2e}")
    g.close()
