"""Dev script: perturbed LPnetlib batch, GPU vs oracle exit codes / iterations per instance."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from eicos_amd.generate import perturbed_batch
from eicos_amd.problem_io import Values
from oracle.oracle import OracleSolver
name = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pat, sets = read_epb(f'tests/golden/{name}.epb')
d = perturbed_batch(pat, sets[0], 0, B)
g = BatchSolver(pat, B); g.update(d['Gpr'], d['Apr'], d['c'], d['h'], d['b']); codes = g.solve(); ia = g.info_arrays()
oc, oi, op = [], [], []
for i in range(B):
    o = OracleSolver(pat, Values(d['Gpr'][i], d['Apr'][i], d['c'][i], d['h'][i], d['b'][i])); oc.append(o.solve()); inf = o.info(); oi.append(inf['iter']); op.append(inf['pcost'])
oc = np.array(oc); oi = np.array(oi); op = np.array(op)
print(name, "gpu codes", dict(zip(*np.unique(codes, return_counts=True))), "oracle codes", dict(zip(*np.unique(oc, return_counts=True))))
print(" code mismatch:", int((codes != oc).sum()), " iter mismatch:", int((ia['iter'] != oi).sum()), " max |diter|", int(np.abs(ia['iter'] - oi).max()))
both = (codes == 0) & (oc == 0)
if both.any(): print(" pcost rel diff (both optimal) max", np.max(np.abs(ia['pcost'][both] - op[both]) / np.maximum(1, np.abs(op[both]))))
bad = np.where(codes != oc)[0][:6]
for i in bad: print("  inst", i, "gpu", codes[i], ia['iter'][i], ia['pcost'][i], ia['pres'][i], ia['dres'][i], ia['gap'][i], "| oracle", oc[i], oi[i], op[i])
