"""Dev script (not a pytest file): time the batched solve for the current library / env knobs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb, BatchSolver
from eicos_amd.generate import feasible_batch
name = sys.argv[1] if len(sys.argv) > 1 else "MPC02"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if name == 'dense-front':
    from eicos_amd.generate import dense_front_pattern
    pat, base = dense_front_pattern(2000, 32, 64); sets = [base]
else:
    pat, sets = read_epb(f'tests/golden/{name}.epb')
d = feasible_batch(pat, sets[0], 0, min(B, 256))
tile = lambda a: np.tile(a, ((B + a.shape[0] - 1) // a.shape[0], 1))[:B]
g = BatchSolver(pat, B)
g.update(tile(d['Gpr']), tile(d['Apr']), tile(d['c']), tile(d['h']), tile(d['b']))
ms = []
for r in range(reps):
    codes = g.solve(); ms.append(g.last_solve_ms())
ia = g.info_arrays(); dm = g.dims()
tag = f"lib={os.path.basename(os.environ.get('EICOS_AMD_LIB','default'))} T={dm['threads_per_block']} lds={dm['lds_bytes']} resident={dm['resident_blocks']} ldsres={dm.get('lds_resident')} path={dm.get('factor_path')}"
print(f"{name} B={B} {tag}: ms={min(ms):.2f} (all {['%.1f'%m for m in ms]}) iters={ia['iter'].sum()} ok={(codes==0).sum()} -> {ia['iter'].sum()/min(ms)*1e3:.0f} iter/s  pcost0={ia['pcost'][0]:.10e}", flush=True)

if B <= dm['resident_blocks']:
    tr = g.debug_trace(0)[-1]
    print("   phase us (inst 0): factor %.0f ldl(excl fwd) %.0f fwd %.0f kkt-resid %.0f kkt-post %.0f resid-stage %.0f total %.0f" % (tr[0], tr[1], tr[5], tr[2], tr[3], tr[4], tr[6]), " per iter:", ["%.0f" % (v / max(1, ia['iter'][0])) for v in (tr[0], tr[1], tr[5], tr[2], tr[3], tr[4], tr[6])], "nsolve", ia['n_ldlsolve'][0], "nfactor", ia['n_factor'][0])
    nf = max(1, ia['n_factor'][0])
    print("   factor us per call: [8] level 0 / hybrid tile part %.1f  [9] phase A %.1f  [10] barriers %.1f  [11] phase B %.1f  (of %.1f)" % (tr[7] / nf, tr[8] / nf, tr[9] / nf, tr[10] / nf, tr[0] / nf))
