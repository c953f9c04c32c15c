"""Dev script: oracle (CPU baseline) thread scaling on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eicos_amd import read_epb
from eicos_amd.generate import feasible_batch
from oracle import oracle as orc
pat, sets = read_epb('tests/golden/MPC02.epb')
d = feasible_batch(pat, sets[0], 0, 512)
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)
for nt in (1, 8, 32, 64, 128, 256):
    if nt > len(os.sched_getaffinity(0)): break
    nb = min(512, max(16, 4 * nt))
    r = orc.batch_solve(pat, d['Gpr'][:nb], d['Apr'][:nb], d['c'][:nb], d['h'][:nb], d['b'][:nb], nt)
    w = r['seconds'] + r['update_seconds']
    print(nt, 'threads', nb, 'instances: wall', round(w, 3), 'iter/s', round(r['iters'].sum() / w), 'per thread', round(r['iters'].sum() / w / nt), flush=True)
