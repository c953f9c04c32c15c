"""Dev script (GPU): randomised sequences of updateData calls through every host-pointer path, against a mirror.

usage: python tools/dev/r5_api_stress.py [sequences] [seed]
Handle A (and a two-shard multi handle M on {0, 0}) receive a random sequence of `update` calls: random instance sub-ranges, random
subsets of the array groups ((Gpr, h, Apr, b) and / or c; None = keep), from pageable arrays, from library-pinned arrays, from arrays the
caller registered.  Handle R receives the SAME sequence through the device-pointer entry point (rows uploaded with torch) -- a kept group
is un-equilibrated and re-equilibrated like the reference does (src/eicos.cpp:389-404, 302-374), so only the same sequence gives the same
bits, not one update with the final values.  After every sequence all three solve: exit codes, iteration counts, x, y, z, s must be
bit-identical, and the result copies into pageable and pinned memory must agree.
"""
import os, sys
# (the single handle of 600 runs three workgroups per CU -- no dense apex -- its shards of 300 two per CU WITH one: different rounding, 1e-13 on x.
#  This tool compares bits across the two, so the apex is switched off for both; its own test is test_dense_apex_agrees_with_the_level_schedule)
os.environ["EICOS_EXPERIMENT"] = "1"; os.environ["EICOS_APEX"] = "0"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import eicos_amd
from eicos_amd.generate import feasible_batch
from eicos_amd.problem_io import read_problem

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
pat, sets = read_problem(os.path.join(ROOT, "tests", "golden", "MPC02.epb"))
B = 600  # 600 x 124 KB: five bounce chunks of ~ 135 instances, the last one short; two shards of 300 keep the single handle's launch shape
        # (256 threads: a different workgroup size rounds differently -- 1e-13 on x -- and would hide a real difference)
keys = ("Gpr", "Apr", "c", "h", "b")
pool = [feasible_batch(pat, sets[0], 1000 * v, B, seed=100 + v) for v in range(3)]  # three feasible data sets to draw rows from
A = eicos_amd.BatchSolver(pat, B)
R = eicos_amd.BatchSolver(pat, B)
M = eicos_amd.MultiBatchSolver(pat, B, [0, 0])
paths_seen = {}
bad = 0
print("launch shapes: A", {k: A.dims()[k] for k in ("threads_per_block", "resident_blocks")}, " M shard 0", {k: M.shard_dims(0)[k] for k in ("threads_per_block", "resident_blocks")})
for q in range(nseq):
    mirror = {k: pool[0][k].copy() for k in keys}
    A.update(*[mirror[k] for k in keys]); M.update(*[mirror[k] for k in keys])
    dev0 = {k: torch.from_numpy(mirror[k]).cuda() for k in keys}
    R.update_device(*[dev0[k].data_ptr() if dev0[k].numel() else 0 for k in keys])
    nops = int(rng.integers(3, 9))
    log = []
    for op in range(nops):
        first = int(rng.integers(0, B)); count = int(rng.integers(1, B - first + 1))
        if rng.random() < 0.25: first, count = 0, B
        src = pool[int(rng.integers(0, 3))]
        giveGA, givec = rng.random() < 0.6, rng.random() < 0.6   # (G, h, A, b come from ONE data set: the mixture stays feasible)
        if not (giveGA or givec): givec = True
        given = {"Gpr": giveGA, "h": giveGA, "Apr": giveGA, "b": giveGA, "c": givec}
        kind = ("pageable", "pinned", "registered")[int(rng.integers(0, 3))]
        arrs, keep = {}, []
        for k in keys:
            if not given[k]: arrs[k] = None; continue
            rows = np.ascontiguousarray(src[k][first:first + count])
            if kind == "pinned":
                pa = eicos_amd.PinnedArray(rows.shape); pa.a[...] = rows; keep.append(pa); arrs[k] = pa.a
            elif kind == "registered":
                own = rows.copy(); eicos_amd.host_register(own); keep.append(own); arrs[k] = own
            else:
                arrs[k] = rows.copy()
            mirror[k][first:first + count] = rows
        dv = {k: (torch.from_numpy(np.ascontiguousarray(src[k][first:first + count])).cuda() if given[k] else None) for k in keys}
        R.update_device(*[(dv[k].data_ptr() if (dv[k] is not None and dv[k].numel()) else 0) for k in keys], first=first, count=count)
        torch.cuda.synchronize()
        A.update(arrs["Gpr"], arrs["Apr"], arrs["c"], arrs["h"], arrs["b"], first=first, count=count)
        p = A.last_update_path(); paths_seen[p] = paths_seen.get(p, 0) + 1
        M.update(arrs["Gpr"], arrs["Apr"], arrs["c"], arrs["h"], arrs["b"], first=first, count=count)
        for k in keys:  # the arrays are the caller's again on return
            if arrs[k] is not None: arrs[k][...] = np.nan
        if kind == "registered":
            for own in keep: eicos_amd.host_unregister(own)
        for pa in keep:
            if isinstance(pa, eicos_amd.PinnedArray): pa.close()
        log.append((first, count, kind, "".join(k[0] for k in keys if given[k])))
    cR = R.solve(); cA = A.solve(); cM = M.solve()
    xR, (yR, zR, sR), iR = R.solution(), R.duals(), R.info_arrays()
    ok = True
    for name, S, c in (("A", A, cA), ("M", M, cM)):
        x, (y, z, s), ia = S.solution(), S.duals(), S.info_arrays()
        same = (np.array_equal(c, cR) and np.array_equal(x, xR) and np.array_equal(y, yR) and np.array_equal(z, zR) and np.array_equal(s, sR)
                and np.array_equal(ia["iter"], iR["iter"]) and np.array_equal(ia["n_ldlsolve"], iR["n_ldlsolve"]))
        if not same:
            ok = False
            print(f"sequence {q}: handle {name} DIFFERS from the mirror (codes equal {np.array_equal(c, cR)}, iterations equal {np.array_equal(ia['iter'], iR['iter'])}, "
                  f"max |x - x_R| {np.nanmax(np.abs(x - xR)):.3e}); ops {log}")
    px = eicos_amd.PinnedArray((B, pat.n))
    if not np.array_equal(A.solution_into(px.a), xR): ok = False; print(f"sequence {q}: pinned result copy differs")
    px.close()
    bad += not ok
    print(f"sequence {q}: {nops} ops {log} -> {'ok' if ok else 'MISMATCH'}; optimal {int((cR == 0).sum())}/{B}", flush=True)
print(f"{nseq} sequences, {bad} with differences; update paths taken by handle A: {paths_seen}")
A.close(); R.close(); M.close()
sys.exit(1 if bad else 0)
