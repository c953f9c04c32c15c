"""Dev (GPU): bit-level fingerprint of the solver's outputs on a fixed set of problems -- run before and after a change that must
not alter a single bit (fused passes, layout changes): python tools/dev/hash_outputs.py out.json ; diff the two files."""
import hashlib, json, os, sys
os.environ["EICOS_EXPERIMENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import numpy as np
import eicos_amd
from eicos_amd.generate import perturbed_batch, feasible_batch, dense_front_pattern, random_socp_pattern, mpc_soc_variant, SEED

out = {}
def fp(*arrs):
    h = hashlib.sha256()
    for a in arrs: h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]
def run(tag, pat, d, B, env=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve().copy(); ia = g.info_arrays(); y, z, s = g.duals()
    out[tag] = dict(codes=fp(codes), iters=fp(ia["iter"]), x=fp(g.solution()), yzs=fp(y, z, s), pcost=fp(ia["pcost"]), n_ldl=int(ia["n_ldlsolve"].sum()) if "n_ldlsolve" in ia else -1,
                    it=int(ia["iter"].sum()), code_list=sorted(set(int(c) for c in codes)))
    g.close()
    for k in (env or {}): os.environ.pop(k)
    print(tag, out[tag], flush=True)
names = ["MPC02", "lp_afiro", "lp_bandm", "lp_agg2", "lp_25fv47", "issue98", "infeasible1", "update_data", "unboundedMaxSqrt", "lp_beaconfd"]
for name in names:
    try: pat, sets = eicos_amd.read_problem(f"tests/golden/{name}.epb")
    except Exception as e: print(name, e); continue
    B = 8
    d = perturbed_batch(pat, sets[0], 0, B, SEED)
    run(name, pat, d, B)
    if name in ("MPC02", "lp_bandm", "issue98"):
        run(name + "/nlds1", pat, d, B, {"EICOS_NLDS": "1", "EICOS_DUAL": "0"})
        run(name + "/nlds0", pat, d, B, {"EICOS_NLDS": "0"})
        run(name + "/nlds2", pat, d, B, {"EICOS_NLDS": "2", "EICOS_DUAL": "0"})
pat, sets = eicos_amd.read_problem("tests/golden/MPC02.epb")
d = perturbed_batch(pat, sets[0], 0, 600, SEED); run("MPC02/b600", pat, d, 600)
spat = mpc_soc_variant(pat, sets[0]); d = perturbed_batch(spat, sets[0], 0, 8, SEED); run("MPC02-soc", spat, d, 8)
run("MPC02-soc/nlds1", spat, d, 8, {"EICOS_NLDS": "1", "EICOS_DUAL": "0"})
pat, base = dense_front_pattern(n=150, k=4, d=40); run("dense_front", pat, feasible_batch(pat, base, 0, 6), 6)
run("dense_front/tiles0", pat, feasible_batch(pat, base, 0, 6), 6, {"EICOS_TILES": "0"})
for seed in range(12):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 120)); p = int(rng.integers(0, n // 2)); l = int(rng.integers(0, 80)); q = [int(rng.choice([2, 3, 4, 7, 12, 33])) for _ in range(int(rng.integers(0, 6)))]
    if l + sum(q) == 0: l = 3
    pat, base = random_socp_pattern(n, p, l, q, density=0.2, seed=seed)
    run(f"rand{seed}", pat, feasible_batch(pat, base, 0, 3, seed=seed), 3)
    if seed % 3 == 0: run(f"rand{seed}/nlds1", pat, feasible_batch(pat, base, 0, 3, seed=seed), 3, {"EICOS_NLDS": "1", "EICOS_DUAL": "0", "EICOS_LDSRES": "0"})
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
