"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Tolerances (fp64): same exit code; iteration count within +-1;
|pcost - pcost_ref| <= 1e-8 max(1,|pcost_ref|) for OPTIMAL instances; x to 1e-8 relative
(||x - x_ref||_inf <= 1e-8 max(1, ||x_ref||_inf), north_star) on every instance of the full-size batches whose iteration count equals
the oracle's (all of them on MPC02 / MPC-SOC / dense-front: _all_instances_against_the_oracle, measured maxima in its docstring);
small random SOCPs with non-unique or ill-conditioned optima are held to 1e-6 / 1e-7 plus objective + residuals; pres/dres below the
solver tolerance."""
import numpy as np
import pytest

from conftest import ALL_FIXTURES, load_fixture
import eicos_amd
from eicos_amd.generate import SEED, feasible_batch, mpc_soc_variant, perturbed_batch
from eicos_amd.problem_io import Values
from oracle.oracle import OracleSolver
from test_oracle_golden import in_cone, kkt_residuals

pytestmark = pytest.mark.gpu
PCOST_RTOL = 1e-8
CHAOTIC = {"unboundedMaxSqrt"}
ROUNDING_SET = {2, 12, -2}  # DINF, its reduced-accuracy form, the safeguard exit: what rounding can turn unboundedMaxSqrt's exit into


def rep(v, B):
    return [np.repeat(a[None, :], B, 0) for a in (v.Gpr, v.Apr, v.c, v.h, v.b)]


@pytest.mark.parametrize("name", ALL_FIXTURES)
def test_fixture_matches_oracle(name, expected):
    pat, sets = load_fixture(name)
    o = OracleSolver(pat, sets[0])
    g = eicos_amd.BatchSolver(pat, 3)
    for k, v in enumerate(sets):
        if k:
            o.update(v)
        g.update(*rep(v, 3))
        oc = o.solve(); oi = o.info()
        codes = g.solve(); gi = g.info()
        assert oc in expected[name]["exit_codes"]
        if name in CHAOTIC:
            # Unbounded problem: the iterates diverge (x/tau -> inf), the KKT systems become arbitrarily ill-conditioned and
            # after ~7 passes rounding differences are amplified into different line-search outcomes (see
            # test_unbounded_max_sqrt_exit_distribution_matches_the_oracles below).  The trajectory is pinned while it is well
            # defined; the exit code on the exact data is the one the reference's header asserts (DINF,
            # test/unboundedProblems/unboundedMaxSqrt.h:33) -- since round 3 the expansion columns of a cone are eliminated
            # after the cone's rows (symbolic.cpp), which is what decides this fixture.
            tg, to = g.debug_trace(0), o.trace()
            assert np.allclose(tg[:6, :11], to[:6, :11], rtol=1e-7, atol=1e-12)
            # the outcome on the exact data is decided by rounding (the distribution test below: DINF on ~47 % of 1e-16
            # perturbations on the GPU, ~60 % on the oracle), so a different compiler / box may legitimately land on the safeguard
            # exit: the accepted set is ROUNDING_SET; today's toolchain gives the header's DINF, and a deviation is reported.
            assert len(set(codes)) == 1 and set(codes) <= ROUNDING_SET and oc == 2, (name, codes, oc)
            if codes[0] != 2:
                import warnings
                warnings.warn(f"unboundedMaxSqrt: GPU exit {codes[0]} on the exact data (reference header asserts 2; rounding-determined)")
            continue
        assert list(codes) == [oc] * 3, (name, codes, oc)
        for i in range(3):
            assert gi[i]["exitcode"] == oc
            if oc in (0, 10):
                assert abs(gi[i]["iter"] - oi["iter"]) <= 1
                assert abs(gi[i]["pcost"] - oi["pcost"]) <= PCOST_RTOL * max(1.0, abs(oi["pcost"])), (name, gi[i]["pcost"], oi["pcost"])
                assert gi[i]["pres"] < 1e-8 and gi[i]["dres"] < 1e-8
        if oc in (0, 10) and pat.n:
            x = g.solution(); y, z, s = g.duals()
            rp, re, rd, gap = kkt_residuals(pat, v, x[0], y[0], z[0], s[0])
            assert max(rp, re, rd) < 1e-6 and abs(gap) < 1e-6 * max(1.0, abs(oi["pcost"]))
            assert np.array_equal(x[0], x[1]) and np.array_equal(x[1], x[2])  # same input -> same bits
    g.close(); o.close()


def _check_batch(pat, d, B, n_oracle, x_rtol=1e-8):
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve()
    ia = g.info_arrays()
    x = g.solution(); y, z, s = g.duals()
    assert np.all(codes == 0), np.unique(codes, return_counts=True)
    assert np.all(ia["pres"] < 1e-8) and np.all(ia["dres"] < 1e-8)
    for i in np.linspace(0, B - 1, n_oracle).astype(int):
        v = Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i])
        o = OracleSolver(pat, v)
        assert o.solve() == 0
        oi = o.info()
        assert abs(ia["iter"][i] - oi["iter"]) <= 1
        assert abs(ia["pcost"][i] - oi["pcost"]) <= PCOST_RTOL * max(1.0, abs(oi["pcost"]))
        if ia["iter"][i] == oi["iter"]:
            assert np.abs(x[i] - o.x()).max() <= x_rtol * max(1.0, np.abs(o.x()).max())
        rp, re, rd, gap = kkt_residuals(pat, v, x[i], y[i], z[i], s[i])
        assert max(rp, re, rd) < 1e-7
        assert in_cone(pat, s[i], 1e-7) and in_cone(pat, z[i], 1e-7)
        o.close()
    g.close()
    return ia


def test_trajectory_matches_oracle_iteration_by_iteration():
    # stronger than end-point parity: every pass of the main loop (pcost, dcost, gap, pres, dres, kap/tau, mu,
    # step, sigma, tau, kap) against the oracle's history on well-posed fixtures with LP and SOC cones
    for name, rtol in (("lp_afiro", 1e-6), ("update_data", 1e-6), ("issue98", 1e-5), ("MPC02", 1e-5)):
        pat, sets = load_fixture(name)
        o = OracleSolver(pat, sets[0]); o.solve(); to = o.trace()
        g = eicos_amd.BatchSolver(pat, 1); g.update(*rep(sets[0], 1)); g.solve()
        tg = g.debug_trace(0, o.info()["iter"])
        assert tg.shape == to.shape, (name, tg.shape, to.shape)
        scale = np.maximum(np.abs(to[:, :11]), 1e-9)
        # late iterations sit at the 1e-9..1e-12 noise floor of the residual norms: compare relative to the
        # solver tolerance there
        err = np.abs(tg[:, :11] - to[:, :11]) / np.maximum(scale, 1e-8)
        assert err[:-2].max() < rtol * 1e2 and err[: max(3, len(err) // 2)].max() < rtol, (name, err.max(axis=1))
        g.close(); o.close()


def test_mpc_batch_lp():
    pat, sets = load_fixture("MPC02")
    _check_batch(pat, feasible_batch(pat, sets[0], 0, 96), 96, 6)


def test_mpc_batch_soc():
    pat, sets = load_fixture("MPC02")
    spat = mpc_soc_variant(pat, sets[0])
    assert spat.ncones == 332 and spat.l == 3000
    _check_batch(spat, feasible_batch(spat, sets[0], 0, 48), 48, 4, x_rtol=1e-8)


def test_big_cone_wavefront_path():
    # one 40-dimensional cone (>= CONE_BIG) + LP rows on a small dense problem
    rng = np.random.default_rng(5)
    n, l, d = 12, 6, 40
    m = l + d
    Gd = rng.standard_normal((m, n)) / 3
    from scipy.sparse import csc_matrix
    G = csc_matrix(Gd)
    from eicos_amd.problem_io import Pattern
    pat = Pattern(n, m, 0, l, np.array([d], np.int32), G.indptr.astype(np.int32), G.indices.astype(np.int32),
                  np.zeros(n + 1, np.int32), np.zeros(0, np.int32))
    base = Values(G.data.copy(), np.zeros(0), np.zeros(n), np.zeros(m), np.zeros(0))
    _check_batch(pat, feasible_batch(pat, base, 0, 8), 8, 8, x_rtol=1e-7)


def dense_front_pattern(n, k, d, seed=3):
    """SURVEY.md 8d config 5 in small: k cones of dim d, each a dense d x d block of G on d consecutive
    variables, neighbouring blocks overlapping so that G has full column rank."""
    from scipy.sparse import csc_matrix
    from eicos_amd.problem_io import Pattern
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(k):
        c0 = (i * (n - d)) // max(1, k - 1)
        B = rng.standard_normal((d, d)) / 8
        for r in range(d):
            for c in range(d):
                rows.append(i * d + r); cols.append(c0 + c); vals.append(B[r, c])
    G = csc_matrix((vals, (rows, cols)), shape=(k * d, n)); G.sum_duplicates(); G.sort_indices()
    pat = Pattern(n, k * d, 0, 0, np.full(k, d, np.int32), G.indptr.astype(np.int32), G.indices.astype(np.int32),
                  np.zeros(n + 1, np.int32), np.zeros(0, np.int32))
    return pat, Values(G.data.copy(), np.zeros(0), np.zeros(n), np.zeros(k * d), np.zeros(0))


def test_dense_front_socp():
    # BASELINE.json config 4 shape (dense cone blocks, big cones -> wavefront-per-cone path, deep tree)
    pat, base = dense_front_pattern(n=150, k=4, d=40)
    assert pat.n <= (pat.ncones - 1) * 40 + 40  # blocks cover every column
    _check_batch(pat, feasible_batch(pat, base, 0, 6), 6, 6, x_rtol=1e-6)


@pytest.mark.parametrize("env", [{"EICOS_NLDS": "0"}, {"EICOS_NLDS": "1"}, {"EICOS_NLDS": "2"}, {"EICOS_THREADS": "128"},
                                 {"EICOS_THREADS": "256"}, {"EICOS_THREADS": "512"}, {"EICOS_IDX16": "0"}, {"EICOS_IDX16": "0", "EICOS_NLDS": "0"},
                                 {"EICOS_TILES": "1"}, {"EICOS_TILES": "1", "EICOS_NLDS": "0"}, {"EICOS_TILES": "1", "EICOS_THREADS": "128", "EICOS_NLDS": "2"},
                                 {"EICOS_TILES": "1", "EICOS_THREADS": "256", "EICOS_NLDS": "1"},
                                 {"EICOS_NLDS": "1", "EICOS_DUAL": "0"}, {"EICOS_NLDS": "1", "EICOS_DUAL": "0", "EICOS_TILES": "2", "EICOS_THREADS": "256"},
                                 {"EICOS_W2": "0", "EICOS_THREADS": "256"}, {"EICOS_W2": "1", "EICOS_THREADS": "256", "EICOS_NLDS": "0"},
                                 {"EICOS_FAC_DEFER": "0"}, {"EICOS_FAC_DEFER": "1", "EICOS_TILES": "2"}, {"EICOS_FAC_DEFER": "1", "EICOS_IDX16": "0", "EICOS_TILES": "0"},
                                 {"EICOS_FAC_DEFER": "1", "EICOS_LDSRES": "0", "EICOS_THREADS": "128", "EICOS_NLDS": "1"},
                                 {"EICOS_LDSRES": "0"}, {"EICOS_LDSRES": "0", "EICOS_THREADS": "128", "EICOS_TILES": "0"},
                                 {"EICOS_TILES": "1", "EICOS_GTILES": "2"}, {"EICOS_TILES": "1", "EICOS_GTILES": "2", "EICOS_DUAL": "0", "EICOS_THREADS": "256"},
                                 {"EICOS_TILES": "1", "EICOS_GTILES": "0"},
                                 {"EICOS_TILES": "0"}, {"EICOS_TILES": "2", "EICOS_NLDS": "0"}, {"EICOS_TILES": "2", "EICOS_THREADS": "128"}, {"EICOS_TILES": "2", "EICOS_THREADS": "512", "EICOS_IDX16": "0"}])
def test_every_kernel_variant_matches_oracle(env, monkeypatch):
    # the launch shape is chosen per pattern/batch; force each template instantiation (KKT vectors in LDS
    # or in the workspace slab, 128/256/512 threads, 16-/32-bit gather indices, and the tile (dense-front) factor/solve
    # path, which is normally taken only when L is dense) through an LP, an SOC and an infeasible fixture
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B = 2
    for name in ("lp_bandm", "issue98", "infeasible1", "update_data"):
        pat, sets = load_fixture(name)
        o = OracleSolver(pat, sets[0]); oc = o.solve(); oi = o.info()
        g = eicos_amd.BatchSolver(pat, B); g.update(*rep(sets[0], B))
        codes = g.solve(); gi = g.info()
        assert list(codes) == [oc] * B, (env, name, codes, oc)
        for i in range(B):
            assert abs(gi[i]["iter"] - oi["iter"]) <= 1
            if oc == 0:
                assert abs(gi[i]["pcost"] - oi["pcost"]) <= PCOST_RTOL * max(1.0, abs(oi["pcost"]))
        g.close(); o.close()


@pytest.mark.parametrize("name", ["lp_bandm", "MPC02", "issue98", "lp_25fv47"])
def test_deferred_l_factorisation_is_bit_identical_to_the_stored_l_one(name, monkeypatch):
    # deferred L: a pair's L[j,k] is formed on the fly as U[j,k] * (1/D[k]) -- the same product, rounded the same way, as the
    # stored entry of the other form -- so the two forms may not differ in a single bit of the iterates
    pat, sets = load_fixture(name)
    B = 3
    d = perturbed_batch(pat, sets[0], 0, B, SEED)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("EICOS_FAC_DEFER", flag)
        g = eicos_amd.BatchSolver(pat, B)
        g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve().copy(); ia = g.info_arrays()
        y, z, s_ = g.duals()
        out.append((codes, ia["iter"].copy(), g.solution().copy(), z.copy()))
        g.close()
    for a_, b_ in zip(out[0], out[1]):
        assert np.array_equal(a_, b_)


@pytest.mark.parametrize("env", [{}, {"EICOS_NLDS": "0"}, {"EICOS_NLDS": "2"}, {"EICOS_THREADS": "512"}, {"EICOS_DUAL": "0", "EICOS_NLDS": "1"},
                                 {"EICOS_THREADS": "256", "EICOS_W2": "0"}, {"EICOS_IDX16": "0"}])
def test_dense_apex_agrees_with_the_level_schedule(env, monkeypatch):
    # the last levels of the elimination tree (MPC02: levels 10..20, 63 nodes) swept as a dense triangular system by one wavefront
    # (the apex, reference ldlt.solve src/eicos.cpp:1477,1599) against the same handle with the apex off: a column-oriented
    # substitution rounds differently from the row sums of the level schedule, so not the bits -- but the same exit codes and iteration
    # counts on every instance, refinement counts within one or two solves, and x to 1e-9; in every launch shape that carries an apex: the LDS image (one and two vectors,
    # dual solves at this batch), the fallback from the images in the workspace slab (no LDS vector), 512 threads, 32-bit indices
    pat, sets = load_fixture("MPC02")
    B = 48
    d = feasible_batch(pat, sets[0], 0, B)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("EICOS_APEX", flag)
        g = eicos_amd.BatchSolver(pat, B)
        g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve().copy(); ia = g.info_arrays()
        out.append((codes, ia["iter"].copy(), ia["n_ldlsolve"].copy(), g.solution().copy()))
        g.close()
    assert np.all(out[0][0] == 0) and np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    # (refinement steps stop on an error norm: the different rounding moves a few instances by one LDL solve out of ~ 86)
    assert np.max(np.abs(out[0][2] - out[1][2])) <= 2 and abs(int(out[0][2].sum()) - int(out[1][2].sum())) <= 0.01 * out[0][2].sum()
    xa, xb = out[0][3], out[1][3]
    assert not np.array_equal(xa, xb)  # (the apex was really on: different rounding)
    assert np.max(np.abs(xa - xb) / np.maximum(1.0, np.max(np.abs(xa), axis=1, keepdims=True))) < 1e-9


def test_dense_apex_in_the_lds_resident_build(monkeypatch):
    # lp_afiro (dim_K 129: levels 1..9 of its 10, 59 nodes, are the apex; 128 threads, slabs AND the apex image in LDS): against the same
    # build with the apex off -- equal exit codes and iteration counts, x to 1e-8 (perturbed data, |x| = 500) -- on the Netlib data and a perturbed batch; a batch
    # beyond one instance per CU takes the HBM-slab kernel, which carries no apex (the set-up is repeated without it)
    from eicos_amd.generate import perturbed_batch
    pat, sets = load_fixture("lp_afiro")
    d = perturbed_batch(pat, sets[0], 0, 64, seed=11)
    d["c"][0], d["h"][0], d["b"][0] = sets[0].c, sets[0].h, sets[0].b
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("EICOS_APEX", flag)
        g = eicos_amd.BatchSolver(pat, 64)
        assert g.dims()["lds_resident"] == 1
        g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve().copy(); ia = g.info_arrays()
        out.append((codes, ia["iter"].copy(), g.solution().copy(), ia["pcost"].copy()))
        g.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and abs(out[1][3][0] - (-464.753142857)) < 1e-5
    xa, xb = out[0][2], out[1][2]
    assert not np.array_equal(xa, xb) and np.max(np.abs(xa - xb) / np.maximum(1.0, np.max(np.abs(xa), axis=1, keepdims=True))) < 1e-8  # (measured 2e-9)
    monkeypatch.setenv("EICOS_APEX", "1")
    g = eicos_amd.BatchSolver(pat, 1024)  # (more instances than LDS-resident workgroups: 128 threads on the slabs in HBM, no apex)
    assert g.dims()["lds_resident"] == 0 and g.dims()["threads_per_block"] == 128
    dd = perturbed_batch(pat, sets[0], 0, 1024, seed=11)
    g.update(dd["Gpr"], dd["Apr"], dd["c"], dd["h"], dd["b"]); codes = g.solve()
    assert np.array_equal(codes[:64], out[0][0]) and np.array_equal(g.solution()[:64], out[0][2])  # (= the apex-off bits of the same instances)
    g.close()


def test_kernel_build_reported_for_the_handle(monkeypatch):
    # which compilation of k_solve a handle launches is part of its description: one workgroup per CU with the factor operand array in
    # the idle LDS ("u-in-lds") when it fits; else 256-thread workgroups at <= 2 per CU take the 256-VGPR build ("w2"), small patterns the
    # LDS-resident one, everything else the default one; EICOS_UBL=0 / EICOS_W2=0 forbid the first two
    pat, sets = load_fixture("MPC02")
    g = eicos_amd.BatchSolver(pat, 4); d = g.dims(); kb = g.kernel_build(); g.close()
    assert kb in ("u-in-lds", "w2" if d["threads_per_block"] == 256 else "default") and not d["lds_resident"]
    monkeypatch.setenv("EICOS_UBL", "0")
    monkeypatch.setenv("EICOS_THREADS", "256")
    g = eicos_amd.BatchSolver(pat, 4); assert g.kernel_build() == "w2"; g.close()
    monkeypatch.setenv("EICOS_W2", "0")
    g = eicos_amd.BatchSolver(pat, 4); assert g.kernel_build() == "default"; g.close()
    monkeypatch.delenv("EICOS_W2"); monkeypatch.delenv("EICOS_THREADS"); monkeypatch.delenv("EICOS_UBL")
    g = eicos_amd.BatchSolver(pat, 1024); assert g.kernel_build() == "w2"; g.close()  # (the headline launch: two per CU)
    pat, sets = load_fixture("lp_afiro")
    g = eicos_amd.BatchSolver(pat, 4); d = g.dims(); assert (g.kernel_build() == "lds-resident") == bool(d["lds_resident"]); g.close()
    pat, sets = load_fixture("lp_bandm")
    g = eicos_amd.BatchSolver(pat, 256); assert g.kernel_build() == "u-in-lds" and g.dims()["threads_per_block"] == 512; g.close()


@pytest.mark.parametrize("name,B,threads", [("lp_bandm", 96, None), ("lp_adlittle", 64, None), ("lp_agg", 48, None), ("lp_beaconfd", 32, None), ("lp_blend", 32, "512"),
                                            ("update_data", 8, "256"), ("issue98", 4, "256"), ("infeasible1", 4, "512"), ("unboundedLP1", 4, "256")])
def test_u_in_lds_build_is_bit_identical_to_the_hbm_slab_kernels(name, B, threads, monkeypatch):
    # at one workgroup per CU the factor operand array U = L.*D lives in the idle LDS when it fits (kernels_ubl*.hip):
    # the same programs in the same order of operations -- every bit of the result equal to the build that keeps U in the workspace slab
    # (LP hybrid / apex / MPC-size scalar / SOC / infeasible patterns; perturbed Netlib batches hold ill-posed instances with long solves)
    pat, sets = load_fixture(name)
    if name.startswith("lp_"):
        d = perturbed_batch(pat, sets[0], 0, B)
    else:  # (the small SOC / infeasible / unbounded fixtures: their own data, workgroup size forced so that they leave the 128-thread LDS-resident build)
        d = dict(zip(("Gpr", "Apr", "c", "h", "b"), rep(sets[0], B)))
    if threads:
        monkeypatch.setenv("EICOS_THREADS", threads)
    out = []
    for ubl in ("1", "0"):
        monkeypatch.setenv("EICOS_UBL", ubl)
        g = eicos_amd.BatchSolver(pat, B)
        g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
        codes = g.solve(); ia = g.info_arrays(); x = g.solution(); y, z, s_ = g.duals()
        codes2 = g.solve()  # (a second solve on the same handle: the LDS copy is rebuilt by every factorisation)
        out.append((g.kernel_build(), codes, ia["iter"], ia["n_ldlsolve"], x, y, z, s_, ia["pcost"], codes2, g.solution()))
        g.close()
    assert out[0][0] == "u-in-lds" and out[1][0] != "u-in-lds", (out[0][0], out[1][0])
    for a, b in zip(out[0][1:], out[1][1:]):
        assert np.array_equal(a, b)


def test_g_tile_products_match_the_ell_products_on_dense_fronts(monkeypatch):
    # dense-front pattern: G goes to 16 x 16 tiles (one pass for G x and G' z); same iteration counts and solutions as with
    # the sliced-ELL products (the sums are associated differently, so not bit for bit); m and n not multiples of 16,
    # overlapping column windows (columns met by 4..8 tiles)
    pat, base = dense_front_pattern(n=150, k=4, d=40)
    d = feasible_batch(pat, base, 0, 6)
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("EICOS_GTILES", flag)
        g = eicos_amd.BatchSolver(pat, 6)
        g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve().copy(); ia = g.info_arrays()
        out.append((codes, ia["iter"].copy(), ia["pcost"].copy(), g.solution().copy(), g.dims()["inst_bytes"]))
        g.close()
    assert out[0][4] < out[1][4]                      # (tiles replace the two ELL copies of G in the instance slab)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.allclose(out[0][2], out[1][2], rtol=1e-9) and np.allclose(out[0][3], out[1][3], rtol=1e-7, atol=1e-9)


def test_lds_resident_variant_is_bit_identical_to_the_hbm_slab_kernel(monkeypatch):
    # small patterns whose slabs fit LDS run the LDS-resident build of k_solve (same code, slab pointers in LDS): same
    # arithmetic in the same order, so every output must be bit-identical to the kernel that works on the slabs in HBM;
    # LP, SOC and an infeasible fixture + a perturbed batch with different iteration counts per instance
    # (the dense apex, which among the 128-thread kernels only the LDS-resident build carries, is off: it rounds differently by design --
    # its own comparison is test_dense_apex_in_the_lds_resident_build)
    from eicos_amd.generate import perturbed_batch
    monkeypatch.setenv("EICOS_APEX", "0")
    cases = []
    for name in ("lp_afiro", "issue98", "infeasible1", "update_data"):
        pat, sets = load_fixture(name)
        cases.append((name, pat, rep(sets[0], 3), 3))
    pat, sets = load_fixture("lp_afiro")
    d = perturbed_batch(pat, sets[0], 0, 40, seed=5)
    cases.append(("lp_afiro/perturbed", pat, (d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]), 40))
    for name, pat, data, B in cases:
        out = []
        for flag in ("1", "0"):
            monkeypatch.setenv("EICOS_LDSRES", flag)
            g = eicos_amd.BatchSolver(pat, B)
            res = g.dims()["lds_resident"]
            assert res == 0 if flag == "0" else (res == 1 or not name.startswith("lp_afiro")), (name, g.dims())
            g.update(*data); codes = g.solve().copy(); ia = g.info_arrays()
            y, z, s_ = g.duals()
            out.append((codes, g.solution().copy(), y.copy(), z.copy(), s_.copy(), ia["iter"].copy(), ia["pcost"].copy(), ia["n_ldlsolve"].copy()))
            codes2 = g.solve().copy()  # second solve: persistent per-instance state went through the copy-out / copy-in
            out[-1] += (codes2, g.solution().copy())
            g.close()
        for a_, b_ in zip(out[0], out[1]):
            assert np.array_equal(a_, b_, equal_nan=True), name


@pytest.mark.parametrize("seed,n,p,l,q", [(1, 12, 3, 6, [4, 3]), (2, 20, 5, 0, [5, 5, 5]), (3, 9, 0, 4, [6]),
                                          (4, 30, 8, 10, [3] * 6), (5, 16, 4, 5, [1, 2, 7]), (6, 60, 10, 20, [40, 33, 2])])
def test_random_socp_with_equalities(seed, n, p, l, q):
    # random sparse patterns with equality rows, LP rows and cones of mixed size (incl. dimension 1 and >= 32)
    from eicos_amd.generate import random_socp_pattern
    pat, base = random_socp_pattern(n, p, l, q, seed=seed)
    _check_batch(pat, feasible_batch(pat, base, 0, 5, seed=100 + seed), 5, 5, x_rtol=1e-6)


def test_batch_larger_than_the_resident_grid_uses_queue_and_history_order(monkeypatch):
    # more instances than resident workgroups: instances are pulled from the kernel's queue, and from the second
    # solve on in longest-first order of the previous solve's LDL-solve counts; neither may change any result
    # (the small comparison batch runs the LDS-resident build, which would carry a dense apex -- different rounding by design: off here)
    monkeypatch.setenv("EICOS_APEX", "0")
    pat, sets = load_fixture("lp_afiro")
    g = eicos_amd.BatchSolver(pat, 8)
    resident = g.dims()["resident_blocks"]
    g.close()
    B = 3000
    assert resident >= 8
    d = feasible_batch(pat, sets[0], 0, B, seed=77)
    g = eicos_amd.BatchSolver(pat, B)
    assert g.dims()["resident_blocks"] < B  # otherwise this test does not exercise the queue
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    c1 = g.solve().copy(); x1 = g.solution().copy(); it1 = g.info_arrays()["iter"].copy()
    c2 = g.solve().copy(); x2 = g.solution().copy(); it2 = g.info_arrays()["iter"].copy()  # ordered by history now
    assert np.array_equal(c1, c2) and np.array_equal(it1, it2) and np.array_equal(x1, x2)
    assert np.all(c1 == 0) and len(np.unique(it1)) > 1  # heterogeneous work, or the order would be trivial
    g.close()
    # same instances in a batch that fits the grid (identity assignment): bit-identical results
    sub = np.linspace(0, B - 1, 64).astype(int)
    g = eicos_amd.BatchSolver(pat, len(sub))
    g.update(d["Gpr"][sub], d["Apr"][sub], d["c"][sub], d["h"][sub], d["b"][sub])
    g.solve()
    assert np.array_equal(g.solution(), x1[sub]) and np.array_equal(g.info_arrays()["iter"], it1[sub])
    g.close()
    for i in sub[::16]:
        o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        assert o.solve() == 0 and o.info()["iter"] == it1[i]
        assert np.abs(x1[i] - o.x()).max() <= 1e-7 * max(1.0, np.abs(o.x()).max())
        o.close()


def test_lpnetlib_perturbed_batch():
    pat, sets = load_fixture("lp_blend")
    d = perturbed_batch(pat, sets[0], 0, 32)
    g = eicos_amd.BatchSolver(pat, 32)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays()
    assert np.all(codes == 0)
    for i in (0, 7, 31):
        o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        assert o.solve() == 0
        assert abs(ia["pcost"][i] - o.info()["pcost"]) <= PCOST_RTOL * max(1.0, abs(o.info()["pcost"]))
    g.close()


def test_lpnetlib_perturbed_ill_posed_instances_match_oracle():
    # perturbing c and h of a Netlib LP makes part of the batch hard (close-to-optimal, unbounded, maxit exits):
    # the GPU must take the same exit as the oracle on every instance, not just on the easy ones
    pat, sets = load_fixture("lp_agg2")
    B = 12
    d = perturbed_batch(pat, sets[0], 0, B)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays()
    for i in range(B):
        o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        oc = o.solve(); oi = o.info()
        assert codes[i] == oc, (i, codes[i], oc)
        assert abs(ia["iter"][i] - oi["iter"]) <= 1
        if oc == 0:  # badly scaled LP (|pcost| ~ 2e7): objective agreement at the solver's own relgap tolerance
            assert abs(ia["pcost"][i] - oi["pcost"]) <= 5e-8 * max(1.0, abs(oi["pcost"]))
        o.close()
    g.close()


@pytest.mark.parametrize("name", ["update_data", "issue98", "MPC02", "dense-front"])
def test_the_three_update_kernels_agree_bit_for_bit(name, monkeypatch):
    # updateData (equilibration, src/eicos.cpp:302-374,2053-2082) exists three times: thread-per-column (EICOS_UPDATE_LDS=0),
    # entry-parallel with values and maxima in LDS (1, default when they fit), entry-parallel with the values streamed in
    # place in the slab and only the maxima in LDS (2, default for patterns like the dense-front config).  Same arithmetic
    # in the same order per entry -> the equilibrated matrices, the scaled vectors and therefore every solve output are
    # bit-identical; second round: c, h only (A, G, b kept -> the un-equilibrate / re-equilibrate path)
    if name == "dense-front":
        pat, base = dense_front_pattern(n=150, k=4, d=40)
        d = feasible_batch(pat, base, 0, 3)
        data = (d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    else:
        pat, sets = load_fixture(name)
        data = rep(sets[0], 3)
    outs = []
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("EICOS_UPDATE_LDS", mode)
        g = eicos_amd.BatchSolver(pat, 3)
        g.update(*data)
        r, c, v1 = g.debug_kkt(2)
        codes = g.solve().copy(); x1 = g.solution().copy(); it1 = g.info_arrays()["iter"].copy()
        g.update(c=data[2] * 1.01, h=data[3] + 0.01 * np.abs(data[3]))
        r, c, v2 = g.debug_kkt(1)
        codes2 = g.solve().copy(); x2 = g.solution().copy()
        outs.append((v1.copy(), codes, x1, it1, v2.copy(), codes2, x2))
        g.close()
    for o in outs[1:]:
        for a_, b_ in zip(outs[0], o):
            assert np.array_equal(a_, b_, equal_nan=True), name


def test_update_keep_semantics_and_ranges():
    # NULL groups keep the previous data (reference updateData(double*...), src/eicos.cpp:2053-2082);
    # partial ranges only touch their instances.
    pat, sets = load_fixture("update_data")
    v1, v2 = sets
    g = eicos_amd.BatchSolver(pat, 4)
    g.update(*rep(v1, 4))
    g.update(c=np.repeat(v2.c[None], 2, 0), first=2, count=2)           # only c of instances 2,3
    o = OracleSolver(pat, v1); o.solve(); p1 = o.info()["pcost"]
    o.update(Values(None, None, v2.c, None, None)) if False else None
    o2 = OracleSolver(pat, Values(v1.Gpr, v1.Apr, v2.c, v1.h, v1.b)); c2 = o2.solve(); p2 = o2.info()["pcost"]
    codes = g.solve(); ia = g.info_arrays()
    assert codes[0] == codes[1] == 0 and codes[2] == codes[3] == c2
    assert abs(ia["pcost"][0] - p1) < 1e-8 * max(1, abs(p1)) and abs(ia["pcost"][3] - p2) < 1e-8 * max(1, abs(p2))
    g.close()


@pytest.mark.parametrize("name,soc,tiles", [("lp_afiro", False, False), ("issue98", False, False), ("MPC02", False, False), ("MPC02", True, False),
                                            ("lp_afiro", False, True), ("issue98", False, True), ("dense-front", False, True)])
def test_ldl_factor_against_dense(name, soc, tiles, monkeypatch):
    # numeric LDL' of one instance against the matrix it factorises: the componentwise backward error of an LDL'
    # without pivoting, |L D L' - P K P'| <= c eps |L||D||L'| (+ 1e-14 max|K|; a plain norm bound does not hold at the
    # last IPM pass of a degenerate SOC problem, where pivots of 1e-10 sit next to entries of order 1), with K
    # assembled independently of the factor program from the instance's own values (equilibrated A/G, scaling block
    # of the last iteration, +-delta) in the reference's layout (src/eicos.cpp:1734-1890) -- SURVEY.md section 7 step 3
    from scipy.sparse import coo_matrix, csc_matrix, diags, identity
    if tiles:  # the tile (dense-front, MFMA) factorisation; forced on the sparse fixtures, chosen by itself for dense fronts
        monkeypatch.setenv("EICOS_TILES", "1")
    if name == "dense-front":
        pat, base = dense_front_pattern(n=150, k=4, d=40)
        d = feasible_batch(pat, base, 0, 1)
        sets = [Values(d["Gpr"][0], d["Apr"][0], d["c"][0], d["h"][0], d["b"][0])]
    else:
        pat, sets = load_fixture(name)
    if soc:
        pat = mpc_soc_variant(pat, sets[0])
        d = feasible_batch(pat, sets[0], 0, 1)
        vals = [d[k] for k in ("Gpr", "Apr", "c", "h", "b")]
    else:
        vals = rep(sets[0], 1)
    g = eicos_amd.BatchSolver(pat, 1)
    g.update(*vals)
    assert g.solve()[0] in (0, 10)
    D, U = g.debug_factor(0)          # factorises K as it stands: scalings of the last pass
    r, c, v = g.debug_kkt(0)
    perm, Lp, Li = g.debug_pattern()
    N = len(D)
    assert len(v) == g.dims()["nnzK"] and np.all(r <= c)
    iperm = np.empty(N, np.int64); iperm[perm] = np.arange(N)
    off = r != c
    K = coo_matrix((np.concatenate([v, v[off]]), (np.concatenate([iperm[r], iperm[c[off]]]), np.concatenate([iperm[c], iperm[r[off]]]))),
                   shape=(N, N)).tocsc()
    cols = np.repeat(np.arange(N), np.diff(Lp))
    L = csc_matrix((U / D[cols], (Li, cols)), shape=(N, N)) + identity(N, format="csc")
    R = abs(L @ diags(D) @ L.T - K)
    bound = 64 * np.finfo(float).eps * (abs(L) @ diags(np.abs(D)) @ abs(L).T)
    excess = (R - bound).tocoo()
    # (the tile path meets the same componentwise bound: its off-diagonal tiles are triangular solves by substitution, not
    # products with an explicit inverse of the diagonal tile)
    assert excess.data.max() <= 1e-14 * np.abs(K.data).max(), (name, tiles, excess.data.max(), np.abs(K.data).max())
    if name not in ("issue98", "dense-front"):  # well-scaled instances also meet the plain norm bound
        assert R.tocoo().data.max() <= 1e-12 * np.abs(K.data).max()
    # quasi-definite signs: + for the x block and the u-expansion of every cone, - elsewhere
    pos = np.zeros(N, bool); pos[: pat.n] = True
    k = pat.n + pat.p + pat.l
    for q in pat.q:
        pos[k + q + 1] = True; k += q + 2
    assert (D[pos[perm]] > 0).all() and (D[~pos[perm]] < 0).all()
    g.close()


def test_update_scalings_failure_modes_match_oracle():
    # updateScalings returns early when a cone leaves the cone (ref :428-431) or fails c2byu02 - d > 0 (ref :460-463);
    # solve() ignores the return value and updateKKTScalings (ref :1162) writes whatever the cone structs hold.  In the
    # second case that is the NEW eta^2 and q with the OLD d1, u0, u1, v1.  Both sides run their own scaling stage on
    # the same (s, z): first an interior pair (sets the "old" state), then pairs that trigger each failure.
    pat, sets = load_fixture("issue98")  # l = 6, one cone of dimension 5
    o = OracleSolver(pat, sets[0]); g = eicos_amd.BatchSolver(pat, 1); g.update(*rep(sets[0], 1))
    rng = np.random.default_rng(0)

    def cone_pt(d, eps):
        t = rng.standard_normal(d - 1)
        return np.concatenate([[np.linalg.norm(t) * (1 + eps)], t])

    def both(s, z):
        ok, Vo = o.debug_scalings(s, z)
        ran, Vg = g.debug_scalings(s, z)
        assert ran
        scale = np.maximum(1.0, np.abs(Vo))
        assert np.all(np.abs(Vg - Vo) <= 1e-9 * scale), (ok, np.abs(Vg - Vo).max())
        return ok, Vo

    lp = lambda: rng.uniform(0.5, 2, pat.l)
    ok, V0 = both(np.concatenate([lp(), cone_pt(5, 0.5)]), np.concatenate([lp(), cone_pt(5, 0.7)]))
    assert ok
    # late failure: both points strictly inside (sres, zres > 0) but so close to the boundary that c2byu02 - d <= 0
    late = 0
    for _ in range(4000):
        s = np.concatenate([lp(), cone_pt(5, 10.0 ** rng.uniform(-16, -6))]); z = np.concatenate([lp(), cone_pt(5, 10.0 ** rng.uniform(-16, -6))])
        if min(s[6] ** 2 - s[7:] @ s[7:], z[6] ** 2 - z[7:] @ z[7:]) <= 0:
            continue
        ok, V = both(s, z)
        if not ok:
            late += 1
            assert not np.allclose(V[pat.l:], V0[pat.l:])  # the cone block DID change (new eta^2, q)
            ok2, V0 = both(np.concatenate([lp(), cone_pt(5, 0.3)]), np.concatenate([lp(), cone_pt(5, 0.9)]))  # fresh old state
            assert ok2
            if late >= 3:
                break
    assert late >= 1
    # early failure: s outside the cone -> the cone's block keeps its previous values, the LP part is new
    s = np.concatenate([lp(), cone_pt(5, -0.1)]); z = np.concatenate([lp(), cone_pt(5, 0.4)])
    ok, V = both(s, z)
    assert not ok and np.array_equal(V[pat.l:], V0[pat.l:]) and not np.array_equal(V[: pat.l], V0[: pat.l])
    g.close(); o.close()


def test_full_size_batch_properties():
    # BASELINE.json config[1] size (batch 1024, MPC02 pattern): size-independent properties only
    pat, sets = load_fixture("MPC02")
    B = 1024
    d = feasible_batch(pat, sets[0], 0, B)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays()
    assert np.all(codes == 0)
    assert np.all(ia["pres"] < 1e-8) and np.all(ia["dres"] < 1e-8) and np.all((ia["gap"] < 1e-8) | (ia["relgap"] < 1e-8))
    x = g.solution(); y, z, s = g.duals()
    # weak duality / complementarity from raw data, all instances at once
    pc = np.einsum("ij,ij->i", d["c"], x)
    dc = -np.einsum("ij,ij->i", d["h"], z) - np.einsum("ij,ij->i", d["b"], y)
    assert np.all(np.abs(pc - dc) <= 1e-7 * np.maximum(1, np.abs(pc)))
    assert np.all(s > -1e-9) and np.all(z > -1e-9)
    # idempotence: solving again from the same data gives the same bits (cold start, ref App. A.7)
    x1 = x.copy(); g.solve()
    assert np.array_equal(g.solution(), x1)
    g.close()
    # ... and ALL 1024 instances against the oracle (1-2 s of CPU): exit code, iteration count +-1, pcost, x
    _all_instances_against_the_oracle(pat, d, codes, ia, x, tag="MPC02 b1024")


def _all_instances_against_the_oracle(pat, d, codes, ia, x, x_rtol=1e-8, tag=None):
    """Every instance against the oracle: exit code, iteration count +-1, pcost to 1e-8, and x to north_star's "within 1e-8 relative"
    (||x - x_ref||_inf <= 1e-8 max(1, ||x_ref||_inf)) on the instances whose iteration counts agree.  Measured on the GPU box, round 5
    (tools/dev/r5_xerr.py -> profiles/r05_log_xerr.log; iteration counts equal on every instance of all four workloads): MPC02 batch
    1024 1.5e-12, batch 4096 4.6e-12, MPC-SOC batch 1024 7.9e-9, dense-front batch 512 3.8e-9.  An instance that stops one pass
    earlier or later sits at a different point of the central path -- an IPM stopped at 1e-8 residuals pins x to about the gap tolerance
    there -- so that group (empty on these workloads) is held to 1e-6 and must stay below 10 % of the batch.  The measured maxima are
    printed (pytest -s) and appended to gpurun_out/parity_xerr.jsonl."""
    import json
    import os
    from oracle import oracle as orc
    r = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], len(os.sched_getaffinity(0)), want_x=True)
    assert np.array_equal(r["exitcodes"], codes)
    it_o, it_g = r["iters"].astype(int), ia["iter"].astype(int)
    assert np.all(np.abs(it_o - it_g) <= 1), np.flatnonzero(np.abs(it_o - it_g) > 1)
    assert np.all(np.abs(ia["pcost"] - r["pcost"]) <= PCOST_RTOL * np.maximum(1.0, np.abs(r["pcost"])))
    same = it_o == it_g
    assert same.sum() >= 0.9 * len(codes)
    err = np.abs(x - r["x"]).max(axis=1) / np.maximum(1.0, np.abs(r["x"]).max(axis=1))
    rec = {"workload": tag or f"n={pat.n} m={pat.m} cones={pat.ncones} batch={len(codes)}", "iters_equal": int(same.sum()), "iters_pm1": int((~same).sum()),
           "xerr_equal_max": float(err[same].max()), "xerr_pm1_max": float(err[~same].max()) if (~same).any() else None,
           "pcost_rel_max": float((np.abs(ia["pcost"] - r["pcost"]) / np.maximum(1.0, np.abs(r["pcost"]))).max())}
    # How far do two CPU builds of the SAME oracle source differ on these instances?  -O2 (portable: the checker) against -O2 -march=native
    # (g++ contracts a * b + c into FMAs there: different rounding, same algorithm).  That spread is the conditioning of the problem at the
    # solver's own stopping tolerance -- the yardstick the GPU's distance has to be read against.
    rn = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], len(os.sched_getaffinity(0)), want_x=True, native=True)
    if rn["native"]:
        same_n = rn["iters"].astype(int) == it_o
        err_n = np.abs(rn["x"] - r["x"]).max(axis=1) / np.maximum(1.0, np.abs(r["x"]).max(axis=1))
        rec.update({"oracle_native_vs_portable_xerr_max": float(err_n[same_n].max()), "oracle_native_iters_equal": int(same_n.sum()),
                    "gpu_vs_native_xerr_max": float((np.abs(x - rn["x"]).max(axis=1) / np.maximum(1.0, np.abs(rn["x"]).max(axis=1)))[same_n & same].max())})
    print("x parity:", json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_xerr.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    assert np.all(err[same] <= x_rtol), (rec, np.flatnonzero(same & (err > x_rtol))[:8])
    assert np.all(err[~same] <= 1e-6), rec


def test_config4_dense_front_full_size():
    # BASELINE.json configs[4] at its stated size: n = 2000, 32 cones of dimension 64, batch 512
    from eicos_amd.generate import dense_front_pattern as full_dense_front
    pat, base = full_dense_front(2000, 32, 64)
    assert (pat.n, pat.m, pat.ncones, pat.l, pat.nnzG) == (2000, 2048, 32, 0, 32 * 64 * 64)
    B = 512
    d = feasible_batch(pat, base, 0, B)
    g = eicos_amd.BatchSolver(pat, B)
    assert g.dims()["dim_K"] == 2000 + 2048 + 64 and g.dims()["nnzK"] == 139248
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays()
    x = g.solution(); y, z, s = g.duals()
    # size-independent properties on all 512 instances
    assert np.all(codes == 0), np.unique(codes, return_counts=True)
    assert np.all(ia["pres"] < 1e-8) and np.all(ia["dres"] < 1e-8) and np.all((ia["gap"] < 1e-8) | (ia["relgap"] < 1e-8))
    pc = np.einsum("ij,ij->i", d["c"], x); dc = -np.einsum("ij,ij->i", d["h"], z)
    assert np.all(np.abs(pc - dc) <= 1e-7 * np.maximum(1, np.abs(pc)))          # strong duality from raw data
    zc, sc = z.reshape(B, 32, 64), s.reshape(B, 32, 64)
    assert np.all(zc[:, :, 0] - np.linalg.norm(zc[:, :, 1:], axis=2) > -1e-7) and np.all(sc[:, :, 0] - np.linalg.norm(sc[:, :, 1:], axis=2) > -1e-7)
    G = __import__("scipy.sparse", fromlist=["csc_matrix"]).csc_matrix((base.Gpr, pat.Gir, pat.Gjc), shape=(pat.m, pat.n))
    assert np.abs((G @ x.T).T + s - d["h"]).max() <= 1e-6 * max(1.0, np.abs(d["h"]).max())  # primal feasibility, all instances
    x1 = x.copy(); g.solve()
    assert np.array_equal(g.solution(), x1)                                        # idempotent re-solve, same bits
    # oracle parity on ALL 512 instances (~3 s on 16 cores): exit code, iteration count +-1, pcost to 1e-8, x
    _all_instances_against_the_oracle(pat, d, codes, ia, x, tag="dense-front b512")
    g.close()


NETLIB = ["lp_afiro", "lp_adlittle", "lp_blend", "lp_bandm", "lp_beaconfd", "lp_agg", "lp_agg2", "lp_agg3", "lp_bnl1", "lp_25fv47"]


def _perturbed_outcomes(pat, d, i, eps, T, cores):
    """(exit code, iterations) of T copies of instance i with ALL data perturbed by eps (relative), on the oracle and on the GPU."""
    from oracle import oracle as orc
    rng = np.random.default_rng(1000 + int(i))
    dd = [d[k][i][None, :] * (1 + eps * rng.uniform(-1, 1, (T,) + d[k][i].shape)) for k in ("Gpr", "Apr", "c", "h", "b")]
    r = orc.batch_solve(pat, *dd, cores)
    g = eicos_amd.BatchSolver(pat, T); g.update(*dd); gc = g.solve(); gi = g.info_arrays(); g.close()
    return set(zip(r["exitcodes"].tolist(), r["iters"].tolist())), set(zip(gc.tolist(), gi["iter"].tolist()))


def _within(res, outcomes):
    """res = (code, iterations) lies inside the spread of `outcomes`: same exit code, iteration count within its range +-1."""
    its = [it for (c, it) in outcomes if c == res[0]]
    return bool(its) and min(its) - 1 <= res[1] <= max(its) + 1


@pytest.mark.parametrize("name", NETLIB)
def test_config3_lpnetlib_batch256(name):
    # BASELINE.json configs[3] at its stated size: batch 256 of perturbed instances per Netlib pattern (SURVEY.md 8d
    # config 4), ALL TEN patterns, ALL 256 instances against the oracle: same exit code, iteration count +-1.
    # Perturbation makes a few instances ill-posed: they stall for tens of passes and the pass in which the stall ends
    # (full-accuracy exit, reduced-accuracy exit, iteration limit) is decided by rounding.  An instance may differ from the
    # oracle ONLY if that is shown right here: under relative perturbations of its data by 1e-16 .. 1e-14 the ORACLE's own
    # outcome must flip (>= 2 distinct outcomes) and the two sides must land in each other's spread; anything else is a bug.
    import os
    from oracle import oracle as orc
    pat, sets = load_fixture(name)
    B = 256
    cores = len(os.sched_getaffinity(0))
    d = perturbed_batch(pat, sets[0], 0, B)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays(); x = g.solution(); y, z, s = g.duals()
    g.close()
    assert codes[0] == 0
    opt = codes == 0
    # per-pattern floor: the OPTIMAL count measured in round 3 (the others are ill-posed perturbed instances that end
    # in a reduced-accuracy exit or at the iteration limit on BOTH sides) minus the 6 instances the flip argument below may move
    floor = {"lp_afiro": 256, "lp_adlittle": 256, "lp_blend": 256, "lp_bandm": 256, "lp_bnl1": 256, "lp_beaconfd": 247,
             "lp_agg": 224, "lp_agg2": 166, "lp_agg3": 161, "lp_25fv47": 166}[name] - 6
    assert opt.sum() >= floor, (name, int(opt.sum()), floor)
    assert np.all(ia["pres"][opt] < 1e-8) and np.all(ia["dres"][opt] < 1e-8)
    pc = np.einsum("ij,ij->i", d["c"], x); dc = -np.einsum("ij,ij->i", d["h"], z) - np.einsum("ij,ij->i", d["b"], y)
    assert np.all(np.abs(pc - dc)[opt] <= 1e-6 * np.maximum(1, np.abs(pc))[opt])
    assert np.all(s[opt] > -1e-9) and np.all(z[opt] > -1e-9)
    r = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], cores)
    it_o, it_g = r["iters"].astype(int), ia["iter"].astype(int)
    differs = np.flatnonzero((r["exitcodes"] != codes) | (np.abs(it_o - it_g) > 1))
    assert len(differs) <= 6, (name, differs)  # (at most 4 of 256 on any pattern when this was written)
    for i in differs:
        res_g, res_o = (int(codes[i]), int(it_g[i])), (int(r["exitcodes"][i]), int(it_o[i]))
        spread_o, spread_g = set(), set()
        for eps in (1e-16, 1e-15, 1e-14):
            so, sg = _perturbed_outcomes(pat, d, i, eps, 32, cores)
            spread_o |= so; spread_g |= sg
        assert len(spread_o) >= 2, (name, i, res_g, res_o, "the oracle does not flip under perturbation: a real difference")
        assert _within(res_g, spread_o) or _within(res_o, spread_g), (name, i, res_g, res_o, sorted(spread_o), sorted(spread_g))
    same = np.setdiff1d(np.arange(B), differs)
    okb = same[(codes[same] == 0) & (it_o[same] == it_g[same])]
    # |pcost| ~ 4e7 on lp_agg: agreement at the solver's own relative-gap tolerance when both sides stop at the same pass
    assert np.all(np.abs(ia["pcost"][okb] - r["pcost"][okb]) <= 2e-7 * np.maximum(1.0, np.abs(r["pcost"][okb]))), name
    # ... and a looser bound where the two sides stop one pass apart (both OPTIMAL: both within the relative-gap tolerance of the optimum)
    ok1 = same[(codes[same] == 0) & (np.abs(it_o[same] - it_g[same]) == 1)]
    assert np.all(np.abs(ia["pcost"][ok1] - r["pcost"][ok1]) <= 5e-7 * np.maximum(1.0, np.abs(r["pcost"][ok1]))), name


def test_unbounded_max_sqrt_exit_distribution_matches_the_oracles():
    # VERDICT r2 item 1a.  300 copies of unboundedMaxSqrt perturbed by 1e-16 (the batch of
    # tests/test_oracle_golden.py::test_unbounded_max_sqrt_exit_is_rounding_determined): the oracle returns DINF on ~60 %
    # and the safeguard exit NUMERICS on the rest.  The GPU must show the same two outcomes in comparable proportion
    # (measured: 140 x DINF / 160 x NUMERICS vs the oracle's 182 / 118; before the cone-aware elimination order of round 3:
    # 40 / 260, and 0 / 300 with both expansion columns eliminated first) -- and DINF on the exact data.
    import os
    from oracle import oracle as orc
    pat, sets = load_fixture("unboundedMaxSqrt")
    v = sets[0]
    B = 300
    rng = np.random.default_rng(1)
    Gp, cp, hp = [], [], []
    for _ in range(B):
        pert = lambda a: a * (1 + 1e-16 * rng.uniform(-1, 1, a.shape))
        Gp.append(pert(v.Gpr)); cp.append(pert(v.c)); hp.append(pert(v.h))
    rp = lambda a: np.repeat(a[None, :], B, 0)
    dd = (np.array(Gp), rp(v.Apr), np.array(cp), np.array(hp), rp(v.b))
    r = orc.batch_solve(pat, *dd, len(os.sched_getaffinity(0)))
    g = eicos_amd.BatchSolver(pat, B); g.update(*dd); codes = g.solve(); g.close()
    assert set(codes) <= {2, 12, -2} and set(r["exitcodes"]) <= {2, 12, -2}
    n_g, n_o = int((codes == 2).sum()), int((r["exitcodes"] == 2).sum())
    assert 0.25 * B <= n_g <= 0.85 * B and 0.25 * B <= n_o <= 0.85 * B, (n_g, n_o)
    g = eicos_amd.BatchSolver(pat, 1); g.update(*[a[None, :] for a in (v.Gpr, v.Apr, v.c, v.h, v.b)])
    assert g.solve()[0] in ROUNDING_SET  # (2 with today's toolchain = test/unboundedProblems/unboundedMaxSqrt.h:33; the distribution above is the property)
    g.close()


def test_ordering_dependent_fatal_seeds_are_pinned():
    # VERDICT r3 item 5.  The reference keeps static regularisation only, so a pivot that cancels to exactly 0.0 ends the solve with
    # `fatal` (src/eicos.cpp:901-905, 1166-1170) -- and WHICH pivots cancel depends on the elimination order (Eigen's AMD order in the
    # reference, plain minimum degree in the oracle, minimum degree with multiple elimination + level / tile renumbering here).  The two
    # round-3 campaigns (9000 random patterns each) found 29 patterns where exactly one side hits such a pivot on one instance, in both
    # directions, almost all with a cone of dimension 2.  tests/golden/ordering_fatal_seeds.json (tools/dev/r4_fatal_seeds.py) pins, per
    # seed, the exit codes of both sides on today's default paths; this test re-checks them and shows that every difference is a
    # `fatal` on exactly one side and that with the dynamic-regularisation extension (2e-7, 1e-13) BOTH sides solve every instance
    # and agree.  Nothing here says which side the reference's AMD order would be on: that needs Eigen (absent).
    import json, os
    from conftest import ROOT, fuzz_case_r3
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ordering_fatal_seeds.json")))
    assert len(fx["cases"]) == 29
    n_gpu_fatal = n_orc_fatal = 0
    for rec in fx["cases"]:
        pat, d = fuzz_case_r3(rec["seed"], rec["scale"])
        assert (pat.n, pat.p, pat.l, [int(v) for v in pat.q]) == (rec["n"], rec["p"], rec["l"], rec["q"])
        for tag, dyn in (("static", None), ("dynreg", (2e-7, 1e-13))):
            oc, opc = [], []
            for i in range(3):
                o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
                if dyn:
                    o.set_dynamic_regularization(*dyn)
                oc.append(int(o.solve())); opc.append(o.info()["pcost"]); o.close()
            g = eicos_amd.BatchSolver(pat, 3)
            if dyn:
                g.set_dynamic_regularization(*dyn)
            g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
            gc = [int(c) for c in g.solve()]; gpc = g.info_arrays()["pcost"]; g.close()
            assert oc == rec["oracle_" + tag], (rec["seed"], tag, oc)
            assert gc == rec["gpu_" + tag], (rec["seed"], tag, gc, rec["gpu_" + tag])
            for i in range(3):
                if dyn:  # the extension repairs the cancelling pivot on whichever side has it: both solve, same optimum
                    assert oc[i] in (0, 10) and gc[i] in (0, 10), (rec["seed"], i, oc, gc)
                    assert abs(gpc[i] - opc[i]) <= 1e-6 * max(1.0, abs(opc[i])), (rec["seed"], i, gpc[i], opc[i])
                elif gc[i] != oc[i]:  # static regularisation only: a difference is a fatal on exactly one side
                    assert (gc[i] == -7) != (oc[i] == -7), (rec["seed"], i, oc, gc)
                    n_gpu_fatal += gc[i] == -7; n_orc_fatal += oc[i] == -7
    assert n_orc_fatal >= 15  # (the oracle-side cases are independent of the GPU path: 18 when this was written)


def test_soc_patterns_on_the_unconstrained_order_do_not_depend_on_rounding():
    # the cone-aware elimination order (a cone's expansion columns after its rows: numerically preferable, symbolic.cpp) is
    # taken only where it is free under the cost model; MPC-SOC and the dense-front pattern keep the unconstrained order
    # (eicos_dims.cone_order = 0).  Unlike unboundedMaxSqrt, whose exit is decided by that order, these well-posed patterns must not
    # care: 64 copies of one instance with ALL data perturbed by 1e-14 give the same exit and iteration counts within one pass, on the
    # GPU and on the oracle alike.
    import os
    from oracle import oracle as orc
    from eicos_amd.generate import dense_front_pattern as full_dense_front
    pat0, sets = load_fixture("MPC02")
    cases = [(mpc_soc_variant(pat0), sets[0]), full_dense_front(600, 8, 64)]
    for pat, base in cases:
        d1 = feasible_batch(pat, base, 0, 1)
        T = 64
        rng = np.random.default_rng(7)
        dd = [d1[k][0][None, :] * (1 + 1e-14 * rng.uniform(-1, 1, (T,) + d1[k][0].shape)) for k in ("Gpr", "Apr", "c", "h", "b")]
        g = eicos_amd.BatchSolver(pat, T)
        assert g.dims()["cone_order"] == 0
        g.update(*dd); gc = g.solve(); gi = g.info_arrays()["iter"].astype(int); g.close()
        r = orc.batch_solve(pat, *dd, len(os.sched_getaffinity(0)))
        assert np.all(gc == 0) and np.all(r["exitcodes"] == 0)
        assert gi.max() - gi.min() <= 1 and r["iters"].max() - r["iters"].min() <= 1 and np.all(np.abs(gi - r["iters"].astype(int)) <= 1)


def test_ecos_shim_runs_every_registered_reference_test(tmp_path, expected):
    # N2: the reference's registered tests (test/ecostester.cpp:54-72) driven through the ECOS shim of include/ecos.h
    # (ECOS_setup -> ECOS_solve [-> ECOS_updateData -> ECOS_solve] -> ECOS_cleanup) by a compiled C++ runner; the
    # exit codes are those the reference's test headers assert (tests/golden/expected.json cites each)
    import os, subprocess
    from conftest import ROOT
    exe = str(tmp_path / "ecos_runner")
    lib = os.path.join(ROOT, "eicos_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "ecos_runner.cpp"),
                           "-L", lib, "-leicos_amd", "-Wl,-rpath," + lib, "-o", exe])
    manifest = tmp_path / "manifest.txt"
    with open(manifest, "w") as f:
        for name in ALL_FIXTURES:
            codes = sorted(ROUNDING_SET) if name in CHAOTIC else list(expected[name]["exit_codes"])  # (rounding-determined on that fixture, see above)
            f.write(f"{name} {os.path.join(ROOT, 'tests', 'golden', name + '.epb')} {','.join(str(c) for c in codes)}\n")
    out = subprocess.run([exe, str(manifest)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL TESTS PASSED" in out.stdout and f"Tests run: {len(ALL_FIXTURES)}" in out.stdout, out.stdout


def test_eigen_typed_surface_runs_the_reference_demo_flow(tmp_path):
    # the Eigen-typed Solver(G,A,c,h,b,q) / updateData(G,A,c,h,b) / const VectorXd& solution() surface
    # (reference include/eicos.hpp:138-148,160) in the shape of the reference's src/run.cpp:11-50, compiled against the
    # minimal Eigen stand-in of tests/eigen_standin (Eigen itself is not installed in this image) and run on the GPU
    import os, subprocess
    from conftest import ROOT
    exe = str(tmp_path / "run_eigen")
    lib = os.path.join(ROOT, "eicos_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "tests", "eigen_standin"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "run_eigen_demo.cpp"), "-L", lib, "-leicos_amd", "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "update_data.epb")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pcost -36.2505" in out.stdout and "pcost -20.0115" in out.stdout, out.stdout  # udd_optval1/2 of the reference


def test_cpp_solver_surface_demo(tmp_path):
    # examples/run_demo.cpp = the reference's src/run.cpp flow (ctor -> solve -> updateData -> solve) through
    # include/eicos.hpp; built with the host compiler and run on the GPU
    import os, subprocess
    from conftest import ROOT
    exe = str(tmp_path / "run_demo")
    lib = os.path.join(ROOT, "eicos_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "run_demo.cpp"),
                           "-L", lib, "-leicos_amd", "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "MPC02.epb")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("exit 0") == 2 and "pcost 0.16" in out.stdout, out.stdout


def test_multi_gpu_layer_two_shards_on_one_device_bit_identical(monkeypatch):
    # VERDICT r3 item 4 / SURVEY.md 8b, 8e: multi-GPU lives in the product (eicos_multi_* of include/eicos_amd.h, host C++, one handle
    # and stream per shard).  A single-GPU box can exercise everything but the second device: device_ids = {0, 0} = two shards of 512 on
    # the one GPU, solved concurrently on two streams -- bit-identical to ONE handle on all 1024 MPC02 instances; inputs from host
    # arrays, from device arrays read in place, and through the peer-copy path (forced: hipMemcpyPeerAsync with equal devices)
    pat, sets = load_fixture("MPC02")
    B = 1024
    d = feasible_batch(pat, sets[0], 0, B)
    one = eicos_amd.BatchSolver(pat, B, device=0)
    one.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes1 = one.solve(); ia1 = one.info_arrays(); x1 = one.solution(); y1, z1, s1 = one.duals()
    assert np.all(codes1 == 0)
    # NULL = keep on a sub-range (only c of instances 500..523 re-sent, same values): the kept groups are un-equilibrated and
    # re-equilibrated (reference updateData semantics, src/eicos.cpp:2053-2082), which moves the last bits of those instances
    one.update(None, None, d["c"][500:524], None, None, first=500, count=24)
    one.solve(); x1k = one.solution()
    one.close()
    assert not np.array_equal(x1k[500:524], x1[500:524]) and np.array_equal(np.delete(x1k, np.s_[500:524], 0), np.delete(x1, np.s_[500:524], 0))

    def check(m):
        codes = m.solve(); ia = m.info_arrays(); x = m.solution(); y, z, s = m.duals()
        assert np.array_equal(codes, codes1) and np.array_equal(x, x1) and np.array_equal(y, y1) and np.array_equal(z, z1) and np.array_equal(s, s1)
        for k in ("iter", "pcost", "dcost", "pres", "dres", "gap", "n_factor", "n_ldlsolve", "nitref1", "nitref2", "nitref3"):
            assert np.array_equal(ia[k], ia1[k]), k
        mx, per = m.last_solve_ms()
        assert len(per) == 2 and mx >= max(per) and min(per) > 0  # (mx: the device's span over both shards' launches)

    m = eicos_amd.MultiBatchSolver(pat, B, [0, 0])
    assert m.shards() == [(0, 512, 0), (512, 512, 0)]
    m.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])  # host arrays: every shard stages its own rows, in parallel host threads
    check(m)
    # asynchronous form: both shards' kernels are enqueued before either is waited for
    m.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); m.solve_async(); m.sync()
    assert np.array_equal(m.solution(), x1)
    # NULL = keep, on a sub-range that straddles the shard boundary (instances 500..523 = the end of shard 0 and the start of shard 1)
    m.update(None, None, d["c"][500:524], None, None, first=500, count=24)
    assert np.all(m.solve() == 0) and np.array_equal(m.solution(), x1k)
    m.close()
    # device-resident inputs without torch: plain hipMalloc / hipMemcpy through the HIP runtime the library itself is linked against
    import ctypes
    from eicos_amd.binding import _lib
    hip = _lib()  # (dlsym on the library's handle reaches its dependency libamdhip64: the SAME runtime instance the solver uses)
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    dev = {}
    for k, v in d.items():
        pbuf = ctypes.c_void_p()
        if v.size:
            v = np.ascontiguousarray(v)
            assert hip.hipMalloc(ctypes.byref(pbuf), v.nbytes) == 0
            assert hip.hipMemcpy(pbuf, v.ctypes.data, v.nbytes, 1) == 0  # hipMemcpyHostToDevice
        dev[k] = pbuf.value or 0
    ptr = lambda k: dev[k]
    m = eicos_amd.MultiBatchSolver(pat, B, [0, 0])
    m.update_device(0, ptr("Gpr"), ptr("Apr"), ptr("c"), ptr("h"), ptr("b"))  # inputs resident on GPU 0: read in place by both shards
    check(m)
    assert m.shard_last_update(1)[0] == "none"  # (read in place on the source GPU: the other-GPU code was not involved)
    m.close()
    # the other-GPU paths, forced on this one-GPU box and WITNESSED (the knob is read on every call, and
    # eicos_batch_last_update_path says which path a shard really took): (a) the kernel reads the source GPU's HBM in place (peer
    # access), (b) staged hipMemcpyPeerAsync copies in chunks of 256 through the device staging buffer -- ragged shards 334 + 333 + 333
    monkeypatch.setenv("EICOS_EXPERIMENT", "1"); monkeypatch.setenv("EICOS_MULTI_FORCE_PEER", "1")
    for staged, path in (("0", "peer GPU in place"), ("1", "staged peer copies")):
        monkeypatch.setenv("EICOS_PEER_STAGED", staged)
        m = eicos_amd.MultiBatchSolver(pat, 1000, [0, 0, 0])
        assert m.shards() == [(0, 334, 0), (334, 333, 0), (667, 333, 0)]
        m.update_device(0, ptr("Gpr"), ptr("Apr"), ptr("c"), ptr("h"), ptr("b"), count=1000)
        assert [m.shard_last_update(s_)[0] for s_ in range(3)] == [path] * 3
        codes = m.solve()
        assert np.array_equal(codes, codes1[:1000]) and np.array_equal(m.solution(), x1[:1000]) and np.array_equal(m.info_arrays()["iter"], ia1["iter"][:1000])
        m.close()
    monkeypatch.delenv("EICOS_MULTI_FORCE_PEER"); monkeypatch.delenv("EICOS_PEER_STAGED")
    for pbuf in dev.values():
        if pbuf:
            hip.hipFree(ctypes.c_void_p(pbuf))


def test_host_pointer_update_and_result_paths_are_bit_identical():
    # the reference's real signature is updateData(double *...) / solution() on HOST memory
    # (/root/reference include/eicos.hpp:155-160).  Pageable arrays travel through the pinned double-buffer bounce (the kernel reads the
    # bounce buffer in place over PCIe), pinned arrays (eicos_host_alloc) are read / written in place; both must give the bits of the
    # device-resident path, on the full range, on a sub-range with kept groups, and for results copied into pageable and pinned memory
    import ctypes
    from eicos_amd.binding import _lib
    pat, sets = load_fixture("MPC02")
    B = 600  # 600 x 124 KB = 74 MB: five bounce chunks of 128 instances (the last one short)
    d = feasible_batch(pat, sets[0], 0, B)
    keys = ("Gpr", "Apr", "c", "h", "b")
    hip = _lib()
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    dev = {}
    for k in keys:
        pbuf = ctypes.c_void_p()
        v = np.ascontiguousarray(d[k])
        assert hip.hipMalloc(ctypes.byref(pbuf), max(v.nbytes, 8)) == 0 and hip.hipMemcpy(pbuf, v.ctypes.data, v.nbytes, 1) == 0
        dev[k] = pbuf.value
    g = eicos_amd.BatchSolver(pat, B, device=0)
    g.update_device(*[dev[k] for k in keys])
    codes0 = g.solve(); x0 = g.solution(); ia0 = g.info_arrays()
    assert np.all(codes0 == 0) and g.last_update_path() == "none"
    # a DEVICE pointer handed to the host-pointer entry point is refused (the bounce copy would fault on it), nothing is changed
    rc = hip.eicos_batch_update(g._h, 0, B, ctypes.cast(dev["Gpr"], ctypes.POINTER(ctypes.c_double)), None, None, ctypes.cast(dev["h"], ctypes.POINTER(ctypes.c_double)), None)
    assert rc == -1 and b"device memory" in hip.eicos_last_error()
    # pageable host arrays: the bounce pipeline
    g.update(*[d[k] for k in keys])
    assert g.last_update_path() == "pinned bounce" and g.last_update_ms() > 0
    codes = g.solve()
    assert np.array_equal(codes, codes0) and np.array_equal(g.solution(), x0) and np.array_equal(g.info_arrays()["iter"], ia0["iter"])
    # the caller's arrays are free again on return: scribbling over a copy that was passed must not change anything
    scratch = {k: d[k].copy() for k in keys}
    g.update(*[scratch[k] for k in keys])
    for k in keys:
        scratch[k][...] = np.nan
    assert np.array_equal(g.solve(), codes0) and np.array_equal(g.solution(), x0)
    # pinned host arrays: read in place, x written in place
    pins = {k: eicos_amd.PinnedArray(d[k].shape) for k in keys}
    for k in keys:
        pins[k].a[...] = d[k]
    px = eicos_amd.PinnedArray((B, pat.n))
    g.update(*[pins[k].a for k in keys])
    assert g.last_update_path() == "pinned source in place"
    for k in keys:
        pins[k].a[...] = 0.0  # (the call has waited for the kernel: overwriting the arrays now is allowed, as with the reference)
    assert np.array_equal(g.solve(), codes0) and np.array_equal(g.solution_into(px.a), x0)
    y, z, s_ = g.duals()
    # a sub-range with kept groups through the bounce (c of instances 100..355 only: two chunks would need > 16 MB -- one here) and a
    # short range below one chunk: same bits as the same calls with device pointers
    g.update(None, None, d["c"][100:356], None, None, first=100, count=256)
    g.solve(); xk = g.solution()
    g.update_device(*[dev[k] for k in keys]); g.solve()
    assert np.array_equal(g.solution(), x0)
    g.update_device(0, 0, dev["c"] + 100 * pat.n * 8, 0, 0, first=100, count=256)
    g.solve()
    assert np.array_equal(g.solution(), xk) and not np.array_equal(xk[100:356], x0[100:356])
    # the caller's own arrays pinned IN PLACE (eicos_host_register): the in-place path without an allocation by the library
    own = {k: np.ascontiguousarray(d[k]).copy() for k in keys}
    for k in keys:
        eicos_amd.host_register(own[k])
    g.update(*[own[k] for k in keys])
    assert g.last_update_path() == "pinned source in place"
    assert np.array_equal(g.solve(), codes0) and np.array_equal(g.solution(), x0)
    for k in keys:
        eicos_amd.host_unregister(own[k])
    g.update(*[own[k] for k in keys])  # unregistered again: back on the bounce
    assert g.last_update_path() == "pinned bounce"
    # ... and the same sub-range from a PINNED slice (read in place, NULL groups kept)
    g.update_device(*[dev[k] for k in keys]); g.solve()
    pc = eicos_amd.PinnedArray((256, pat.n)); pc.a[...] = d["c"][100:356]
    g.update(None, None, pc.a, None, None, first=100, count=256)
    assert g.last_update_path() == "pinned source in place"
    g.solve()
    assert np.array_equal(g.solution(), xk)
    pc.close()
    g.close()
    for pa in list(pins.values()) + [px]:
        pa.close()
    for v in dev.values():
        hip.hipFree(ctypes.c_void_p(v))


def test_handles_on_concurrent_host_threads_share_the_copy_pool():
    # the host-pointer pipeline is shared state (one copy pool per process, one creation lock per device): four handles created, updated from
    # pageable arrays, solved and read back on four host threads at once must give the bits of the same calls made one after the other
    import threading
    pat, sets = load_fixture("MPC02")
    B = 160  # 160 x 124 KB = 20 MB per update: two bounce chunks each
    datas = [feasible_batch(pat, sets[0], 1000 * t, B) for t in range(4)]

    def run(t, out):
        g = eicos_amd.BatchSolver(pat, B, device=0)
        d = datas[t]
        for _ in range(2):
            g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
            codes = g.solve()
        out[t] = (codes, g.solution(), g.info_arrays()["iter"], g.last_update_path())
        g.close()

    seq, par = {}, {}
    for t in range(4):
        run(t, seq)
    th = [threading.Thread(target=run, args=(t, par)) for t in range(4)]
    for x_ in th:
        x_.start()
    for x_ in th:
        x_.join()
    for t in range(4):
        assert par[t][3] == "pinned bounce" and np.all(par[t][0] == 0)
        assert np.array_equal(par[t][0], seq[t][0]) and np.array_equal(par[t][1], seq[t][1]) and np.array_equal(par[t][2], seq[t][2])


def test_cpp_batch_solver_over_a_device_list(tmp_path):
    # examples/multi_gpu_demo.cpp: EiCOS::BatchSolver(device_ids) from host C++ (no torch), device list {0, 0} on this box;
    # the program itself compares the sharded run with a single-device run bit for bit
    import os, subprocess
    from conftest import ROOT
    exe = str(tmp_path / "multi_gpu_demo")
    lib = os.path.join(ROOT, "eicos_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multi_gpu_demo.cpp"),
                           "-L", lib, "-leicos_amd", "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "MPC02.epb"), "48", "0,0"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "2 shard(s)" in out.stdout and "1 shard(s)" in out.stdout and "single-device results: bit-identical" in out.stdout and "48 / 48 optimal" in out.stdout, out.stdout
    assert "one-call solve on pinned arrays vs updateData + solve: bit-identical" in out.stdout, out.stdout  # (eicos_multi_update_solve: fused per shard)


def test_ecos_shim_runs_a_reference_style_test(tmp_path):
    # include/ecos.h = drop-in for the reference's test/ecos.h: the shape of a reference test (ECOS_setup ->
    # ECOS_solve -> ECOS_updateData -> ECOS_solve -> ECOS_cleanup, test/updateData/update_data.h) on inline data:
    # min -x1 - x2  s.t.  x1 + x2 <= 1, x >= 0  (optimum -1), then h0 = 3 (optimum -3)
    import os, subprocess
    from conftest import ROOT
    src = tmp_path / "shim.cpp"
    src.write_text(r'''
#include "ecos.h"
#include <cstdio>
int main() {
    idxint Gjc[3] = {0, 2, 4}, Gir[4] = {0, 1, 0, 2};
    pfloat Gpr[4] = {1, -1, 1, -1}, c[2] = {-1, -1}, h[3] = {1, 0, 0};
    pwork *w = ECOS_setup(2, 3, 0, 3, 0, nullptr, 0, Gpr, Gjc, Gir, nullptr, nullptr, nullptr, c, h, nullptr);
    if (!w) { std::printf("setup failed: %s\n", eicos_last_error()); return 1; }
    idxint e1 = ECOS_solve(w);
    eicos_info i1; eicos_info_get(w->h, &i1);
    pfloat h2[3] = {3, 0, 0};
    ECOS_updateData(w, Gpr, nullptr, c, h2, nullptr);
    idxint e2 = ECOS_solve(w);
    eicos_info i2; eicos_info_get(w->h, &i2);
    double x[2]; eicos_solution(w->h, x);
    std::printf("exit %d pcost %.9f | exit %d pcost %.9f x1+x2 %.9f\n", e1, i1.pcost, e2, i2.pcost, x[0] + x[1]);
    ECOS_cleanup(w, 0);
    return (e1 == ECOS_OPTIMAL && e2 == ECOS_OPTIMAL) ? 0 : 2;
}
''')
    exe = str(tmp_path / "shim")
    lib = os.path.join(ROOT, "eicos_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", lib, "-leicos_amd",
                           "-Wl,-rpath," + lib, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    toks = out.stdout.split()
    assert abs(float(toks[3]) + 1.0) < 1e-7 and abs(float(toks[8]) + 3.0) < 1e-7 and abs(float(toks[10]) - 3.0) < 1e-6, out.stdout


def test_large_pattern_uses_slab_vectors_and_32bit_indices():
    # dim_K > 65535: the solve vector does not fit LDS (NLDS = 0, slab vectors, plain barriers), gather indices stay
    # 32-bit, and the factor program's slice table is read from global memory -- the paths small fixtures never take
    from scipy.sparse import csc_matrix, vstack, identity, diags
    from eicos_amd.problem_io import Pattern
    n = 24000
    rng = np.random.default_rng(11)
    band = diags([rng.uniform(0.5, 1.5, n - 1), rng.uniform(0.5, 1.5, n - 2)], [1, 2], shape=(n // 2, n))
    G = csc_matrix(vstack([-identity(n), identity(n), band])); G.sort_indices()
    rows = np.repeat(np.arange(n // 4), 3)
    cols = (4 * np.arange(n // 4))[:, None] + np.arange(3)[None, :]
    A = csc_matrix((rng.uniform(0.5, 1.5, rows.size) * np.tile([1, 0.1, -1], n // 4), (rows, cols.ravel())), shape=(n // 4, n))
    A.sort_indices()
    pat = Pattern(n, G.shape[0], A.shape[0], G.shape[0], np.zeros(0, np.int32), G.indptr.astype(np.int32),
                  G.indices.astype(np.int32), A.indptr.astype(np.int32), A.indices.astype(np.int32))
    base = Values(G.data.copy(), A.data.copy(), np.zeros(n), np.zeros(G.shape[0]), np.zeros(A.shape[0]))
    assert pat.n + pat.p + pat.m > 65535
    g = eicos_amd.BatchSolver(pat, 2)
    assert g.dims()["lds_bytes"] == 0 and g.dims()["dim_K"] > 65535
    g.close()
    _check_batch(pat, feasible_batch(pat, base, 0, 2, seed=5), 2, 1, x_rtol=1e-7)


@pytest.mark.parametrize("soc", [False, True])
def test_warm_start_matches_oracle_and_saves_iterations(soc):
    # N3 (extension, off by default): after updateData with slightly different data the solve starts from the previous
    # solution.  Same rule in the oracle -> same iteration counts and optimum; fewer iterations than the cold start.
    pat, sets = load_fixture("MPC02")
    if soc:
        pat = mpc_soc_variant(pat, sets[0])
    B = 4
    d = feasible_batch(pat, sets[0], 0, B, seed=3)
    rng = np.random.default_rng(0)
    d2 = dict(d)
    d2["c"] = d["c"] * (1 + 0.01 * rng.uniform(-1, 1, d["c"].shape))
    d2["h"] = d["h"] + 0.01 * (1 + np.abs(d["h"])) * rng.uniform(0, 1, d["h"].shape)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    assert np.all(g.solve() == 0)
    cold_first = g.info_arrays()["iter"].copy()
    g.update(d2["Gpr"], d2["Apr"], d2["c"], d2["h"], d2["b"])
    assert np.all(g.solve() == 0)
    cold = g.info_arrays()["iter"].copy(); pc_cold = g.info_arrays()["pcost"].copy()
    # again, warm: solve the first data set, switch the option on, update, solve
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); g.solve()
    g.set_warm_start(0.1)
    g.update(d2["Gpr"], d2["Apr"], d2["c"], d2["h"], d2["b"])
    assert np.all(g.solve() == 0)
    ia = g.info_arrays(); x = g.solution()
    assert np.all(ia["iter"] < cold) and np.all(ia["nitref1"] == 0)
    assert np.all(np.abs(ia["pcost"] - pc_cold) <= 1e-7 * np.maximum(1, np.abs(pc_cold)))
    for i in range(B):
        o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        o.set_warm_start(0.1)
        assert o.solve() == 0 and o.info()["iter"] == cold_first[i]
        o.update(Values(d2["Gpr"][i], d2["Apr"][i], d2["c"][i], d2["h"][i], d2["b"][i]))
        assert o.solve() == 0
        assert abs(o.info()["iter"] - ia["iter"][i]) <= 1, (i, o.info()["iter"], ia["iter"][i])
        assert abs(o.info()["pcost"] - ia["pcost"][i]) <= 1e-8 * max(1.0, abs(o.info()["pcost"]))
        if o.info()["iter"] == ia["iter"][i]:
            assert np.abs(x[i] - o.x()).max() <= 1e-7 * max(1.0, np.abs(o.x()).max())
        o.close()
    # switching it off again restores the cold start bit for bit
    g.set_warm_start(0.0)
    g.update(d2["Gpr"], d2["Apr"], d2["c"], d2["h"], d2["b"]); g.solve()
    assert np.array_equal(g.info_arrays()["iter"], cold)
    g.close()


def test_partly_pinned_extents_take_the_bounce_path_and_reregistering_larger_is_refused():
    # an array is read / written in place only when EVERY byte of its extent is pinned.  A registration of the first
    # rows only, a pointer whose count runs past the registered end, and a second registration of the same base with a larger size
    # (hipHostRegister answers "already registered" and maps nothing new) used to be classified from the first byte alone -> GPU page fault.
    from eicos_amd.binding import _lib
    pat, sets = load_fixture("lp_afiro")
    B = 64
    d = feasible_batch(pat, sets[0], 0, B)
    keys = ("Gpr", "Apr", "c", "h", "b")
    L = _lib()
    g = eicos_amd.BatchSolver(pat, B, device=0)
    g.update(*[d[k] for k in keys]); codes0 = g.solve(); x0 = g.solution()
    big = np.ascontiguousarray(np.tile(d["Gpr"], (400, 1)))  # 400 * 64 rows: several MB, only the first rows get registered
    half = big[: B // 2]
    assert L.eicos_host_register(half.ctypes.data, half.nbytes) == 0
    try:
        # all B rows from the half-registered buffer: the extent is not fully pinned -> bounce path, same bits
        others = {k: np.ascontiguousarray(d[k]).copy() for k in keys if k != "Gpr"}
        for v in others.values():
            if v.size:
                eicos_amd.host_register(v)
        g.update(big[:B], others["Apr"], others["c"], others["h"], others["b"])
        assert g.last_update_path() == "pinned bounce"
        assert np.array_equal(g.solve(), codes0) and np.array_equal(g.solution(), x0)
        # ... and the fully registered half IS read in place
        g.update(big[: B // 2], others["Apr"][: B // 2], others["c"][: B // 2], others["h"][: B // 2], others["b"][: B // 2], first=0, count=B // 2)
        assert g.last_update_path() == "pinned source in place"
        assert np.array_equal(g.solve(), codes0) and np.array_equal(g.solution(), x0)
        for v in others.values():
            if v.size:
                eicos_amd.host_unregister(v)
        # results into a buffer that is pinned at its first byte only: staged, not written past the mapping
        xb = np.zeros((40 * B, pat.n))
        assert L.eicos_host_register(xb.ctypes.data, xb[: B // 2].nbytes) == 0
        assert np.array_equal(g.solution_into(xb[:B]), x0)
        assert L.eicos_host_unregister(xb.ctypes.data) == 0
        # the same base pointer registered again with a LARGER size: either the runtime maps the larger range (then all of it is usable in
        # place) or it answers "already registered" and maps nothing new -- which is refused, not reported as success
        rc = L.eicos_host_register(big.ctypes.data, big.nbytes)
        assert rc == 0 or (rc == -1 and b"SMALLER" in L.eicos_last_error())
        others = {k: np.ascontiguousarray(d[k]).copy() for k in keys if k != "Gpr"}
        for v in others.values():
            if v.size:
                eicos_amd.host_register(v)
        g.update(big[:B], others["Apr"], others["c"], others["h"], others["b"])
        assert g.last_update_path() in (("pinned source in place", "pinned bounce") if rc == 0 else ("pinned bounce",))  # (in place only if every probe of the extent answers "pinned")
        assert np.array_equal(g.solve(), codes0) and np.array_equal(g.solution(), x0)
        for v in others.values():
            if v.size:
                eicos_amd.host_unregister(v)
    finally:
        L.eicos_host_unregister(half.ctypes.data)
    g.close()


def test_arithmetic_profile_one_gives_batch_independent_bits():
    # the plans of a handle (workgroup size, dense apex, single-wavefront tree top) follow the batch size, so the last bits of an
    # instance's result depend on the batch / shard it is solved in.  eicos_set_arithmetic_profile(1) shapes the plans by the pattern alone:
    # the same instances give the same bits alone, inside a batch of 300 (one workgroup per CU), of 700 and of 1100 (two per CU, queue)
    pat, sets = load_fixture("MPC02")
    d = feasible_batch(pat, sets[0], 0, 1100)
    keys = ("Gpr", "Apr", "c", "h", "b")
    try:
        eicos_amd.set_arithmetic_profile(1)
        ref = None
        for B in (6, 300, 700, 1100):
            g = eicos_amd.BatchSolver(pat, B)
            dm = g.dims()
            assert dm["arithmetic_profile"] == 1 and dm["apex_nodes"] == 0 and dm["threads_per_block"] == 256
            g.update(*[d[k][:B] for k in keys])
            codes = g.solve(); ia = g.info_arrays()
            out = (codes[:6], ia["iter"][:6], ia["pcost"][:6], g.solution()[:6], g.duals()[1][:6])
            g.close()
            if ref is None:
                ref = out
            for a, b in zip(ref, out):
                assert np.array_equal(a, b), B
    finally:
        eicos_amd.set_arithmetic_profile(0)
    g = eicos_amd.BatchSolver(pat, 6); dm = g.dims(); g.close()
    assert dm["arithmetic_profile"] == 0


@pytest.mark.parametrize("name,B", [("MPC02", 600), ("lp_afiro", 300), ("lp_bandm", 64), ("update_data", 40), ("MPC02", 40)])
def test_fused_update_solve_is_bit_identical_to_update_then_solve(name, B, monkeypatch):
    # eicos_batch_update_solve: with pinned / registered host arrays the solve kernel's workgroups run updateData for the
    # instance they are about to solve (the PCIe transfer hides behind the other workgroups' compute) and write x straight into a pinned
    # result array; pageable arrays take update + solve.  Same bits on every path, also with kept groups and after a previous solve
    # (the un-equilibration of kept groups), in the LDS-resident, U-in-LDS, two-per-CU queue and dual-solve launch shapes.
    pat, sets = load_fixture(name)
    if name.startswith("lp_"):
        d = perturbed_batch(pat, sets[0], 0, B)
    elif name == "MPC02":
        d = feasible_batch(pat, sets[0], 0, B)
    else:
        d = dict(zip(("Gpr", "Apr", "c", "h", "b"), rep(sets[0], B)))
    keys = ("Gpr", "Apr", "c", "h", "b")
    g = eicos_amd.BatchSolver(pat, B)
    g.update(*[d[k] for k in keys]); codes0 = g.solve(); x0 = g.solution(); ia0 = g.info_arrays(); y0, z0, s0 = g.duals()
    # a second data set (c, h changed; G, A, b kept) through the classic calls
    c2 = d["c"] * 1.01
    g.update(None, None, c2, None, None); codes1 = g.solve(); x1 = g.solution()
    g.close()
    pins = {k: eicos_amd.PinnedArray(d[k].shape) for k in keys}
    for k in keys:
        pins[k].a[...] = d[k]
    px = eicos_amd.PinnedArray((B, pat.n))
    g = eicos_amd.BatchSolver(pat, B)
    codes = g.update_solve(*[pins[k].a for k in keys], x_out=px.a)
    assert g.last_update_path() == "fused into the solve"
    ia = g.info_arrays(); y, z, s_ = g.duals()
    assert np.array_equal(codes, codes0) and np.array_equal(px.a, x0) and np.array_equal(g.solution(), x0) and np.array_equal(ia["iter"], ia0["iter"])
    assert np.array_equal(y, y0) and np.array_equal(z, z0) and np.array_equal(s_, s0) and np.array_equal(ia["pcost"], ia0["pcost"])
    pc = eicos_amd.PinnedArray((B, pat.n)); pc.a[...] = c2
    xb = np.zeros((B, pat.n))  # (pageable result array: fetched after the launch)
    assert np.array_equal(g.update_solve(None, None, pc.a, None, None, x_out=xb), codes1) and np.array_equal(xb, x1)
    assert g.last_update_path() == "fused into the solve"
    # pageable inputs: update (bounce pipeline) + solve inside the same call ...
    assert np.array_equal(g.update_solve(*[d[k] for k in keys], x_out=px.a), codes0) and np.array_equal(px.a, x0)
    assert g.last_update_path() == "pinned bounce"
    # ... or, switched on, staged: the host copies them into the handle's pinned staging buffer while the kernel runs, one flag per chunk
    monkeypatch.setenv("EICOS_FUSED_STAGED", "1")
    assert np.array_equal(g.update_solve(*[d[k] for k in keys], x_out=px.a), codes0) and np.array_equal(px.a, x0)
    assert g.last_update_path() == "fused into the solve, staged while it runs"
    # ... mixed: G, A pinned, the small arrays pageable; and again (the flags carry a sequence number per call)
    for _ in range(2):
        xb[...] = 0.0
        assert np.array_equal(g.update_solve(pins["Gpr"].a, pins["Apr"].a, d["c"], d["h"], d["b"], x_out=xb), codes0) and np.array_equal(xb, x0)
    assert g.last_update_path() == "fused into the solve, staged while it runs"
    # kept groups from pageable memory (c only): the un-equilibrated G, A of the handle, exactly as in the two-call sequence above
    assert np.array_equal(g.update_solve(None, None, c2, None, None, x_out=xb), codes1) and np.array_equal(xb, x1)
    g.close()
    for pa in list(pins.values()) + [px, pc]:
        pa.close()


def test_launch_durations_are_kept_in_a_ring_of_events():
    # eicos_batch_ms_history: K steps enqueued back to back, every launch's duration read afterwards (bench.py's timed loop has no host
    # synchronisation inside); the ring holds 64, oldest first; "step" = updateData start -> solve end >= update + solve
    pat, sets = load_fixture("lp_afiro")
    B = 32
    d = feasible_batch(pat, sets[0], 0, B)
    g = eicos_amd.BatchSolver(pat, B, device=0)
    assert g.ms_history("solve") == [] and g.ms_history("update") == []
    for k in range(5):
        g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
        g.solve_async()
    g.sync()
    hs, hu, ht = g.ms_history("solve"), g.ms_history("update"), g.ms_history("step")
    assert len(hs) == len(hu) == len(ht) == 5 and all(v > 0 for v in hs + hu)
    assert abs(hs[-1] - g.last_solve_ms()) < 1e-6 and abs(hu[-1] - g.last_update_ms()) < 1e-6
    assert all(t >= s_ + u - 1e-3 for t, s_, u in zip(ht, hs, hu))
    assert g.ms_history("solve", 2) == hs[-2:]
    for k in range(70):
        g.solve_async()
    g.sync()
    assert len(g.ms_history("solve", 100)) == 64
    assert np.all(g.info_arrays()["exitcode"] == 0)
    g.close()


def test_bench_emits_the_contract_json_line():
    # bench.py's one-line JSON: metric/unit of BASELINE.json, roofline and cpu_baseline objects, whole-job value
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "ipm_iterations_per_sec" and d["unit"] == "iter/s" and d["dtype"] == "f64" and d["n_gpus"] == 1
    assert d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["optimal"] == 64
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["iters_match_gpu"] is True
    assert abs(d["value"] - d["config"]["mean_iter"] * 64 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    # the metric's "SOCP": the MPC-SOC variant of the same workload, timed the same way -- a compact entry INSIDE config (the driver's
    # record keeps `config` whole), the full object on stderr
    s = d["config"]["summary"]["soc"]
    assert s["optimal"] == 64 and s["value"] > 0 and 0 < s["frac"] < 1 and s["cpu"] > 0 and s["iters_equal"] == "64/64"
    assert set(d["config"]["summary"]) == {"headline", "soc"}  # (--batch: no `configs` legs)
    full = json.loads(next(l for l in out.stderr.splitlines() if l.startswith("bench details: "))[len("bench details: "):])
    assert full["soc"]["cones"] == 332 and full["soc"]["roofline"]["bound"] == "hbm" and "refinement_vs_oracle" in full["soc"]
    # traffic is only quoted from a PMC summary of exactly this code + workload (none for batch 64) -> null
    assert r["traffic"] is None and s["traffic_ratio"] is None
    # a batch of 64 fits one workgroup per CU: the two independent KKT systems of a pass are solved as one dual solve, so the yardstick
    # that charges the passes over L really made is the stricter one; the launch's tail is reported
    assert 0 < r["frac_dual"] < r["frac"] and s["inst_ms_max"] >= s["inst_ms_p95"] > 0
    assert len(out.stdout.strip().splitlines()[-1]) < 8192
    # every summary entry: the kernel-only rate beside the wall-clock one, the steps timed and the step's span (min / median / max, GPU clock)
    for e in (d["config"]["summary"]["headline"], s):
        assert e["steps"] == 2 and e["value_kernel"] >= e["value"] > 0 and len(e["step_ms"]) == 3 and e["step_ms"][0] <= e["step_ms"][1] <= e["step_ms"][2]


def test_bench_legs_json_is_what_prev_round_parses():
    # `bench.py --legs-json` (the child process behind config.summary.prev_round: every leg of the default line, GPU part only, on the library
    # EICOS_AMD_LIB names): one json object, [value, value_kernel] per leg, every leg timed >= 10 steps
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--legs-json"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert set(d) == {"headline", "soc", "dense_front_b512", "lp_afiro_b256", "lp_bandm_b256", "lp_25fv47_b256", "mpc_b512", "mpc_b4096"}
    for k, v in d.items():
        assert isinstance(v, list) and len(v) == 2 and v[1] >= v[0] > 0, (k, v)


def test_bench_multi_flag_drives_the_product_multi_gpu_layer():
    # bench.py --multi IDS: the benchmark step from ONE process through eicos_multi_* (device list {0, 0} on this box)
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--multi", "0,0", "--total", "128",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["config"]["total_instances"] == 128 and d["config"]["optimal"] == 128
    assert "eicos_multi" in d["config"]["io"] and "(0, 64, 0), (64, 64, 0)" in d["config"]["io"] and d["value"] > 0


def test_bench_eight_shards_keep_the_roofline_below_one():
    # the N > 1 measurement path, dry-run on this one-GPU box -- eight handles / streams / host threads
    # (device list {0 x 8}, 512 instances each = the per-GPU share of configs[2]); every instance optimal, the roofline fraction of the
    # multi-device line stays <= 1 (peak = distinct devices x 8 TB/s), the shard witnesses are in the line
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--multi", "0,0,0,0,0,0,0,0", "--total", "4096",
                          "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    c, r = d["config"], d["roofline"]
    assert d["n_gpus"] == 1 and c["total_instances"] == 4096 and c["optimal"] == 4096 and c["devices"] == [0] * 8 and "eicos_multi" in c["launch"]
    assert 0 < r["frac"] <= 1 and 0 < r["frac_dual"] <= 1 and r["peak"] == 8000.0 and r["devices"] == 1
    assert len(r["per_shard"]) == 8 and all(p_["instances"] == 512 and p_["kernel_ms"] > 0 for p_ in r["per_shard"])
    assert all(0 < p_["frac_of_one_gpu"] <= 1 for p_ in r["per_shard"]) and r["kernel_ms"] >= max(p_["kernel_ms"] for p_ in r["per_shard"])  # (span of the device >= any one launch)
    assert abs(d["value"] - c["mean_iter"] * 4096 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]


def test_bench_launcher_path_rehearsed_with_two_ranks_on_this_gpu():
    # the one-process-per-GPU launch (the driver's N > 1 command line) end to end on the one leased GPU: EICOS_BENCH_REHEARSAL=1 lets the two
    # ranks of torch.distributed.run share device 0 and reduce their counters over gloo (RCCL refuses two ranks on one device); the line must
    # say what it is -- n_gpus = the distinct devices, launch = REHEARSAL -- and hold the contract: both ranks seen, the fixed total split
    # into contiguous shards, every instance optimal, value = iterations of ALL ranks over the slowest rank's time
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    env = dict(os.environ, EICOS_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--total", "1024", "--no-soc"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0]); c = d["config"]
    assert d["n_gpus"] == 1 and c["launch"].startswith("REHEARSAL: 2 ranks") and c["ranks_seen"] == 2 and c["devices"] == [0, 0]
    assert c["total_instances"] == 1024 and c["batch_per_gpu"] == 512 and c["optimal"] == 1024 and d["scaling"] == "strong"
    assert 0 < d["roofline"]["frac"] <= 1 and "cpu_baseline" not in d  # (the CPU leg belongs to the N = 1 line)
    assert abs(d["value"] - c["mean_iter"] * 1024 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]
    # without the rehearsal switch the same command line must fail loudly on this box (LOCAL_RANK 1 has no device), never print a line
    env.pop("EICOS_BENCH_REHEARSAL")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          "--max-restarts", "0", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--total", "64", "--no-soc"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    if eicos_amd.device_count() < 2:
        assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith('{"metric"')]


def test_mpc_soc_batch_1024_all_instances_against_the_oracle():
    # the metric says "SOCP" -- the MPC-SOC variant (332 cones of dimension 3) at the headline's batch, every
    # instance against the oracle: exit code, iterations +-1, pcost 1e-8, x
    pat, sets = load_fixture("MPC02")
    spat = mpc_soc_variant(pat, sets[0])
    B = 1024
    d = feasible_batch(spat, sets[0], 0, B)
    g = eicos_amd.BatchSolver(spat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays(); x = g.solution(); g.close()
    assert np.all(codes == 0)
    _all_instances_against_the_oracle(spat, d, codes, ia, x, tag="MPC02-SOC b1024")


def test_mpc_batch_4096_all_instances_against_the_oracle():
    # north_star's ">= 10x the host at batch 4096" configuration (three workgroups per CU, the 168-VGPR build, the instance queue):
    # all 4096 instances against the oracle (~2 s of 16-core CPU time)
    pat, sets = load_fixture("MPC02")
    B = 4096
    d = feasible_batch(pat, sets[0], 0, B)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays(); x = g.solution(); g.close()
    assert np.all(codes == 0)
    _all_instances_against_the_oracle(pat, d, codes, ia, x, tag="MPC02 b4096")


def test_dynamic_regularisation_extension_matches_oracle():
    # N4 (extension, off by default): same rule on both sides -> same result on the instance whose pivot cancels in
    # the oracle's elimination order; and switching it on must not disturb a well-conditioned batch
    from test_oracle_golden import _fuzz_case
    pat, d = _fuzz_case(9024)
    g = eicos_amd.BatchSolver(pat, 3)
    g.set_dynamic_regularization(2e-7, 1e-13)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays(); x = g.solution()
    assert np.all(codes == 0)
    for i in range(3):
        o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        o.set_dynamic_regularization(2e-7, 1e-13)
        assert o.solve() == 0
        assert abs(o.info()["iter"] - ia["iter"][i]) <= 1
        assert abs(o.info()["pcost"] - ia["pcost"][i]) <= 1e-7 * max(1.0, abs(o.info()["pcost"]))
        o.close()
    g.close()
    pat, sets = load_fixture("MPC02")
    dd = feasible_batch(pat, sets[0], 0, 8)
    g = eicos_amd.BatchSolver(pat, 8)
    g.update(dd["Gpr"], dd["Apr"], dd["c"], dd["h"], dd["b"]); g.solve()
    x0 = g.solution().copy(); it0 = g.info_arrays()["iter"].copy()
    g.set_dynamic_regularization(2e-7, 1e-13); g.solve()
    assert np.array_equal(g.info_arrays()["iter"], it0) and np.abs(g.solution() - x0).max() <= 1e-9 * np.abs(x0).max()
    g.close()


def test_invalid_arguments_are_refused_with_error_codes():
    # the ABI never throws or crashes on bad input: negative codes + a message (eicos_last_error)
    import ctypes as C
    from eicos_amd import binding as b
    L = b._lib()
    ip = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(C.POINTER(C.c_int))
    h = C.c_void_p()
    create = lambda n, m, p, q, Gjc, Gir: L.eicos_batch_create(n, m, p, m, len(q), ip(q) if len(q) else None, ip(Gjc), ip(Gir), None, None, 1, -1, C.byref(h))
    good = ([0, 1, 2], [0, 1])
    assert create(2, 2, 0, [], *good) == 0
    L.eicos_batch_destroy(h)
    assert create(-1, 2, 0, [], *good) == -1                       # negative dimension
    assert create(2, 2, 0, [3], *good) == -1                       # sum(q) > m
    assert create(2, 2, 0, [0], *good) == -1                       # cone of dimension 0
    assert create(2, 2, 0, [], [0, 1, 2], [0, 5]) == -1            # row index out of range
    assert create(2, 2, 0, [], [0, 2, 1], [0, 1]) == -1            # column pointers decrease
    assert create(2, 2, 0, [], [0, 2, 2], [1, 1]) == -1            # duplicate row index in a column
    assert b"column" in L.eicos_last_error() or b"row" in L.eicos_last_error()
    pat, sets = load_fixture("lp_afiro")
    g = eicos_amd.BatchSolver(pat, 2)
    dp = C.POINTER(C.c_double)
    z = np.zeros(8)
    assert L.eicos_batch_update(g._h, 1, 2, None, None, z.ctypes.data_as(dp), None, None) == -1   # range out of bounds
    assert L.eicos_batch_set_warm_start(g._h, -1.0) == -1
    assert L.eicos_solve(g._h, None) == -1                          # single-instance call on a batch of two
    g.close()


def test_dynamic_regularisation_on_dense_fronts_with_delta_sized_pivots():
    # regression (round-2 differential campaign, case 61161): a dense pattern (tile path chosen by the analysis) whose last
    # blocks hold pivots of the size of the static regularisation.  With off-diagonal tiles formed through an explicit
    # inverse of the diagonal tile those pivots came out with the wrong sign, the extension "repaired" them and the
    # factorisation blew up (NUMERICS on every instance); with triangular solves no pivot triggers and x matches the oracle
    from eicos_amd.generate import random_socp_pattern
    pat, base = random_socp_pattern(51, 19, 27, [3, 3], density=0.3, seed=61161)
    d = feasible_batch(pat, base, 0, 3, seed=61161)
    g = eicos_amd.BatchSolver(pat, 3)
    assert g.dims()["factor_path"] == 1
    g.set_dynamic_regularization(2e-7, 1e-13)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"]); codes = g.solve(); ia = g.info_arrays(); x = g.solution()
    for i in range(3):
        o = OracleSolver(pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        o.set_dynamic_regularization(2e-7, 1e-13)
        oc = o.solve(); oi = o.info()
        assert codes[i] == oc == 0 and ia["iter"][i] == oi["iter"]
        assert np.abs(x[i] - o.x()).max() <= 1e-9 * max(1.0, np.abs(o.x()).max())
        o.close()
    g.close()


@pytest.mark.parametrize("name,env", [("lp_bandm", {}), ("lp_bandm", {"EICOS_TILES": "1"}), ("issue98", {"EICOS_TILES": "1"})])
def test_dynamic_regularisation_sign_pattern_on_the_tile_and_hybrid_paths(name, env, monkeypatch):
    # N4 on the dense-front code: the 16 x 16 diagonal LDL' repairs a pivot whose sign disagrees with the quasi-definite sign
    # pattern (tl_psign, one entry per slot of the KKT-space vectors).  On a well-conditioned instance no pivot may trigger:
    # with a wrong sign table (or a wrong slot offset in hybrid mode, where the top block sits behind the scalar part)
    # every pivot of the block would be replaced and the solve would fall apart.
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pat, sets = load_fixture(name)
    o = OracleSolver(pat, sets[0]); oc = o.solve(); oi = o.info()
    g = eicos_amd.BatchSolver(pat, 2); g.update(*rep(sets[0], 2))
    g.set_dynamic_regularization(2e-7, 1e-13)
    codes = g.solve(); gi = g.info()
    assert list(codes) == [oc, oc] and abs(gi[0]["iter"] - oi["iter"]) <= 1
    assert abs(gi[0]["pcost"] - oi["pcost"]) <= 1e-7 * max(1.0, abs(oi["pcost"]))
    g.close(); o.close()
