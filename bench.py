#!/usr/bin/env python3
"""bench.py -- headline benchmark: IPM iterations/sec (fp64) on batched MPC-shaped SOCPs.

  python bench.py --gpus N --steps K --warmup W          (N=1 default)
  N>1, one process per GPU : python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                             bench.py --gpus N --steps K --warmup W      (RANK / LOCAL_RANK / WORLD_SIZE from the launcher; RCCL for the counters only)
  N>1, ONE process         : python bench.py --gpus N ...  WITHOUT a launcher (no RANK in the environment) drives devices 0..N-1 through the
                             product's own multi-GPU layer (eicos_multi_*), exactly as `--multi 0,1,...,N-1`; it exits non-zero when fewer
                             than N devices are visible -- it never falls through to a 1-GPU run labelled N

Workload: the MPC02 sparsity pattern (the MPC01 blob named by BASELINE.json is missing from the
reference mount) with strictly feasible (c,h,b) from eicos_amd.generate keyed by (seed, GLOBAL
instance index).  One "step" = one pass of the hot path over the batch with the raw inputs already resident in
HBM: updateData (equilibrate + transposes, on device) followed by the batched cold-start solve.
Metric value = sum over instances of Information.iter / time.

  N = 1 : BASELINE.json configs[1] -- batch 1024 on one MI355X.
  N > 1 : BASELINE.json configs[2] -- a FIXED total of 4096 instances in contiguous shards of 4096/N (512 per GPU
          at N = 8): "scaling": "strong".  `--batch B` forces B instances per GPU instead (weak scaling).
Instances are independent: no data-path collective; only the timing / iteration counters are reduced.

`config.summary` of the line carries one compact entry per workload: the headline, "soc" -- the same step on the MPC-SOC
variant of the pattern (332 second-order cones of dimension 3), so that the metric's "SOCP" is what
gets timed -- and, at N = 1, every other BASELINE.json config (dense-front batch 512, three LPnetlib patterns at batch 256,
MPC02 at batch 512 and 4096).  The full per-workload objects go to stderr / gpurun_out/bench_details.json.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 measured achievable
TOTAL_STRONG = 4096    # BASELINE.json configs[2]


def usable_cores():
    """CPU cores this process may really use: affinity mask, capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def kernel_source_hash():
    """Identity of the kernels a PMC summary was taken on: sha256 over the device sources."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "eicos_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile":  # (the build flags are part of the identity)
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(dims, ia, sweeps="n_ldlsolve"):
    """Algorithmic HBM bytes of ONE solve launch (fp64 values only; shared index arrays excluded),
    SURVEY.md 8(d) formula with the measured counters of every instance.
    sweeps="n_sweep": charge the passes over L, A, G that were really made (a dual right-hand-side solve streams them once for
    two solves) instead of one pass per LDL solve -- the stricter yardstick for paths that use dual solves (`frac_dual`)."""
    n, p, m = dims["n"], dims["p"], dims["m"]
    N, nnzK, nnzL = dims["dim_K"], dims["nnzK"], dims["nnzL"]
    nnzAG = dims["nnzA"] + dims["nnzG"]
    f = ia["n_factor"].astype(np.float64).sum()
    r = ia[sweeps].astype(np.float64).sum()
    it = (ia["iter"].astype(np.float64) + 1).sum()
    per_factor = nnzK + nnzL + N                 # read K values, write L and D
    per_solve = 2 * nnzL + 3 * N                 # L forward + L backward, D, rhs in / x out
    per_resid = 2 * nnzAG + 4 * N                # refinement residual: A,A',G,G' values + vectors
    per_iter = 2 * nnzAG + 6 * (n + p + 2 * m) + 30 * m + 6 * (n + p)  # residuals, scalings, RHS, line searches
    return 8.0 * (f * per_factor + r * (per_solve + per_resid) + it * per_iter)


def pmc_traffic(tag):
    """HBM traffic of one launch from the committed summary of separate `rocprofv3 --pmc` passes (PMC counters cannot
    be collected inside a timed run).  Only a summary taken on THESE kernel sources and THIS workload counts; anything
    else would describe a different code state -> null."""
    import glob
    want = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            pm = json.load(open(f))
        except (OSError, ValueError):
            continue
        if pm.get("kernel_source_sha256") == want and pm.get("workload") == tag and "traffic_bytes" in pm:
            return float(pm["traffic_bytes"]), os.path.relpath(f, ROOT)
    return None, None


class Job:
    """One workload on this rank's GPU: inputs resident in HBM, solver handle, step()."""

    def __init__(self, args, pat, sets, first, B, local_rank, soc=False, multi_ids=None):
        import torch
        import eicos_amd
        from eicos_amd.generate import SEED, feasible_batch, mpc_soc_variant, perturbed_batch
        self.torch, self.args, self.B, self.first = torch, args, B, first
        self.pat = mpc_soc_variant(pat) if soc else pat
        self.base = sets[0]
        gen = perturbed_batch if args.perturb else feasible_batch
        self.data = gen(self.pat, sets[0], first, B, SEED)
        self.multi = multi_ids
        if multi_ids:
            # --multi: ONE process drives every listed device through the product's own multi-GPU layer (eicos_multi_* of
            # include/eicos_amd.h: contiguous shards, one handle + stream per list entry); every shard's inputs are resident in the
            # HBM of ITS device before the timed region (no copy in a step)
            self.solver = eicos_amd.MultiBatchSolver(self.pat, B, multi_ids)
            self.shards = self.solver.shards()
            self.sdevs = [{k: torch.from_numpy(v[f:f + c]).to(f"cuda:{dv}") for k, v in self.data.items()} for (f, c, dv) in self.shards]
            self.devs = [self.sdevs[0]]
            self.dims = self.solver.shard_dims(0)
            self.dims["kernel_build"] = "per shard"
            self.step_no = 0
            return
        dev = f"cuda:{local_rank}"
        self.devs = [{k: torch.from_numpy(v).to(dev) for k, v in self.data.items()}]
        if args.resolve > 0:
            rng = np.random.default_rng(SEED + 17 + first)
            alt = dict(self.data)
            alt["c"] = self.data["c"] * (1 + args.resolve * rng.uniform(-1, 1, self.data["c"].shape))
            alt["h"] = self.data["h"] + args.resolve * (1 + np.abs(self.data["h"])) * rng.uniform(0, 1, self.data["h"].shape)
            self.devs.append({k: torch.from_numpy(v).to(dev) for k, v in alt.items()})
        self.step_no = 0
        self.solver = eicos_amd.BatchSolver(self.pat, B, device=local_rank)
        self.dims = self.solver.dims()
        self.dims["kernel_build"] = self.solver.kernel_build()
        if args.warm > 0:
            self.solver.set_warm_start(args.warm)

    def step(self):
        if self.multi:
            for s, d in enumerate(self.sdevs):
                ptr = lambda k: d[k].data_ptr() if d[k].numel() else 0
                self.solver.shard_update_device(s, ptr("Gpr"), ptr("Apr"), ptr("c"), ptr("h"), ptr("b"))
            self.solver.solve_async()  # every shard's kernels are enqueued on its own stream before any is waited for
            self.step_no += 1
            return
        d = self.devs[self.step_no % len(self.devs)]
        ptr = lambda k: d[k].data_ptr() if d[k].numel() else 0
        self.solver.update_device(ptr("Gpr"), ptr("Apr"), ptr("c"), ptr("h"), ptr("b"))
        self.solver.solve_async()
        self.step_no += 1

    def run(self, dist, steps, warmup, min_seconds=0.0):
        """W untimed steps, then exactly K timed steps between barrier + synchronize on both sides.  The timed loop only ENQUEUES (update +
        solve per step, no host synchronisation inside): every launch's duration is read afterwards from the handle's ring of HIP events
        (eicos_batch_ms_history), so the wall time holds no host round trip per step.  min_seconds > 0 (auxiliary legs): K is raised until the
        timed region lasts at least that long (sized from one fenced pilot step)."""
        torch = self.torch

        def fence():
            if dist is not None:
                dist.barrier()
            self.solver.sync()
            for dv in (sorted(set(self.multi)) if self.multi else [None]):
                torch.cuda.synchronize(dv)

        # one untimed parity step first: the counters of every instance's FIRST solve are what the CPU sample is compared
        # with (the reference's pinfres / dinfres are sticky across solve() calls on one object -- SURVEY App. A.2 -- so an
        # infeasible instance exits at once when it is solved AGAIN; the CPU sample solves every instance once)
        self.step()
        fence()
        self.ia_first = self.solver.info_arrays()
        for _ in range(warmup):
            self.step()
        fence()
        if min_seconds > 0:
            tp = time.perf_counter()
            self.step()
            fence()
            steps = min(5000, max(steps, int(np.ceil(min_seconds / max(time.perf_counter() - tp, 1e-6)))))
        # (a library of a previous round -- the prev_round leg -- has no event ring: its launches are read one by one, with a host round trip per step)
        ring = (not self.multi) and self.solver.ms_history("solve", 1) is not None
        kernel_ms, update_ms, shard_ms, step_ms = [], [], [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            self.step()
            if ring:
                continue
            # per-launch kernel duration from HIP events recorded on the solver's own stream
            if self.multi:
                mx, per = self.solver.last_solve_ms()
                kernel_ms.append(mx); shard_ms.append(per)  # (mx = slowest shard of the step)
                update_ms.append(max(self.solver.shard_last_update(s_)[1] for s_ in range(len(self.shards))))
            else:
                kernel_ms.append(self.solver.last_solve_ms())
                update_ms.append(self.solver.last_update_ms())
            step_ms.append((time.perf_counter() - ts) * 1e3)
        fence()
        dt = time.perf_counter() - t0
        if ring:  # the last min(K, 64) launches of the timed region, on the GPU's own clock
            k = min(steps, 64)
            kernel_ms, update_ms, step_ms = (self.solver.ms_history(w, k) for w in ("solve", "update", "step"))
        ia = self.solver.info_arrays()
        dev = getattr(self, "ctl_device", None) or self.devs[0]["c"].device
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        cnt = torch.tensor([int(ia["iter"].sum()), int((ia["exitcode"] == 0).sum()), self.B, 1], dtype=torch.float64, device=dev)  # ([3]: ranks that took part)
        km = float(np.mean(kernel_ms))
        kmin = torch.tensor([km], dtype=torch.float64, device=dev); kmax = kmin.clone()
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            dist.all_reduce(kmin, op=dist.ReduceOp.MIN)  # launch skew between the ranks (timing only: no data-path collective)
            dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        tot_iters, tot_ok, tot_B, ranks_seen = (int(v) for v in cnt.tolist())
        sm = np.asarray(step_ms, dtype=np.float64)
        return dict(ranks_seen=ranks_seen, dt=float(t.item()), iters=tot_iters, ok=tot_ok, instances=tot_B, ia=ia, kernel_ms=km, update_ms=float(np.mean(update_ms)),
                    steps=steps, timing=("event ring, no host synchronisation inside the timed loop" if ring else "one host round trip per step"),
                    # one step = updateData start -> solve end (ring: on the GPU's clock; else the host's clock around the step incl. its event wait)
                    step_ms={"min": float(np.nanmin(sm)), "median": float(np.nanmedian(sm)), "max": float(np.nanmax(sm)), "n": int(sm.size)},
                    kernel_ms_minmax=[float(np.min(kernel_ms)), float(np.max(kernel_ms))],
                    kernel_ms_min_over_ranks=float(kmin.item()), kernel_ms_max_over_ranks=float(kmax.item()),
                    shard_kernel_ms=([float(v) for v in np.mean(np.asarray(shard_ms), axis=0)] if shard_ms else None))

    def report(self, r, steps, tag):
        dims, ia = self.dims, r["ia"]
        # --multi: `ia` covers the instances of ALL shards and kernel_ms is the slowest shard's launch, so the yardstick is the HBM
        # peak of all DISTINCT devices together (per device: its shards' bytes over the same span); frac stays <= 1 by construction
        ndev = len(set(self.multi)) if self.multi else 1
        peak = HBM_PEAK_GBS * ndev
        abytes = algorithmic_bytes(dims, ia)
        achieved = abytes / (r["kernel_ms"] * 1e-3) / 1e9
        abytes_dual = algorithmic_bytes(dims, ia, "n_sweep")
        traffic, src = pmc_traffic(tag)
        per_shard = None
        if self.multi and r.get("shard_kernel_ms"):
            per_shard = []
            for (f, c, dv), ms in zip(self.shards, r["shard_kernel_ms"]):
                sb = algorithmic_bytes(dims, {k: v[f:f + c] for k, v in ia.items()})
                per_shard.append({"device": dv, "instances": c, "kernel_ms": ms, "frac_of_one_gpu": sb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        inst_ms = np.sort(ia["solve_us"]) * 1e-3  # device wall time of every instance's solve (its workgroup): the launch's tail
        return {
            "value": r["iters"] * steps / r["dt"], "unit": "iter/s", "ms_per_step": r["dt"] / steps * 1e3,
            # the same iterations over the kernel time alone (mean HIP-event duration of the solve launches of the timed region): what the
            # wall-clock value would be with nothing but the solve kernel in a step -- the gap between the two is updateData + launch gaps
            "value_kernel": r["iters"] / (r["kernel_ms_max_over_ranks"] * 1e-3), "steps": steps,
            "step_ms": r["step_ms"], "kernel_ms_minmax": r["kernel_ms_minmax"], "timing": r["timing"],
            "solves_per_sec": r["instances"] * steps / r["dt"], "optimal": r["ok"], "instances": r["instances"],
            "dim_K": dims["dim_K"], "nnzK": dims["nnzK"], "nnzL": dims["nnzL"], "levels": dims["nlevels"], "cones": dims["ncones"],
            "mean_iter": r["iters"] / max(1, r["instances"]),  # (over ALL ranks / shards, like `value`)
            "mean_ldl_solves_per_iter": float(ia["n_ldlsolve"].sum() / max(1, ia["iter"].sum())),
            "threads_per_block": dims["threads_per_block"], "resident_blocks": dims["resident_blocks"],
            "factor_path": ("scalar", "tile", "hybrid")[dims.get("factor_path", 0)], "lds_resident": bool(dims.get("lds_resident", 0)),
            "kernel_build": dims.get("kernel_build"),
            "update_kernel_ms": r["update_ms"],
            # how much of the launch is its slowest instances: a launch cannot end before its slowest instance does
            "instance_ms": {"mean": float(inst_ms.mean()), "p95": float(inst_ms[int(0.95 * (len(inst_ms) - 1))]), "max": float(inst_ms[-1])},
            "kernel_ms_min_over_ranks": r["kernel_ms_min_over_ranks"], "kernel_ms_max_over_ranks": r["kernel_ms_max_over_ranks"],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": src,
                         "kernel": "k_solve", "kernel_ms": r["kernel_ms"], "algorithmic_bytes_per_launch": abytes,
                         # the same with the passes over L / A / G a dual right-hand-side solve really makes (= frac when no dual solves ran)
                         "frac_dual": abytes_dual / (r["kernel_ms"] * 1e-3) / 1e9 / peak, "algorithmic_bytes_dual": abytes_dual,
                         **({"devices": ndev, "per_shard": per_shard} if self.multi else {})},
        }

    def cpu_baseline(self, ia, target_s=30.0):
        """CPU oracle (a port, NOT the EiCOS binary: Eigen is absent) on a bounded sample of the same workload."""
        from oracle import oracle as orc
        data, pat, B = self.data, self.pat, self.B
        cores = usable_cores()
        sub = lambda k, a, b: data[k][a:b]
        run = lambda a, b: orc.batch_solve(pat, sub("Gpr", a, b), sub("Apr", a, b), sub("c", a, b), sub("h", a, b), sub("b", a, b), cores, native=True)
        # bounded sample: a pilot of one instance per core sizes the sample to 10..30 s of CPU work (capped at four passes over
        # the batch; the pilot runs cold and overestimates the time per instance by up to 3x, hence the target of 30), so that
        # small and large patterns are both timed over a comparable span
        npil = int(min(B, cores))
        r0 = run(0, npil)
        per_inst_cpu = (r0["seconds"] + r0["update_seconds"]) * min(cores, npil) / npil
        want = int(max(npil, min(4 * B, target_s / max(per_inst_cpu, 1e-9))))
        reps, ns = (1, want) if want <= B else (min(4, -(-want // B)), B)
        tot_iters, wall, native = 0, 0.0, False
        for _ in range(reps):
            r = run(0, ns)
            native = r["native"]
            tot_iters += int(r["iters"].sum()); wall += r["seconds"] + r["update_seconds"]
        # Parity counts come from the PORTABLE oracle build (liboracle.so: no FMA contraction, the checker of the test suite), not from
        # the -march=native build that was timed above (g++ contracts a*b+c there, so it rounds differently from the GPU's unfused
        # arithmetic); a bounded sample of the same instances, outside every timed region.
        npar = int(min(ns, 512))
        rp = orc.batch_solve(pat, sub("Gpr", 0, npar), sub("Apr", 0, npar), sub("c", 0, npar), sub("h", 0, npar), sub("b", 0, npar), cores, native=False)
        diff = np.abs(rp["iters"].astype(np.int64) - ia["iter"][:npar].astype(np.int64))
        match, maxdiff = bool(diff.max() == 0), int(diff.max())
        n_eq, n_1, n_code = int((diff == 0).sum()), int((diff <= 1).sum()), int((rp["exitcodes"] == ia["exitcode"][:npar]).sum())
        return {"value": float(tot_iters / wall), "unit": "iter/s", "cores": cores, "kind": "port",
                "sample": f"first {ns} instances x {reps} pass(es), one instance per thread at a time (updateData+solve), "
                          f"{wall:.2f}s wall = {wall * cores:.0f} core-s; oracle built {'-O2 -march=native on this host (reference Release flags)' if native else '-O2 (portable)'}; "
                          "symbolic analysis hoisted out of the timed solve (the reference repeats analyzePattern in every solve(), src/eicos.cpp:897)",
                "build": "-O2 -march=native" if native else "-O2",
                # parity tolerance on iteration counts is +-1: rounding may move an exit by one pass
                # (a perturbed, ill-conditioned instance can stall for tens of passes before a reduced-accuracy exit: which pass
                # that is depends on rounding; the exit codes still agree)
                "iters_match_gpu": match, "iters_max_abs_diff_vs_gpu": maxdiff,
                "instances_compared": npar, "parity_build": "-O2 (portable, no FMA contraction)", "iters_equal": n_eq, "iters_within_1": n_1, "exitcodes_equal": n_code,
                "per_core": float(tot_iters / wall / cores)}


def refinement_profile(job, n=16):
    """Per-solve refinement counts of the first n instances, GPU vs oracle (Information.nitref1..3 of the LAST pass and the
    totals n_ldlsolve / n_factor of the whole solve): explains the LDL solves per pass of a workload -- every solveKKT costs
    1 + k_ref solves, k_ref decided by the reference's stopping rules (src/eicos.cpp:1579-1593) on the refinement residual."""
    from oracle.oracle import OracleSolver
    from eicos_amd.problem_io import Values
    d, ia = job.data, job.ia_first
    n = min(n, job.B)
    rows, same = [], 0
    for i in range(n):
        o = OracleSolver(job.pat, Values(d["Gpr"][i], d["Apr"][i], d["c"][i], d["h"][i], d["b"][i]))
        o.solve(); oi = o.info(); o.close()
        g = (int(ia["n_ldlsolve"][i]), int(ia["iter"][i]), int(ia["nitref1"][i]), int(ia["nitref2"][i]), int(ia["nitref3"][i]))
        c = (int(oi["n_ldlsolve"]), int(oi["iter"]), int(oi["nitref1"]), int(oi["nitref2"]), int(oi["nitref3"]))
        same += g == c
        rows.append({"gpu": g, "oracle": c})
    nf_g = ia["n_factor"][:n].astype(np.int64)
    calls = int((2 + 3 * (nf_g - 1)).sum())  # solveKKT calls: two initialisation solves + three per pass that factorised
    tot_g = sum(r["gpu"][0] for r in rows); tot_o = sum(r["oracle"][0] for r in rows)
    passes = int(nf_g.sum())
    return {"instances": n, "identical_(n_ldlsolve,iter,nitref1,nitref2,nitref3)": same,
            "ldl_solves_gpu": tot_g, "ldl_solves_oracle": tot_o, "solveKKT_calls": calls,
            "mean_refinement_steps_per_solveKKT": tot_g / max(1, calls) - 1.0,
            "ldl_solves_per_factorisation": tot_g / max(1, passes), "first": rows[:2]}

def host_e2e(pat, sets, B, local_rank, device_value, steps=12, warmup=3):
    """The reference's REAL call sequence on HOST arrays (updateData(double *...) -> solve() -> solution(), include/eicos.hpp:155-160) for
    the headline batch: every step hands over all five host arrays, solves, and copies x back to the host.  Two variants: `pageable`
    (plain numpy arrays: the pinned double-buffer bounce of csrc/api.cpp), `registered` (the same kind of arrays pinned in place once with
    eicos_host_register) and `pinned` (arrays from eicos_host_alloc); the latter two are read / written in place over PCIe; `pinned_fused` = the same pinned
    arrays through the one-call form eicos_batch_update_solve (the solve kernel's workgroups pull their inputs themselves).  This is the number the CPU baseline -- which reads host arrays -- is directly comparable with; the headline
    value has its inputs resident in HBM."""
    import eicos_amd
    from eicos_amd.generate import SEED, feasible_batch
    data = feasible_batch(pat, sets[0], 0, B, SEED)
    keys = ("Gpr", "Apr", "c", "h", "b")
    solver = eicos_amd.BatchSolver(pat, B, device=local_rank)
    out = {"batch": B, "steps": steps, "bytes_in_per_step": int(sum(data[k].nbytes for k in keys)), "bytes_out_per_step": int(B * pat.n * 8)}
    pinned = []
    try:
        registered = []
        for variant in ("pageable", "registered", "pinned", "pinned_fused"):
            try:
                if variant == "registered":  # the caller's own (numpy) arrays pinned in place once: eicos_host_register
                    arrs = {k: np.ascontiguousarray(data[k]).copy() for k in keys}
                    x = np.zeros((B, pat.n))
                    for a in list(arrs.values()) + [x]:
                        if a.size:
                            eicos_amd.host_register(a)
                            registered.append(a)
                elif variant == "pinned_fused":
                    pass  # (the arrays of the "pinned" variant, through eicos_batch_update_solve)
                elif variant == "pinned":
                    arrs = {}
                    for k in keys:
                        pa = eicos_amd.PinnedArray(data[k].shape)
                        pa.a[...] = data[k]
                        pinned.append(pa)
                        arrs[k] = pa.a
                    px = eicos_amd.PinnedArray((B, pat.n))
                    pinned.append(px)
                    x = px.a
                else:
                    arrs, x = {k: data[k] for k in keys}, np.zeros((B, pat.n))

                def step():
                    if variant == "pinned_fused":  # ONE call: the solve kernel pulls every instance's inputs itself and writes x into the pinned result array
                        solver.update_solve(arrs["Gpr"], arrs["Apr"], arrs["c"], arrs["h"], arrs["b"], x_out=x)
                        return
                    solver.update(arrs["Gpr"], arrs["Apr"], arrs["c"], arrs["h"], arrs["b"])
                    solver.solve_async()
                    solver.sync()
                    solver.solution_into(x)
                for _ in range(warmup):
                    step()
                upd, t0 = [], time.perf_counter()
                for _ in range(steps):
                    step()
                    upd.append(solver.last_update_ms())
                dt = time.perf_counter() - t0
                ia = solver.info_arrays()
                out[variant] = {"value": float(ia["iter"].sum() * steps / dt), "ms_per_step": dt / steps * 1e3, "optimal": int((ia["exitcode"] == 0).sum()),
                                "update_ms": float(np.mean(upd)), "update_path": solver.last_update_path(), "kernel_ms": solver.last_solve_ms(),
                                "vs_device_resident": float(ia["iter"].sum() * steps / dt / device_value)}
            except RuntimeError as e:  # (pinning refused on this box -- locked-memory limit: the variant is skipped, the others still run)
                out[variant + "_error"] = str(e)[:200]
    finally:
        solver.close()
        for a in registered:
            eicos_amd.host_unregister(a)
        for pa in pinned:
            pa.close()
    return out


def summarise(rep):
    """Compact form of one workload's report for `config.summary` (short keys, numbers rounded: the whole line stays below 6 KB)."""
    roof, cpu = rep["roofline"], rep.get("cpu_baseline")
    r3 = lambda v: None if v is None else float(f"{v:.4g}")
    sm = rep["step_ms"]
    o = {"value": r3(rep["value"]), "value_kernel": r3(rep["value_kernel"]), "steps": rep["steps"], "step_ms": [r3(sm["min"]), r3(sm["median"]), r3(sm["max"])],
         "batch": rep["instances"], "optimal": rep["optimal"], "mean_iter": r3(rep["mean_iter"]),
         "ldl_per_iter": r3(rep["mean_ldl_solves_per_iter"]), "path": rep["factor_path"] + "/" + str(rep.get("kernel_build")),
         "kernel_ms": r3(roof["kernel_ms"]), "frac": r3(roof["frac"]), "frac_dual": r3(roof["frac_dual"]),
         "algo_GB": r3(roof["algorithmic_bytes_per_launch"] / 1e9),
         "traffic_ratio": r3(roof["traffic"] / roof["algorithmic_bytes_per_launch"]) if roof.get("traffic") else None,
         "inst_ms_p95": r3(rep["instance_ms"]["p95"]), "inst_ms_max": r3(rep["instance_ms"]["max"])}
    if cpu:
        o.update({"cpu": r3(cpu["value"]), "cpu_cores": cpu["cores"], "x_cpu": r3(rep["value"] / cpu["value"]),
                  "iters_equal": f"{cpu['iters_equal']}/{cpu['instances_compared']}", "codes_equal": f"{cpu['exitcodes_equal']}/{cpu['instances_compared']}"})
    return o


def run_config(args, name, pattern, batch, local_rank, perturb=False, soc=False, steps=10, warmup=3, cpu_s=6.0, min_seconds=0.3):
    """One additional BASELINE.json config on this GPU, timed like the headline (updateData + solve per step; at least `steps` steps AND
    `min_seconds` of timed work, after `warmup` steps that bring the clocks back up behind the previous leg's CPU baseline): its own
    value, roofline and CPU baseline (a smaller sample than the headline's, so that the default run stays within minutes)."""
    import copy
    import eicos_amd
    a = copy.copy(args)
    a.perturb, a.resolve, a.warm = perturb, 0.0, 0.0
    if pattern == "dense-front":
        from eicos_amd.generate import dense_front_pattern
        pat, base = dense_front_pattern(2000, 32, 64)
        sets = [base]
    else:
        pat, sets = eicos_amd.read_problem(os.path.join(ROOT, "tests", "golden", pattern + ".epb"))
    job = Job(a, pat, sets, 0, batch, local_rank, soc=soc)
    res = job.run(None, steps, warmup, min_seconds)
    rep = job.report(res, res["steps"], f"{pattern}{'-SOC' if soc else ''} batch={batch}")
    rep["workload"] = f"{pattern}{'-SOC' if soc else ''}, batch {batch}, {'perturbed (c,h)' if perturb else 'strictly feasible generated (c,h,b)'}"
    rep["exit_codes"] = {str(k): int(v) for k, v in zip(*np.unique(res["ia"]["exitcode"], return_counts=True))}
    if not args.no_cpu_baseline:
        rep["cpu_baseline"] = job.cpu_baseline(job.ia_first, cpu_s)
        rep["gpu_over_cpu"] = rep["value"] / rep["cpu_baseline"]["value"]
    job.solver.close()
    return rep


def aux_legs():
    """The auxiliary legs of the default N = 1 line: (summary key, run_config arguments).  Every leg: >= 10 steps and >= 0.3 s timed, warm-up 3."""
    legs = [("dense_front_b512", dict(name="dense_front", pattern="dense-front", batch=512, cpu_s=8.0))]
    legs += [(f"{nm}_b256", dict(name=nm, pattern=nm, batch=256, perturb=True, cpu_s=4.0)) for nm in ("lp_afiro", "lp_bandm", "lp_25fv47")]
    legs += [("mpc_b512", dict(name="mpc_b512", pattern="MPC02", batch=512, cpu_s=6.0)),
             ("mpc_b4096", dict(name="mpc_b4096", pattern="MPC02", batch=4096, cpu_s=6.0))]
    return legs


PREV_LIB = os.path.join(ROOT, "build_exp", "r05", "libeicos_amd.so")  # the previous round's final library (git worktree at 87ad960 + make; build_exp/ is not tracked)


def legs_only(args, local_rank=0):
    """`--legs-json`: every workload of the default line (headline, soc, the auxiliary legs) on whatever library EICOS_AMD_LIB names, GPU part
    only (no CPU baseline), one compact json object on stdout.  The default run calls this in CHILD processes for the previous round's
    library and for the current one, back to back on the same box: `config.summary.prev_round`."""
    r3 = lambda v: float(f"{v:.4g}")
    out = {}
    legs = [("headline", dict(name="headline", pattern="MPC02", batch=1024)), ("soc", dict(name="soc", pattern="MPC02", batch=1024, soc=True))] + aux_legs()
    args.no_cpu_baseline = True
    for k, kw in legs:
        kw.pop("cpu_s", None)
        try:
            v = run_config(args, kw.pop("name"), kw.pop("pattern"), kw.pop("batch"), local_rank, **kw)
            out[k] = [r3(v["value"]), r3(v["value_kernel"])]
        except Exception as e:  # noqa: BLE001
            out[k] = str(e)[:120]
    print(json.dumps(out), flush=True)


def prev_round_legs(local_rank):
    """Same-box round-over-round delta: the legs on the previous round's library and on the current one, each in its own child process, back to
    back (process-to-process spread on one box is +-3 %: read the pair, not one number).  None when the previous library is not there."""
    import subprocess
    if not os.path.exists(PREV_LIB):
        return None
    out = {"lib": os.path.relpath(PREV_LIB, ROOT), "columns": "[value, value_kernel] iter/s", "order": "prev, cur, prev, cur"}
    for tag, lib in (("prev", PREV_LIB), ("cur", ""), ("prev2", PREV_LIB), ("cur2", "")):
        env = dict(os.environ)
        env.pop("EICOS_AMD_LIB", None)
        if lib:
            env["EICOS_AMD_LIB"] = lib
        env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", str(local_rank))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--legs-json"], env=env, capture_output=True, text=True, timeout=300)
            out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            out[tag] = str(e)[:120]
    return out


def launch_mode(gpus, env, multi, visible_devices):
    """How this invocation reaches its GPUs -> ("single", None) | ("dist", None) | ("multi", [device ids]).  Raises SystemExit with a
    message instead of ever running fewer GPUs than --gpus names.  `visible_devices` is a callable (no GPU is touched unless needed)."""
    world = int(env.get("WORLD_SIZE", "1"))
    launched = "RANK" in env and world >= 1 and ("MASTER_PORT" in env or world > 1)
    if multi:
        if launched and world > 1:
            raise SystemExit("--multi is the single-process multi-GPU path: do not launch it under torch.distributed.run")
        ids = [int(t) for t in multi.split(",")]
        nvis = visible_devices()
        if max(ids) >= nvis or min(ids) < 0:
            raise SystemExit(f"--multi {multi}: only {nvis} device(s) visible")
        return "multi", ids
    if launched:
        if world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}")
        return ("dist" if world > 1 else "single"), None
    if gpus > 1:  # no launcher: one process, the product's own multi-GPU layer over devices 0..N-1 -- or a loud failure
        nvis = visible_devices()
        if nvis < gpus:
            raise SystemExit(f"--gpus {gpus} without a torch.distributed launcher needs {gpus} visible devices for the in-process "
                             f"eicos_multi_* path, found {nvis}; refusing to run a smaller job under that label")
        return "multi", list(range(gpus))
    return "single", None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="instances PER GPU (weak scaling when N > 1).  Default: 1024 at "
                    "N = 1 (BASELINE configs[1]); at N > 1 a fixed total of 4096 split over the ranks (configs[2], strong)")
    ap.add_argument("--total", type=int, default=None, help="fixed total number of instances split over the ranks (strong scaling)")
    ap.add_argument("--pattern", default="MPC02", help="fixture name under tests/golden (BASELINE config 3: lp_*), a path to an EPB1 / ECOS data.h problem file, "
                    "or 'dense-front' (BASELINE config 4: n=2000, 32 cones x 64, generated)")
    ap.add_argument("--soc", action="store_true", help="headline on the MPC-SOC variant (332 cones of dim 3) only")
    ap.add_argument("--no-soc", action="store_true", help="skip the additional MPC-SOC run of the default workload")
    ap.add_argument("--perturb", action="store_true", help="LPnetlib-style batch: perturb c,h of the fixture "
                    "(SURVEY.md 8d config 4) instead of generating strictly feasible (c,h,b)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the additional BASELINE.json configs of the default N = 1 line "
                    "(dense-front batch 512, LPnetlib batch 256, MPC02 batch 512)")
    ap.add_argument("--resolve", type=float, default=0.0, metavar="EPS", help="MPC-style re-solves: steps alternate between "
                    "the batch and a copy with c, h perturbed by EPS (relative), so every solve follows an updateData with "
                    "nearby data (not the headline workload)")
    ap.add_argument("--warm", type=float, default=0.0, metavar="SHIFT", help="with --resolve: warm-start each solve from the "
                    "previous solution (extension, eicos_batch_set_warm_start); 0 = cold start as in the reference")
    ap.add_argument("--multi", default=None, metavar="IDS", help="ONE process, several devices through the product's own multi-GPU layer "
                    "(eicos_multi_* of include/eicos_amd.h; no torch.distributed): comma-separated device ids, one contiguous shard per entry; "
                    "a device may be listed twice (0,0 = two concurrent shards on one GPU).  Default total: 4096 (1024 on one distinct device)")
    ap.add_argument("--io", choices=("local", "root"), default="local", help="local: every rank regenerates its own shard "
                    "(no collective, default); root: rank 0 holds the whole batch and scatters shards over RCCL/xGMI "
                    "before the timed region, results are gathered back after it (times reported in config)")
    ap.add_argument("--legs-json", action="store_true", help="(internal) the GPU part of every workload of the default line on the library "
                    "EICOS_AMD_LIB names; used by the default run for config.summary.prev_round")
    ap.add_argument("--no-prev-round", action="store_true", help="skip the previous-round comparison legs (config.summary.prev_round)")
    args = ap.parse_args()
    if args.legs_json:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch  # noqa: F401
        args.perturb, args.resolve, args.warm = False, 0.0, 0.0
        legs_only(args)
        return

    # (the host driver of this pool supports dmabuf IPC only: RCCL / peer mappings across processes need it; already exported on the GPU boxes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # (first: torch brings its own HIP runtime; counting devices does not initialise a context on this image)
    import eicos_amd
    mode, multi_ids = launch_mode(args.gpus, os.environ, args.multi, torch.cuda.device_count)
    rank = int(os.environ.get("RANK", "0")) if mode == "dist" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if mode != "multi" else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if mode == "dist" else 1
    from eicos_amd.generate import shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU fallback")
    # REHEARSAL of the one-process-per-GPU launch on a box with fewer GPUs than ranks (EICOS_BENCH_REHEARSAL=1, never the default): the
    # ranks share the visible devices (local_rank mod their number) and the control-plane reductions go over gloo (RCCL refuses two ranks
    # on one device).  It exercises the launcher path end to end -- shard arithmetic, one handle per rank, the fences, the reductions, the
    # line rank 0 prints -- and labels the line as what it is: `n_gpus` = the DISTINCT devices used, `config.launch` says REHEARSAL.
    rehearsal = os.environ.get("EICOS_BENCH_REHEARSAL", "0") == "1" and mode == "dist"
    device_of_rank = local_rank % max(1, torch.cuda.device_count()) if rehearsal else local_rank
    torch.cuda.set_device(device_of_rank)
    dist = None
    ctl_device = None  # device of the control-plane tensors (None: the rank's GPU, reduced over RCCL)
    if mode == "dist" or (mode == "single" and "RANK" in os.environ and "MASTER_PORT" in os.environ):  # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ctl_device = torch.device("cpu")
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    local_rank = device_of_rank
    rank_devices = [r % max(1, torch.cuda.device_count()) for r in range(world)] if rehearsal else list(range(world))  # (one node: local rank = rank)

    if args.pattern == "dense-front":
        from eicos_amd.generate import dense_front_pattern
        pat, base = dense_front_pattern(2000, 32, 64)
        sets = [base]
    else:
        # a fixture name, or a path to a problem file (EPB1 container or an ECOS-style data header such as the
        # reference's data_MPC01.hpp, which is not in the mount: SURVEY.md F4) -- drop it in to bench the real thing
        path = args.pattern if os.path.exists(args.pattern) else os.path.join(ROOT, "tests", "golden", args.pattern + ".epb")
        pat, sets = eicos_amd.read_problem(path)

    # ---- which instances does this rank own? ----
    if multi_ids:
        ndist = len(set(multi_ids))
        total = args.total if args.total is not None else (args.batch * len(multi_ids) if args.batch is not None else (1024 if ndist == 1 else TOTAL_STRONG))
        B, first, scaling = total, 0, ("weak" if args.batch is not None else "strong")
    elif args.batch is not None:                     # weak: B per GPU
        B, first, scaling, total = args.batch, rank * args.batch, "weak", args.batch * world
    else:                                            # strong: fixed total in contiguous shards
        total = args.total if args.total is not None else (1024 if world == 1 else TOTAL_STRONG)
        first, B = shard_range(total, rank, world)
        scaling = "strong"
    default_workload = args.pattern == "MPC02" and not args.perturb and args.resolve == 0

    job = Job(args, pat, sets, first, B, local_rank, soc=args.soc, multi_ids=multi_ids)
    job.ctl_device = ctl_device
    io_ms = {}
    if args.io == "root" and rehearsal:
        raise SystemExit("--io root moves the shards over RCCL: not available in a rehearsal (two ranks on one device)")
    if args.io == "root" and dist is not None and world > 1:
        # the batch originates on rank 0's GPU: scatter the shards (outside the timed region: inputs are resident in
        # HBM when timing starts), results gathered back after it; equal shards only
        from eicos_amd.dist_io import KEYS, scatter_batch, gather_rows
        from eicos_amd.generate import SEED, feasible_batch, perturbed_batch
        assert total % world == 0, "--io root needs equal shards"
        widths = {k: job.data[k].shape[1] for k in KEYS}
        full = None
        if rank == 0:
            gen = perturbed_batch if args.perturb else feasible_batch
            allv = gen(job.pat, sets[0], 0, total, SEED)
            full = {k: torch.from_numpy(allv[k]).to(f"cuda:{local_rank}") for k in KEYS}
        dist.barrier(); torch.cuda.synchronize(); t_sc = time.perf_counter()
        dev = scatter_batch(full, widths, B, rank, world, f"cuda:{local_rank}", dist)
        torch.cuda.synchronize(); dist.barrier(); io_ms["scatter_ms"] = (time.perf_counter() - t_sc) * 1e3
        del full
        assert all(torch.equal(dev[k].cpu(), torch.from_numpy(job.data[k])) for k in KEYS), "scattered shard differs"
        job.devs[0] = dev

    res = job.run(dist, args.steps, args.warmup)

    if io_ms:  # results back to the root: x [B, n] straight from the instance slabs, then one gather
        torch.cuda.synchronize(); dist.barrier(); t_g = time.perf_counter()
        xl = torch.from_numpy(job.solver.solution()).to(f"cuda:{local_rank}")
        xall = gather_rows(xl, rank, world, dist)
        torch.cuda.synchronize(); dist.barrier(); io_ms["gather_ms"] = (time.perf_counter() - t_g) * 1e3
        if rank == 0:
            assert xall.shape == (total, job.dims["n"])

    # ---- the SOC variant of the same workload, timed the same way (its own solver handle; the LP one is released) ----
    soc_rep = None
    if default_workload and not args.soc and not args.no_soc and not multi_ids:
        main_rep = job.report(res, args.steps, f"MPC02 batch={B}") if rank == 0 else None
        cpu = job.cpu_baseline(job.ia_first) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
        job.solver.close()
        sjob = Job(args, pat, sets, first, B, local_rank, soc=True)
        sjob.ctl_device = ctl_device
        sres = sjob.run(dist, args.steps, args.warmup)
        if rank == 0:
            soc_rep = sjob.report(sres, args.steps, f"MPC02-SOC batch={B}")
            soc_rep["workload"] = (f"MPC-SOC variant: rows 3000.. of G regrouped into {sjob.dims['ncones']} second-order cones of dimension 3 "
                                   f"(l={sjob.pat.l}), same A/G values, generated strictly feasible (c,h,b), same instances and step")
            if world == 1 and not args.no_cpu_baseline:
                soc_rep["cpu_baseline"] = sjob.cpu_baseline(sjob.ia_first)
                soc_rep["refinement_vs_oracle"] = refinement_profile(sjob)
        sjob.solver.close()
    else:
        main_rep = job.report(res, args.steps, f"{args.pattern}{'-SOC' if args.soc else ''} batch={B}") if rank == 0 else None
        cpu = job.cpu_baseline(job.ia_first) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    if rank == 0:
        dims = job.dims
        roof = main_rep.pop("roofline")
        value, unit, ms_per_step = main_rep.pop("value"), main_rep.pop("unit"), main_rep.pop("ms_per_step")
        details = {"headline": {**main_rep, "value": value, "roofline": roof, "cpu_baseline": cpu}}
        # One compact entry per workload INSIDE `config` (the driver's record keeps `config` whole but only a tail of stdout):
        # value (iter/s), roofline fraction by algorithmic bytes (and with the passes a dual solve really makes), kernel ms per launch,
        # PMC traffic / algorithmic bytes (null without a hash-matched summary in profiles/), the launch's tail, CPU baseline and parity counts
        summary = {"headline": summarise({**main_rep, "value": value, "roofline": roof, "cpu_baseline": cpu})}
        if soc_rep is not None:
            details["soc"] = soc_rep
            summary["soc"] = summarise(soc_rep)
        if default_workload and world == 1 and not args.no_configs and not args.soc and args.batch is None and args.total is None and not multi_ids:
            # every other BASELINE.json config, driver-run: configs[4] dense-front (MFMA path), configs[3] LPnetlib (three of the
            # ten patterns: the smallest, a mid-size hybrid one, the deepest), the per-GPU share of configs[2] and north_star's
            # ">= 10x the host at batch 4096 on one GPU" configuration
            for k, kw in aux_legs():
                try:  # (an auxiliary leg must not cost the headline line: its failure is reported in its own entry)
                    v = run_config(args, kw.pop("name"), kw.pop("pattern"), kw.pop("batch"), local_rank, **kw)
                    details[k] = v
                    summary[k] = summarise(v)
                except Exception as e:  # noqa: BLE001
                    summary[k] = {"error": str(e)[:200]}
            try:  # (an auxiliary leg: a box that cannot pin memory -- locked-memory limit -- must not cost the headline line)
                he = host_e2e(pat, sets, B, local_rank, value)
                details["host_e2e"] = he
                r3 = lambda v: float(f"{v:.4g}")
                summary["host_e2e"] = {"batch": B, "in_MB": r3(he["bytes_in_per_step"] / 1e6), "out_MB": r3(he["bytes_out_per_step"] / 1e6),
                                       **{v: {"value": r3(he[v]["value"]), "x_device_resident": r3(he[v]["vs_device_resident"]), "update_ms": r3(he[v]["update_ms"]),
                                              "path": he[v]["update_path"]} for v in ("pageable", "registered", "pinned", "pinned_fused") if v in he}}
            except Exception as e:  # noqa: BLE001
                summary["host_e2e"] = {"error": str(e)[:200]}
            if not args.no_prev_round:
                pr = prev_round_legs(local_rank)
                if pr is not None:
                    details["prev_round"] = pr
                    summary["prev_round"] = pr
        out = {
            "metric": "ipm_iterations_per_sec", "value": value, "unit": unit,
            "n_gpus": (len(set(multi_ids)) if multi_ids else len(set(rank_devices))), "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{total} instances total in {len(multi_ids)} shard(s) over devices {sorted(set(multi_ids))} from one process ({scaling}), " if multi_ids else
                                    f"{total} instances total = {B}/GPU x {world} GPU ({scaling}), ") + f"{args.pattern}{'-SOC' if args.soc else ''} pattern "
                                   f"(n={dims['n']} m={dims['m']} p={dims['p']} cones={dims['ncones']}), "
                                   f"{'perturbed (c,h)' if args.perturb else 'strictly feasible generated (c,h,b)'}, updateData+solve per step",
                       "batch_per_gpu": (total // len(set(multi_ids)) if multi_ids else B), "total_instances": total,
                       "solves_per_sec": main_rep["solves_per_sec"], "optimal": main_rep["optimal"], "mean_iter": main_rep["mean_iter"],
                       "dim_K": dims["dim_K"], "nnzK": dims["nnzK"], "nnzL": dims["nnzL"], "levels": dims["nlevels"],
                       "threads_per_block": dims["threads_per_block"], "resident_blocks": dims["resident_blocks"], "kernel_build": dims["kernel_build"],
                       "kernel_ms_min_over_ranks": main_rep["kernel_ms_min_over_ranks"], "kernel_ms_max_over_ranks": main_rep["kernel_ms_max_over_ranks"],
                       # how the GPUs were reached, and the witnesses: ranks that reduced into the counters (RCCL world) / the shard device list
                       "launch": (f"REHEARSAL: {world} ranks of torch.distributed share {len(set(rank_devices))} device(s), counters over gloo (EICOS_BENCH_REHEARSAL=1)" if rehearsal else
                                  {"single": "one process, one GPU", "dist": "torch.distributed, one process per GPU (RCCL for counters only)",
                                   "multi": "one process, eicos_multi_* (no collective)"}[mode]),
                       "ranks_seen": res["ranks_seen"], "devices": (multi_ids if multi_ids else rank_devices),
                       "io": (f"one process, eicos_multi_* over devices {multi_ids} (shards {job.shards}), inputs resident per shard" if multi_ids else
                              "root scatter/gather over RCCL" if io_ms else "per-rank generation, no collective"), **io_ms,
                       **({"resolve_eps": args.resolve, "warm_shift": args.warm} if args.resolve > 0 else {}),
                       "summary": summary},
            "roofline": roof,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        # the full per-workload objects (dims, exit-code histograms, refinement profile of the SOC leg, ...) go to stderr and,
        # when the directory exists, to gpurun_out/bench_details.json: stdout carries exactly ONE json line
        try:
            sys.stderr.write("bench details: " + json.dumps(details) + "\n")
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                json.dump(details, open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "w"))
        except (OSError, TypeError, ValueError):
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
