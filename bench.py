#!/usr/bin/env python3
"""bench.py -- headline benchmark: IPM iterations/sec (fp64) on batched MPC-shaped SOCPs.

  python bench.py --gpus N --steps K --warmup W          (N=1 default)
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): per GPU a batch of 1024 instances on the
MPC02 sparsity pattern (the MPC01 blob named by BASELINE.json is missing from the reference
mount), strictly feasible (c,h,b) from eicos_amd.generate keyed by
(seed, global instance index).  One "step" = one pass of the hot path over the batch with the
raw inputs already resident in HBM: updateData (equilibrate + transposes, on device) followed
by the batched cold-start solve.  Metric value = sum over instances of Information.iter / time.
Multi-GPU: instances are independent -> each rank owns a contiguous shard (weak scaling,
1024 per GPU), no data-path collective; only the timing/iteration counters are reduced.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); 6290 measured achievable


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


def algorithmic_bytes(dims, ia):
    """Algorithmic HBM bytes of ONE solve launch (fp64 values only; shared index arrays excluded),
    SURVEY.md 8(d) formula with the measured counters of every instance."""
    n, p, m = dims["n"], dims["p"], dims["m"]
    N, nnzK, nnzL = dims["dim_K"], dims["nnzK"], dims["nnzL"]
    nnzAG = dims["nnzA"] + dims["nnzG"]
    f = ia["n_factor"].astype(np.float64).sum()
    r = ia["n_ldlsolve"].astype(np.float64).sum()
    it = (ia["iter"].astype(np.float64) + 1).sum()
    per_factor = nnzK + nnzL + N                 # read K values, write L and D
    per_solve = 2 * nnzL + 3 * N                 # L forward + L backward, D, rhs in / x out
    per_resid = 2 * nnzAG + 4 * N                # refinement residual: A,A',G,G' values + vectors
    per_iter = 2 * nnzAG + 6 * (n + p + 2 * m) + 30 * m + 6 * (n + p)  # residuals, scalings, RHS, line searches
    return 8.0 * (f * per_factor + r * (per_solve + per_resid) + it * per_iter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--pattern", default="MPC02", help="fixture name under tests/golden (BASELINE config 3: lp_*), a path to an EPB1 / ECOS data.h problem file, "
                    "or 'dense-front' (BASELINE config 4: n=2000, 32 cones x 64, generated)")
    ap.add_argument("--soc", action="store_true", help="MPC-SOC variant (332 cones of dim 3)")
    ap.add_argument("--perturb", action="store_true", help="LPnetlib-style batch: perturb c,h of the fixture "
                    "(SURVEY.md 8d config 4) instead of generating strictly feasible (c,h,b)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resolve", type=float, default=0.0, metavar="EPS", help="MPC-style re-solves: steps alternate between "
                    "the batch and a copy with c, h perturbed by EPS (relative), so every solve follows an updateData with "
                    "nearby data (not the headline workload)")
    ap.add_argument("--warm", type=float, default=0.0, metavar="SHIFT", help="with --resolve: warm-start each solve from the "
                    "previous solution (extension, eicos_batch_set_warm_start); 0 = cold start as in the reference")
    ap.add_argument("--io", choices=("local", "root"), default="local", help="local: every rank regenerates its own shard "
                    "(no collective, default); root: rank 0 holds the whole batch and scatters shards over RCCL/xGMI "
                    "before the timed region, results are gathered back after it (times reported in config)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import eicos_amd
    from eicos_amd.generate import SEED, feasible_batch, mpc_soc_variant

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):  # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.pattern == "dense-front":
        from eicos_amd.generate import dense_front_pattern
        pat, base = dense_front_pattern(2000, 32, 64)
        sets = [base]
    else:
        # a fixture name, or a path to a problem file (EPB1 container or an ECOS-style data header such as the
        # reference's data_MPC01.hpp, which is not in the mount: SURVEY.md F4) -- drop it in to bench the real thing
        path = args.pattern if os.path.exists(args.pattern) else os.path.join(ROOT, "tests", "golden", args.pattern + ".epb")
        pat, sets = eicos_amd.read_problem(path)
    if args.soc:
        pat = mpc_soc_variant(pat)
    B = args.batch
    first = rank * B  # weak scaling: every rank owns instances [rank*B, (rank+1)*B)
    if args.perturb:
        from eicos_amd.generate import perturbed_batch
        data = perturbed_batch(pat, sets[0], first, B, SEED)
    else:
        data = feasible_batch(pat, sets[0], first, B, SEED)
    io_ms = {}
    if args.io == "root" and dist is not None and world > 1:
        # the batch originates on rank 0's GPU: scatter the shards (outside the timed region: inputs are resident in
        # HBM when timing starts); every rank still knows its own data for the CPU cross-check below
        from eicos_amd.dist_io import KEYS, scatter_batch
        widths = {k: data[k].shape[1] for k in KEYS}
        full = None
        if rank == 0:
            gen = perturbed_batch if args.perturb else feasible_batch
            allv = gen(pat, sets[0], 0, world * B, SEED)
            full = {k: torch.from_numpy(allv[k]).to(f"cuda:{local_rank}") for k in KEYS}
        dist.barrier(); torch.cuda.synchronize(); t_sc = time.perf_counter()
        dev = scatter_batch(full, widths, B, rank, world, f"cuda:{local_rank}", dist)
        torch.cuda.synchronize(); dist.barrier(); io_ms["scatter_ms"] = (time.perf_counter() - t_sc) * 1e3
        del full
        assert all(torch.equal(dev[k].cpu(), torch.from_numpy(data[k])) for k in KEYS), "scattered shard differs"
    else:
        dev = {k: torch.from_numpy(v).to(f"cuda:{local_rank}") for k, v in data.items()}
    devs = [dev]
    if args.resolve > 0:
        rng = np.random.default_rng(SEED + 17 + rank)
        alt = dict(data)
        alt["c"] = data["c"] * (1 + args.resolve * rng.uniform(-1, 1, data["c"].shape))
        alt["h"] = data["h"] + args.resolve * (1 + np.abs(data["h"])) * rng.uniform(0, 1, data["h"].shape)
        devs.append({k: torch.from_numpy(v).to(f"cuda:{local_rank}") for k, v in alt.items()})
    step_no = [0]
    ptr = lambda k: (lambda t: t.data_ptr() if t.numel() else 0)(devs[step_no[0] % len(devs)][k])

    solver = eicos_amd.BatchSolver(pat, B, device=local_rank)
    dims = solver.dims()

    if args.warm > 0:
        solver.set_warm_start(args.warm)

    def step():
        solver.update_device(ptr("Gpr"), ptr("Apr"), ptr("c"), ptr("h"), ptr("b"))
        solver.solve_async()
        step_no[0] += 1

    def fence():
        if dist is not None:
            dist.barrier()
        solver.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # per-launch kernel duration from HIP events recorded on the solver's own stream
        kernel_ms.append(solver.last_solve_ms())
    fence()
    dt = time.perf_counter() - t0

    if io_ms:  # results back to the root: x [B, n] straight from the instance slabs, then one gather
        from eicos_amd.dist_io import gather_rows
        torch.cuda.synchronize(); dist.barrier(); t_g = time.perf_counter()
        xl = torch.from_numpy(solver.solution()).to(f"cuda:{local_rank}")
        xall = gather_rows(xl, rank, world, dist)
        torch.cuda.synchronize(); dist.barrier(); io_ms["gather_ms"] = (time.perf_counter() - t_g) * 1e3
        if rank == 0:
            assert xall.shape == (world * B, dims["n"])
    ia = solver.info_arrays()
    iters = int(ia["iter"].sum())
    ok = int((ia["exitcode"] == 0).sum())
    t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
    cnt = torch.tensor([iters, ok, B], dtype=torch.float64, device=f"cuda:{local_rank}")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    dt_max = float(t.item())
    tot_iters, tot_ok, tot_B = (int(v) for v in cnt.tolist())

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        value = tot_iters * args.steps / dt_max
        k_ms = float(np.mean(kernel_ms))
        abytes = algorithmic_bytes(dims, ia)
        achieved = abytes / (k_ms * 1e-3) / 1e9
        # HBM traffic of one launch: PMC counters cannot be collected inside a timed run, so the figure comes from
        # the committed summary of separate `rocprofv3 --pmc` passes over this same command (default workload only)
        traffic, traffic_src = None, None
        if args.pattern == "MPC02" and not args.soc and not args.perturb and B == 1024:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
                pm = json.load(open(f))
                if "traffic_bytes" in pm:
                    traffic, traffic_src = float(pm["traffic_bytes"]), os.path.relpath(f, ROOT)
                    break
        out = {
            "metric": "ipm_iterations_per_sec", "value": value, "unit": "iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU x {world} GPU, {args.pattern}{'-SOC' if args.soc else ''} pattern "
                                   f"(n={dims['n']} m={dims['m']} p={dims['p']} cones={dims['ncones']}), strictly feasible "
                                   f"generated (c,h,b), updateData+solve per step",
                       "batch_per_gpu": B, "dim_K": dims["dim_K"], "nnzK": dims["nnzK"], "nnzL": dims["nnzL"],
                       "levels": dims["nlevels"], "mean_iter": float(ia["iter"].mean()),
                       "mean_ldl_solves_per_iter": float(ia["n_ldlsolve"].sum() / max(1, ia["iter"].sum())),
                       "solves_per_sec": tot_B * args.steps / dt_max, "optimal": tot_ok, "instances": tot_B, "generator": "perturbed" if args.perturb else "feasible",
                       "threads_per_block": dims["threads_per_block"], "resident_blocks": dims["resident_blocks"],
                       "io": ("root scatter/gather over RCCL" if io_ms else "per-rank generation, no collective"), **io_ms,
                       **({"resolve_eps": args.resolve, "warm_shift": args.warm} if args.resolve > 0 else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_solve", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": abytes},
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (host cores are shared by all ranks)
            # CPU oracle (a port, NOT the EiCOS binary: Eigen is absent) on a bounded sample of the same workload
            from oracle import oracle as orc
            cores = usable_cores()
            sub = lambda k, a, b: data[k][a:b]
            run = lambda a, b: orc.batch_solve(pat, sub("Gpr", a, b), sub("Apr", a, b), sub("c", a, b), sub("h", a, b), sub("b", a, b), cores)
            # bounded sample: a pilot of one instance per core sizes the sample to ~15 s of CPU work (capped at four
            # passes over the batch), so that small and large patterns are both timed over a comparable span
            npil = int(min(B, cores))
            r0 = run(0, npil)
            per_inst_cpu = (r0["seconds"] + r0["update_seconds"]) * min(cores, npil) / npil
            want = int(max(npil, min(4 * B, 15.0 / max(per_inst_cpu, 1e-9))))
            reps, ns = (1, want) if want <= B else (min(4, -(-want // B)), B)
            tot_iters, wall, match = 0, 0.0, True
            for _ in range(reps):
                r = run(0, ns)
                tot_iters += int(r["iters"].sum()); wall += r["seconds"] + r["update_seconds"]
                match = match and bool(np.array_equal(r["iters"], ia["iter"][:ns]))
            out["cpu_baseline"] = {"value": float(tot_iters / wall), "unit": "iter/s", "cores": cores,
                                   "kind": "port", "sample": f"first {ns} instances of the same batch x {reps} pass(es), one "
                                   f"instance per thread at a time (updateData+solve), {wall:.2f}s wall = "
                                   f"{wall * cores:.0f} core-seconds",
                                   "iters_match_gpu": match,
                                   "per_core": float(tot_iters / wall / cores)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
