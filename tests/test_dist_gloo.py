"""CPU, world_size 2, gloo: the multi-GPU path of bench.py is "shard the batch by contiguous
ranges, no data-path collective, reduce only counters/timing".  This test runs that exact
sharding + reduction logic on two CPU ranks, with the oracle standing in for the per-rank
solver (the GPU solver has no CPU mode), and checks the union equals the unsharded run."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_fixture


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eicos_amd.generate import feasible_batch, shard_range
    from oracle import oracle as orc
    pat, sets = load_fixture("update_data")
    first, count = shard_range(total, rank, world)
    d = feasible_batch(pat, sets[0], first, count)
    r = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], 1, want_x=True)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)          # stand-in for the per-rank step time
    cnt = torch.tensor([float(r["iters"].sum()), float((r["exitcodes"] == 0).sum()), float(count)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), first=first, count=count, x=r["x"], iters=r["iters"], t=t.numpy(), cnt=cnt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank(tmp_path):
    total, world = 7, 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    from eicos_amd.generate import feasible_batch
    from oracle import oracle as orc
    pat, sets = load_fixture("update_data")
    d = feasible_batch(pat, sets[0], 0, total)
    ref = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], 1, want_x=True)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert [int(p["first"]) for p in parts] == [0, 4] and [int(p["count"]) for p in parts] == [4, 3]
    x = np.concatenate([p["x"] for p in parts])
    assert np.array_equal(x, ref["x"])                       # shards regenerate exactly their own instances
    for p in parts:
        assert p["t"][0] == 1.5                              # MAX over ranks
        assert p["cnt"][0] == ref["iters"].sum() and p["cnt"][1] == total and p["cnt"][2] == total


def _io_worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eicos_amd.dist_io import KEYS, gather_rows, scatter_batch
    from eicos_amd.generate import feasible_batch
    pat, sets = load_fixture("update_data")
    widths = dict(Gpr=pat.nnzG, Apr=pat.nnzA, c=pat.n, h=pat.m, b=pat.p)
    full = None
    if rank == 0:  # the whole batch originates on the root
        d = feasible_batch(pat, sets[0], 0, world * B)
        full = {k: torch.from_numpy(d[k]) for k in KEYS}
    mine = scatter_batch(full, widths, B, rank, world, "cpu", dist)
    ref = feasible_batch(pat, sets[0], rank * B, B)           # what this rank would have generated locally
    ok = all(np.array_equal(mine[k].numpy(), ref[k]) for k in KEYS)
    res = torch.from_numpy(ref["c"] * (rank + 1.0))           # stand-in for per-instance results
    allres = gather_rows(res, rank, world, dist)
    if rank == 0:
        exp = np.concatenate([feasible_batch(pat, sets[0], r * B, B)["c"] * (r + 1.0) for r in range(world)])
        ok = ok and np.array_equal(allres.numpy(), exp)
    open(os.path.join(out_dir, f"io{rank}.txt"), "w").write("ok" if ok else "bad")
    dist.barrier()
    dist.destroy_process_group()


def test_root_scatter_and_gather_two_ranks(tmp_path):
    # the only collectives of the multi-GPU job: batch scatter from one rank, result gather (RCCL on GPUs, gloo here)
    world = 2
    mp.spawn(_io_worker, args=(world, _free_port(), 3, str(tmp_path)), nprocs=world, join=True)
    assert [open(tmp_path / f"io{r}.txt").read() for r in range(world)] == ["ok", "ok"]
