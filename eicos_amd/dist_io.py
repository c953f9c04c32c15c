"""Batch scatter / result gather for a multi-GPU job whose data originates on ONE rank.

The solve itself needs no exchange (instances are independent); these two collectives are the only traffic:
rank `src` holds the whole batch [world*B, ...] on its device and every rank receives its contiguous shard
(torch.distributed.scatter: RCCL over xGMI with backend "nccl", all 7 links of the root in parallel), and after the
solve the per-instance results are gathered back.  When the data originates on the host, per-device copies
(eicos_batch_update with host pointers, one PCIe link per GPU) need no collective at all -- bench.py's default
regenerates each shard locally instead.  torch is used for device memory and the process group only.
"""
from __future__ import annotations

KEYS = ("Gpr", "Apr", "c", "h", "b")


def scatter_batch(full, widths, B, rank, world, device, dist, src=0):
    """full: {key: tensor [world*B, width]} on `device` of rank `src` (ignored elsewhere); widths: {key: width}.
    Returns {key: tensor [B, width]} on every rank = rows [rank*B, (rank+1)*B) of the root's batch."""
    import torch
    out = {}
    for k in KEYS:
        w = int(widths[k])
        out[k] = torch.empty((B, w), dtype=torch.float64, device=device)
        if w == 0:
            continue
        chunks = None
        if rank == src:
            t = full[k]
            assert t.shape == (world * B, w) and t.dtype == torch.float64
            chunks = [t[r * B:(r + 1) * B].contiguous() for r in range(world)]
        dist.scatter(out[k], chunks, src=src)
    return out


def gather_rows(local, rank, world, dist, dst=0):
    """local: tensor [B, width] per rank -> tensor [world*B, width] on rank `dst` (None elsewhere)."""
    import torch
    parts = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local.contiguous(), parts, dst=dst)
    return torch.cat(parts) if rank == dst else None
