"""Synthetic instance generator for batched benchmarks/tests.

Given a pattern and base values (A, G), every instance i draws a strictly feasible
primal-dual pair and derives (c, h, b) from it, so each instance is guaranteed OPTIMAL:
    x0 ~ N(0,1)^n ; y0 ~ N(0,1)^p ; s0, z0 strictly inside the cone
    (LP part U(0.5,2); SOC: tail ~ N(0,1), head = ||tail|| + U(0.5,2))
    h = G x0 + s0 ;  b = A x0 ;  c = -A' y0 - G' z0
Randomness: numpy Philox keyed by (seed, GLOBAL instance index) -- any shard of the batch
regenerates exactly its own instances, on any rank.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import csc_matrix

from .problem_io import Pattern, Values

SEED = 20261002


def _cone_point(rng, pat: Pattern):
    v = np.empty(pat.m)
    v[: pat.l] = rng.uniform(0.5, 2.0, pat.l)
    o = pat.l
    for d in pat.q:
        tail = rng.standard_normal(d - 1)
        v[o] = np.linalg.norm(tail) + rng.uniform(0.5, 2.0)
        v[o + 1: o + d] = tail
        o += d
    return v


def feasible_batch(pat: Pattern, base: Values, first: int, count: int, seed: int = SEED):
    """Return dict of arrays [count, ...]: Gpr, Apr (base values repeated), c, h, b."""
    G = csc_matrix((base.Gpr, pat.Gir, pat.Gjc), shape=(pat.m, pat.n))
    A = csc_matrix((base.Apr, pat.Air, pat.Ajc), shape=(pat.p, pat.n))
    c = np.empty((count, pat.n)); h = np.empty((count, pat.m)); b = np.empty((count, pat.p))
    for k in range(count):
        rng = np.random.Generator(np.random.Philox(key=[seed, first + k]))
        x0 = rng.standard_normal(pat.n)
        y0 = rng.standard_normal(pat.p)
        s0 = _cone_point(rng, pat)
        z0 = _cone_point(rng, pat)
        h[k] = G @ x0 + s0
        b[k] = A @ x0
        c[k] = -(A.T @ y0) - (G.T @ z0)
    return dict(Gpr=np.broadcast_to(base.Gpr, (count, pat.nnzG)).copy(),
                Apr=np.broadcast_to(base.Apr, (count, pat.nnzA)).copy(), c=c, h=h, b=b)


def perturbed_batch(pat: Pattern, base: Values, first: int, count: int, seed: int = SEED):
    """LPnetlib-style batch: keep A, G, b; c <- c(1+0.01 U(-1,1));
    h <- h + 0.01 (1+|h|) U(0,1) (pure relaxation); global instance 0 is unperturbed."""
    c = np.empty((count, pat.n)); h = np.empty((count, pat.m))
    for k in range(count):
        rng = np.random.Generator(np.random.Philox(key=[seed, first + k]))
        if first + k == 0:
            c[k], h[k] = base.c, base.h
        else:
            c[k] = base.c * (1 + 0.01 * rng.uniform(-1, 1, pat.n))
            h[k] = base.h + 0.01 * (1 + np.abs(base.h)) * rng.uniform(0, 1, pat.m)
    rep = lambda a, w: np.broadcast_to(a, (count, w)).copy()
    return dict(Gpr=rep(base.Gpr, pat.nnzG), Apr=rep(base.Apr, pat.nnzA), c=c, h=h, b=rep(base.b, pat.p))


def mpc_soc_variant(pat: Pattern, base: Values = None, rows_from: int = 3000, dim: int = 3):
    """'MPC-SOC' pattern: rows >= rows_from of an LP-only pattern are
    regrouped, in order, into second-order cones of size `dim` (row order unchanged)."""
    assert pat.ncones == 0 and (pat.m - rows_from) % dim == 0
    k = (pat.m - rows_from) // dim
    q = np.full(k, dim, np.int32)
    return Pattern(pat.n, pat.m, pat.p, rows_from, q, pat.Gjc, pat.Gir, pat.Ajc, pat.Air)


def dense_front_pattern(n: int = 2000, k: int = 32, d: int = 64, seed: int = SEED):
    """Dense-front SOCP: k cones of dimension d, p = 0, l = 0; cone i's d rows of G
    form a dense d x d N(0,1)/8 block on the d consecutive variables starting at floor(i (n-d)/(k-1))
    (neighbouring blocks overlap, together they cover all n columns).  Returns (Pattern, base Values)."""
    rng = np.random.Generator(np.random.Philox(key=[seed, 0xD5E]))
    rows, cols, vals = [], [], []
    for i in range(k):
        c0 = (i * (n - d)) // max(1, k - 1)
        blk = rng.standard_normal((d, d)) / 8
        rr, cc = np.meshgrid(np.arange(d), np.arange(d), indexing="ij")
        rows.append((i * d + rr).ravel()); cols.append((c0 + cc).ravel()); vals.append(blk.ravel())
    G = csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(k * d, n))
    G.sum_duplicates(); G.sort_indices()
    pat = Pattern(n, k * d, 0, 0, np.full(k, d, np.int32), G.indptr.astype(np.int32), G.indices.astype(np.int32),
                  np.zeros(n + 1, np.int32), np.zeros(0, np.int32))
    return pat, Values(G.data.copy(), np.zeros(0), np.zeros(n), np.zeros(k * d), np.zeros(0))


def random_socp_pattern(n: int, p: int, l: int, q, density: float = 0.3, seed: int = 0):
    """Random sparse SOCP pattern + base values with equality rows, LP rows and cones of sizes q
    (every variable appears in G so the problem is bounded in every direction of a feasible batch)."""
    rng = np.random.default_rng(seed)
    q = np.asarray(q, np.int32)
    m = l + int(q.sum())
    Gd = rng.standard_normal((m, n)) * (rng.random((m, n)) < density)
    for j in range(n):  # no empty columns
        if not Gd[:, j].any():
            Gd[rng.integers(m), j] = rng.standard_normal() + 1.5
    Ad = rng.standard_normal((p, n)) * (rng.random((p, n)) < max(density, 2.0 / max(n, 1)))
    for r in range(p):
        if not Ad[r].any():
            Ad[r, rng.integers(n)] = 1.0
    G, A = csc_matrix(Gd), csc_matrix(Ad)
    G.sort_indices(); A.sort_indices()
    pat = Pattern(n, m, p, l, q, G.indptr.astype(np.int32), G.indices.astype(np.int32),
                  A.indptr.astype(np.int32), A.indices.astype(np.int32))
    return pat, Values(G.data.copy(), A.data.copy(), np.zeros(n), np.zeros(m), np.zeros(p))


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [first, first+count) of `total` instances for `rank` of `world`."""
    base, rem = divmod(total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)
