"""Problem containers and the EPB1 fixture format (little-endian).

EPB1 layout:  b"EPB1" | int32 n,m,p,l,ncones,nnzG,nnzA,nsets | int32 q[ncones] |
int32 Gjc[n+1] Gir[nnzG] Ajc[n+1] Air[nnzA] | nsets x float64 {Gpr[nnzG] Apr[nnzA] c[n] h[m] b[p]}.

The fields are the arguments of the reference's raw constructor
(reference include/eicos.hpp:151-154, i.e. the ECOS data.h layout that
src/run.cpp:18-31 consumes): G is m x n CSC, A is p x n CSC, the first l rows of G are
the LP cone, then the second-order cones q[0], q[1], ... in order.
"""
from __future__ import annotations

import dataclasses
import struct

import numpy as np


@dataclasses.dataclass
class Pattern:
    n: int
    m: int
    p: int
    l: int
    q: np.ndarray
    Gjc: np.ndarray
    Gir: np.ndarray
    Ajc: np.ndarray
    Air: np.ndarray

    @property
    def nnzG(self) -> int:
        return int(self.Gir.size)

    @property
    def nnzA(self) -> int:
        return int(self.Air.size)

    @property
    def ncones(self) -> int:
        return int(self.q.size)


@dataclasses.dataclass
class Values:
    Gpr: np.ndarray
    Apr: np.ndarray
    c: np.ndarray
    h: np.ndarray
    b: np.ndarray


def read_epb(path: str):
    """Return (Pattern, [Values, ...])."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"EPB1":
        raise ValueError(f"{path}: not an EPB1 file")
    n, m, p, l, nc, nnzG, nnzA, nsets = struct.unpack_from("<8i", raw, 4)
    off = 36

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off).copy()
        off += a.nbytes
        return a

    q = take("<i4", nc)
    Gjc, Gir = take("<i4", n + 1), take("<i4", nnzG)
    Ajc, Air = take("<i4", n + 1), take("<i4", nnzA)
    pat = Pattern(n, m, p, l, q, Gjc, Gir, Ajc, Air)
    sets = []
    for _ in range(nsets):
        sets.append(Values(take("<f8", nnzG), take("<f8", nnzA), take("<f8", n), take("<f8", m), take("<f8", p)))
    assert off == len(raw), (off, len(raw))
    return pat, sets
