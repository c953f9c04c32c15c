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


def write_epb(path: str, pat: Pattern, sets) -> None:
    """Write (Pattern, [Values, ...]) in the EPB1 container."""
    with open(path, "wb") as f:
        f.write(b"EPB1")
        f.write(struct.pack("<8i", pat.n, pat.m, pat.p, pat.l, pat.ncones, pat.nnzG, pat.nnzA, len(sets)))
        for a in (pat.q, pat.Gjc, pat.Gir, pat.Ajc, pat.Air):
            f.write(np.asarray(a).astype("<i4").tobytes())
        for v in sets:
            for a in (v.Gpr, v.Apr, v.c, v.h, v.b):
                f.write(np.asarray(a).astype("<f8").tobytes())


# ---- ECOS "data.h" problem headers -----------------------------------------------------------------------
# The format of the reference's problem blobs (data_default.hpp / data_MPC01.hpp consumed by src/run.cpp:18-31,
# every header under test/, and the commented saveProblemData writer at src/eicos.cpp:2090-2162): C initialisers
#   idxint n = ..; idxint m, p, l, ncones;  idxint q[] = {..};  pfloat c[], h[], b[];
#   idxint Gjc[], Gir[]; pfloat Gpr[];  idxint Ajc[], Air[]; pfloat Apr[];
# optionally with a common name prefix (MPC02_n, lp_afiro_Gpr, ...) and `pfloat *Apr = NULL;` for absent groups.
_INIT_RE = None


def parse_c_initialisers(text: str) -> dict:
    """{name: int | float | ndarray | None} for every idxint/pfloat/int/double scalar or array initialiser."""
    import re
    global _INIT_RE
    if _INIT_RE is None:
        _INIT_RE = re.compile(r"(?:static\s+|const\s+)*(idxint|pfloat|int|double|long)\s+(\*?\w+)\s*(\[\s*\d*\s*\])?\s*=\s*"
                              r"(\{[^}]*\}|[^;{]+);", re.S)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for typ, name, arr, init in _INIT_RE.findall(text):
        init = init.strip()
        integer = typ in ("idxint", "int", "long")
        if name.startswith("*"):
            out[name[1:]] = None  # NULL pointer: group absent
            continue
        if init.startswith("{"):
            toks = [t for t in (s.strip() for s in init[1:-1].replace("\n", " ").split(",")) if t]
            out[name] = np.array([int(t) for t in toks], np.int32) if integer else np.array([float(t) for t in toks])
        elif arr == "":
            try:
                out[name] = int(init) if integer else float(init)
            except ValueError:
                pass
    return out


def read_ecos_header(path: str, prefix: str | None = None):
    """(Pattern, [Values]) from an ECOS-style data header.  `prefix` = common name prefix of the arrays; detected
    from the identifier that ends in `Gjc` (or `Ajc`) when not given."""
    d = parse_c_initialisers(open(path).read())
    if prefix is None:
        cands = [k[:-3] for k in d if k.endswith("Gjc")] or [k[:-3] for k in d if k.endswith("Ajc")]
        if not cands:
            raise ValueError(f"{path}: no Gjc/Ajc array found")
        prefix = min(cands, key=len)
    g = lambda k: d.get(prefix + k)
    n, m, p = (int(g(k) or 0) for k in ("n", "m", "p"))
    q = np.zeros(0, np.int32) if g("q") is None else np.asarray(g("q"), np.int32)[: int(g("ncones") or 0)]
    iarr = lambda a, k: np.zeros(k, np.int32) if a is None else np.asarray(a, np.int32)
    farr = lambda a, k: np.zeros(k) if a is None else np.asarray(a, np.float64)
    Gjc, Ajc = iarr(g("Gjc"), n + 1), iarr(g("Ajc"), n + 1)
    Gir, Air = iarr(g("Gir"), 0), iarr(g("Air"), 0)
    pat = Pattern(n, m, p, m - int(q.sum()), q, Gjc, Gir, Ajc, Air)  # l is derived, as in src/eicos.cpp:155
    v = Values(farr(g("Gpr"), Gir.size), farr(g("Apr"), Air.size), farr(g("c"), n), farr(g("h"), m), farr(g("b"), p))
    if not (len(Gjc) == n + 1 and len(Ajc) == n + 1 and Gjc[-1] == Gir.size == v.Gpr.size and Ajc[-1] == Air.size == v.Apr.size
            and v.c.size == n and v.h.size == m and v.b.size == p):
        raise ValueError(f"{path}: inconsistent dimensions (prefix '{prefix}')")
    return pat, [v]


def write_ecos_header(path: str, pat: Pattern, v: Values, prefix: str = "") -> None:
    """Write a problem as an ECOS-style data header (the layout of the reference's commented saveProblemData)."""
    def arr(typ, name, a, fmt):
        a = np.asarray(a)
        if a.size == 0:
            return f"{typ} *{prefix}{name} = NULL;\n"
        return f"{typ} {prefix}{name}[{a.size}] = {{" + ", ".join(fmt(x) for x in a) + "};\n"
    fi, ff = (lambda x: str(int(x))), (lambda x: repr(float(x)))
    with open(path, "w") as f:
        for k, val in (("n", pat.n), ("m", pat.m), ("p", pat.p), ("l", pat.l), ("ncones", pat.ncones)):
            f.write(f"idxint {prefix}{k} = {val};\n")
        f.write(arr("idxint", "q", pat.q, fi))
        f.write(arr("pfloat", "c", v.c, ff) + arr("pfloat", "h", v.h, ff) + arr("pfloat", "b", v.b, ff))
        f.write(arr("idxint", "Gjc", pat.Gjc if pat.nnzG else [], fi) + arr("idxint", "Gir", pat.Gir, fi) + arr("pfloat", "Gpr", v.Gpr, ff))
        f.write(arr("idxint", "Ajc", pat.Ajc if pat.nnzA else [], fi) + arr("idxint", "Air", pat.Air, fi) + arr("pfloat", "Apr", v.Apr, ff))


def read_problem(path: str):
    """EPB1 container or ECOS-style C header, by content."""
    with open(path, "rb") as f:
        magic = f.read(4)
    return read_epb(path) if magic == b"EPB1" else read_ecos_header(path)
