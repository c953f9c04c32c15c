"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package eicos_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_NATIVE = None


class OracleInfo(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "pcost", "dcost", "pres", "dres", "gap", "relgap", "sigma", "mu", "step", "step_aff",
        "kapovert", "pinfres", "dinfres", "tau", "kap")] + [(k, C.c_int) for k in (
        "has_relgap", "has_pinfres", "has_dinfres", "pinf", "dinf", "iter", "nitref1", "nitref2",
        "nitref3", "exitcode", "n_factor", "n_ldlsolve")]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "eicos_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_native() -> str | None:
    """The same oracle compiled on THIS machine with the reference's Release flags (`-O2 -march=native`, reference
    CMakeLists.txt:22) -- used by bench.py's cpu_baseline leg for the timing only.  Built where it runs (a -march=native
    object must not travel between machines: liboracle_native.so is git- and gpurun-ignored); None if the compiler is missing."""
    so = os.path.join(_HERE, "liboracle_native.so")
    src = os.path.join(_HERE, "eicos_oracle.cpp")
    tag = so + ".host"  # which CPU the object was built for: a copy that travelled here from another machine is rebuilt, never run
    try:
        import hashlib
        cpu = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
        sig = hashlib.sha256("".join(cpu).encode()).hexdigest()
        have = open(tag).read().strip() if os.path.exists(tag) else ""
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) or have != sig:
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            open(tag, "w").write(sig)
    except (OSError, subprocess.CalledProcessError):
        return None
    return so


def _prototype_batch(L):
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.oracle_batch_solve.restype = C.c_double
    L.oracle_batch_solve.argtypes = [C.c_int] * 4 + [ip] * 5 + [C.c_int] + [dp] * 5 + [C.c_int, ip, ip, dp, dp, dp,
                                                                              C.POINTER(C.c_longlong)]
    return L


def lib_native():
    """liboracle_native.so (only oracle_batch_solve is prototyped), or None when it cannot be built here."""
    global _LIB_NATIVE
    if _LIB_NATIVE is None:
        so = build_native()
        _LIB_NATIVE = _prototype_batch(C.CDLL(so)) if so else False
    return _LIB_NATIVE or None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int] * 5 + [ip, dp, ip, ip, dp, ip, ip, dp, dp, dp]
        L.oracle_update.argtypes = [C.c_void_p, dp, dp, dp, dp, dp]
        L.oracle_solve.argtypes = [C.c_void_p]
        L.oracle_solve.restype = C.c_int
        L.oracle_get_info.argtypes = [C.c_void_p, C.POINTER(OracleInfo)]
        L.oracle_get_x.argtypes = [C.c_void_p, dp]
        L.oracle_get_yzs.argtypes = [C.c_void_p, dp, dp, dp]
        L.oracle_get_dims.argtypes = [C.c_void_p, ip, ip, ip]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_warm_start.argtypes = [C.c_void_p, C.c_double]
        L.oracle_set_dynamic_regularization.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.oracle_get_trace.argtypes = [C.c_void_p, dp, C.c_int]
        L.oracle_get_trace.restype = C.c_int
        L.oracle_debug_scalings.argtypes = [C.c_void_p, dp, dp, dp]
        L.oracle_debug_scalings.restype = C.c_int
        L.oracle_debug_kkt.argtypes = [C.c_void_p, ip, ip, dp]
        L.oracle_debug_kkt.restype = C.c_int
        L.oracle_batch_solve.restype = C.c_double
        L.oracle_batch_solve.argtypes = [C.c_int] * 4 + [ip] * 5 + [C.c_int] + [dp] * 5 + [C.c_int, ip, ip, dp, dp, dp,
                                                                                  C.POINTER(C.c_longlong)]
        _LIB = L
    return _LIB


def _dp(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


class OracleSolver:
    """Mirror of the reference's raw-pointer Solver API (include/eicos.hpp:151-163)."""

    def __init__(self, pat, vals):
        self.pat = pat
        L = lib()
        self._keep = [np.ascontiguousarray(a) for a in (pat.q.astype(np.int32), pat.Gjc, pat.Gir, pat.Ajc, pat.Air)]
        q, Gjc, Gir, Ajc, Air = self._keep
        haveG = pat.m > 0
        haveA = pat.p > 0
        v = self._c(vals)
        self._h = L.oracle_create(
            pat.n, pat.m, pat.p, pat.l, pat.ncones, _ip(q) if pat.ncones else None,
            _dp(v.Gpr) if haveG else None, _ip(Gjc) if haveG else None, _ip(Gir) if haveG else None,
            _dp(v.Apr) if haveA else None, _ip(Ajc) if haveA else None, _ip(Air) if haveA else None,
            _dp(v.c), _dp(v.h), _dp(v.b))

    @staticmethod
    def _c(vals):
        import copy
        v = copy.copy(vals)
        for k in ("Gpr", "Apr", "c", "h", "b"):
            setattr(v, k, np.ascontiguousarray(getattr(vals, k), dtype=np.float64))
        return v

    def update(self, vals):
        v = self._c(vals)
        lib().oracle_update(self._h, _dp(v.Gpr), _dp(v.Apr), _dp(v.c), _dp(v.h), _dp(v.b))

    def solve(self) -> int:
        return lib().oracle_solve(self._h)

    def info(self) -> dict:
        o = OracleInfo()
        lib().oracle_get_info(self._h, C.byref(o))
        return o.asdict()

    def x(self):
        x = np.zeros(max(self.pat.n, 1))
        lib().oracle_get_x(self._h, _dp(x))
        return x[: self.pat.n]

    def yzs(self):
        y, z, s = np.zeros(max(self.pat.p, 1)), np.zeros(max(self.pat.m, 1)), np.zeros(max(self.pat.m, 1))
        lib().oracle_get_yzs(self._h, _dp(y), _dp(z), _dp(s))
        return y[: self.pat.p], z[: self.pat.m], s[: self.pat.m]

    def set_warm_start(self, shift: float):
        lib().oracle_set_warm_start(self._h, float(shift))

    def set_dynamic_regularization(self, delta: float, eps: float):
        lib().oracle_set_dynamic_regularization(self._h, float(delta), float(eps))

    def trace(self):
        out = np.zeros((102, 12))
        n = lib().oracle_get_trace(self._h, _dp(out), 102)
        return out[:n]

    def dims(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().oracle_get_dims(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(dimK=a.value, nnzK=b.value, nnzL=c.value)

    def debug_scalings(self, s, z):
        """updateScalings + updateKKTScalings for (s, z): (ok, scaling block of K in cacheIndices order)."""
        pat = self.pat
        nV = pat.l + int(sum(3 * int(d) + 1 for d in pat.q))
        s, z = np.ascontiguousarray(s, np.float64), np.ascontiguousarray(z, np.float64)
        V = np.zeros(max(nV, 1))
        ok = lib().oracle_debug_scalings(self._h, _dp(s), _dp(z), _dp(V))
        return bool(ok), V[:nV]

    def debug_kkt(self):
        """Upper triangle of the KKT matrix as it stands (CSC: ptr, idx, val), reference column layout."""
        d = self.dims()
        Kp, Ki, Kx = np.zeros(d["dimK"] + 1, np.int32), np.zeros(max(d["nnzK"], 1), np.int32), np.zeros(max(d["nnzK"], 1))
        nnz = lib().oracle_debug_kkt(self._h, _ip(Kp), _ip(Ki), _dp(Kx))
        return Kp, Ki[:nnz], Kx[:nnz]

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def batch_solve(pat, Gpr, Apr, c, h, b, nthreads: int, want_x: bool = False, native: bool = False):
    """CPU baseline: arrays [B][...]; returns dict with wall seconds of the solve phase.
    native=True: run the `-O2 -march=native` build of the same source (build_native) when it is available."""
    L = (lib_native() if native else None) or lib()
    B = c.shape[0] if pat.n else h.shape[0]
    Gpr, Apr, c, h, b = (np.ascontiguousarray(a, dtype=np.float64) for a in (Gpr, Apr, c, h, b))
    q = np.ascontiguousarray(pat.q.astype(np.int32))
    ex = np.zeros(B, np.int32)
    it = np.zeros(B, np.int32)
    pc = np.zeros(B)
    xo = np.zeros((B, pat.n)) if want_x else None
    upd = C.c_double()
    ns = C.c_longlong()
    dpn = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    wall = L.oracle_batch_solve(pat.n, pat.m, pat.p, pat.ncones, _ip(q), _ip(pat.Gjc), _ip(pat.Gir), _ip(pat.Ajc),
                                _ip(pat.Air), B, dpn(Gpr), dpn(Apr), dpn(c), dpn(h), dpn(b), nthreads, _ip(ex), _ip(it),
                                dpn(pc), dpn(xo) if want_x else None, C.byref(upd), C.byref(ns))
    return dict(seconds=wall, update_seconds=upd.value, exitcodes=ex, iters=it, pcost=pc, x=xo,
                ldlsolves=int(ns.value), native=bool(native and L is not lib()))
