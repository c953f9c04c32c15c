"""ctypes mirror of include/eicos_amd.h (the C ABI of libeicos_amd.so).

No torch types cross the boundary: host numpy arrays or raw device pointers (ints) only.
If the HIP library is missing or no GPU is visible every compute call raises -- there is no
CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXIT_NAMES = {0: "optimal", 1: "primal_infeasible", 2: "dual_infeasible", -1: "maxit", -2: "numerics",
              -3: "outcone", -7: "fatal", 10: "close_to_optimal", 11: "close_to_primal_infeasible",
              12: "close_to_dual_infeasible", -87: "not_converged_yet"}


class Info(C.Structure):
    """struct eicos_info (mirror of EiCOS::Information, reference include/eicos.hpp:49-73)."""
    _fields_ = [(k, C.c_double) for k in (
        "pcost", "dcost", "pres", "dres", "gap", "relgap", "sigma", "mu", "step", "step_aff",
        "kapovert", "pinfres", "dinfres", "tau", "kap")] + [(k, C.c_int) for k in (
        "has_relgap", "has_pinfres", "has_dinfres", "pinf", "dinf", "iter", "nitref1", "nitref2",
        "nitref3", "exitcode", "n_factor", "n_ldlsolve", "n_sweep", "reserved_")] + [("solve_us", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Dims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n", "m", "p", "l", "ncones", "dim_K", "nnzA", "nnzG", "nnzK", "nnzL",
                                       "nlevels", "order_mode", "batch", "device")] + \
               [("factor_pairs", C.c_longlong), ("inst_bytes", C.c_size_t), ("work_bytes", C.c_size_t),
                ("pattern_bytes", C.c_size_t), ("threads_per_block", C.c_int), ("resident_blocks", C.c_int),
                ("lds_bytes", C.c_int), ("instances_per_block", C.c_int),
                ("lds_resident", C.c_int), ("factor_path", C.c_int), ("cone_order", C.c_int), ("dual_rhs", C.c_int),
                ("arithmetic_profile", C.c_int), ("apex_nodes", C.c_int), ("solo_slices", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def library_path() -> str:
    # EICOS_AMD_LIB: alternative build of the same library (used by tuning sweeps only)
    return os.environ.get("EICOS_AMD_LIB") or os.path.join(_HERE, "libeicos_amd.so")


def build_library(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    args = ["make", "-C", src] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return library_path()


def _lib():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        L = C.CDLL(path)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.eicos_last_error.restype = C.c_char_p
        L.eicos_batch_create.argtypes = [C.c_int] * 5 + [ip] * 5 + [C.c_int, C.c_int, C.POINTER(vp)]
        L.eicos_batch_update.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
        L.eicos_batch_update_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.eicos_batch_solve.argtypes = [vp, ip]
        L.eicos_batch_solve_async.argtypes = [vp]
        L.eicos_batch_sync.argtypes = [vp]
        L.eicos_batch_solution.argtypes = [vp, dp]
        L.eicos_batch_duals.argtypes = [vp, dp, dp, dp]
        L.eicos_batch_info.argtypes = [vp, C.POINTER(Info)]
        L.eicos_batch_solution_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.eicos_batch_dims.argtypes = [vp, C.POINTER(Dims)]
        L.eicos_batch_kernel_build.argtypes = [vp]
        L.eicos_batch_set_stream.argtypes = [vp, vp]
        L.eicos_batch_set_warm_start.argtypes = [vp, C.c_double]
        L.eicos_batch_set_warm_start.restype = C.c_int
        L.eicos_batch_set_dynamic_regularization.argtypes = [vp, C.c_double, C.c_double]
        L.eicos_batch_set_dynamic_regularization.restype = C.c_int
        L.eicos_batch_last_solve_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.eicos_batch_last_update_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.eicos_batch_destroy.argtypes = [vp]
        if hasattr(L, "eicos_batch_update_solve"):  # (round 6)
            L.eicos_batch_update_solve.argtypes = [vp, dp, dp, dp, dp, dp, dp, ip]
            L.eicos_batch_update_solve.restype = C.c_int
        if hasattr(L, "eicos_batch_ms_history"):  # (round 6; absent from a previous round's library)
            L.eicos_batch_ms_history.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int]
            L.eicos_batch_ms_history.restype = C.c_int
        if hasattr(L, "eicos_host_alloc"):  # absent from a previous round's library (EICOS_AMD_LIB: same-box A/B runs, bench.py's prev_round leg)
            L.eicos_batch_last_update_path.argtypes = [vp]
            L.eicos_batch_last_update_path.restype = C.c_int
            L.eicos_host_alloc.argtypes = [C.c_size_t]
            L.eicos_host_alloc.restype = vp
            L.eicos_host_free.argtypes = [vp]
            L.eicos_host_free.restype = C.c_int
            L.eicos_host_register.argtypes = [vp, C.c_size_t]
            L.eicos_host_register.restype = C.c_int
            L.eicos_host_unregister.argtypes = [vp]
            L.eicos_host_unregister.restype = C.c_int
        L.eicos_debug_factor.argtypes = [vp, C.c_int, dp, dp]
        L.eicos_debug_pattern.argtypes = [vp, ip, ip, ip]
        L.eicos_debug_trace.argtypes = [vp, C.c_int, dp]
        L.eicos_debug_kkt.argtypes = [vp, C.c_int, ip, ip, dp]
        L.eicos_debug_scalings.argtypes = [vp, C.c_int, dp, dp, dp, ip]
        L.eicos_debug_host_check.restype = C.c_double
        L.eicos_debug_host_check.argtypes = [C.c_int] * 4 + [ip] * 5 + [C.c_uint, C.c_int, ip]
        for fn in (L.eicos_debug_host_check_tiles, L.eicos_debug_host_check_hybrid):
            fn.restype = C.c_double
            fn.argtypes = [C.c_int] * 4 + [ip] * 5 + [C.c_uint, C.c_int, ip]
        for f in ("create", "update", "update_device", "solve", "solve_async", "sync", "solution", "duals", "info",
                  "solution_device", "dims", "set_stream", "last_solve_ms", "last_update_ms", "destroy"):
            getattr(L, "eicos_batch_" + f).restype = C.c_int
        # multi-GPU layer (one process, several devices)
        L.eicos_multi_last_error.restype = C.c_char_p
        L.eicos_multi_create.argtypes = [C.c_int] * 5 + [ip] * 5 + [C.c_int, ip, C.c_int, C.POINTER(vp)]
        L.eicos_multi_update.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
        L.eicos_multi_update_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        if hasattr(L, "eicos_multi_update_solve"):
            L.eicos_multi_update_solve.argtypes = [vp, dp, dp, dp, dp, dp, dp, ip]
            L.eicos_multi_update_solve.restype = C.c_int
        L.eicos_multi_solve.argtypes = [vp, ip]
        L.eicos_multi_solve_async.argtypes = [vp]
        L.eicos_multi_sync.argtypes = [vp]
        L.eicos_multi_solution.argtypes = [vp, dp]
        L.eicos_multi_duals.argtypes = [vp, dp, dp, dp]
        L.eicos_multi_info.argtypes = [vp, C.POINTER(Info)]
        L.eicos_multi_set_warm_start.argtypes = [vp, C.c_double]
        L.eicos_multi_set_dynamic_regularization.argtypes = [vp, C.c_double, C.c_double]
        L.eicos_multi_num_shards.argtypes = [vp]
        L.eicos_multi_shard.argtypes = [vp, C.c_int, C.POINTER(vp), ip, ip, ip]
        L.eicos_multi_last_solve_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.eicos_multi_destroy.argtypes = [vp]
        for f in ("create", "update", "update_device", "solve", "solve_async", "sync", "solution", "duals", "info", "set_warm_start",
                  "set_dynamic_regularization", "num_shards", "shard", "last_solve_ms", "destroy"):
            getattr(L, "eicos_multi_" + f).restype = C.c_int
        _LIB = L
    return _LIB


def set_arithmetic_profile(profile: int) -> None:
    """0 (default): plans shaped by the launch; 1: by the pattern alone -- batch- and shard-independent bits (eicos_set_arithmetic_profile)."""
    _chk(_lib().eicos_set_arithmetic_profile(int(profile)))


def device_count() -> int:
    return int(_lib().eicos_device_count())


def _chk(rc):
    if rc != 0:
        raise RuntimeError(f"eicos_amd error {rc}: {_lib().eicos_last_error().decode()}")


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class PinnedArray:
    """A float64 array in pinned host memory (eicos_host_alloc): `.a` is the numpy view.  updateData reads such arrays in place over PCIe
    (no bounce copy), solution() / duals() write into them with one strided copy.  Freed by close() or with the object."""

    def __init__(self, shape):
        shape = tuple(int(v) for v in (shape if hasattr(shape, "__len__") else (shape,)))
        n = int(np.prod(shape)) if shape else 1
        self._p = _lib().eicos_host_alloc(max(n, 1) * 8)
        if not self._p:
            raise RuntimeError("eicos_host_alloc failed: " + _lib().eicos_last_error().decode())
        self.a = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_double)), shape=(max(n, 1),))[:n].reshape(shape)

    def close(self):
        if getattr(self, "_p", None):
            self.a = None
            _lib().eicos_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_register(a):
    """Pin a C-contiguous float64 numpy array IN PLACE (eicos_host_register): updateData then reads it over PCIe without a bounce copy.
    Keep the array alive and call host_unregister(a) before it is freed."""
    assert a.dtype == np.float64 and a.flags.c_contiguous
    _chk(_lib().eicos_host_register(C.c_void_p(a.ctypes.data), a.nbytes))
    return a


def host_unregister(a):
    _chk(_lib().eicos_host_unregister(C.c_void_p(a.ctypes.data)))


UPDATE_PATHS = {0: "none", 1: "pinned bounce", 2: "pinned source in place", 3: "peer GPU in place", 4: "staged peer copies", 5: "fused into the solve", 6: "fused into the solve, staged while it runs"}


class BatchSolver:
    """One sparsity pattern, `batch` numeric instances on one GPU.

    Mirrors the reference's Solver surface (ctor / updateData / solve / solution / getInfo,
    reference include/eicos.hpp:137-163) with a leading batch dimension on every array.
    """

    def __init__(self, pat, batch: int, device: int = -1):
        L = _lib()
        self.pat, self.batch = pat, int(batch)
        self._keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (pat.q, pat.Gjc, pat.Gir, pat.Ajc, pat.Air)]
        q, Gjc, Gir, Ajc, Air = self._keep
        h = C.c_void_p()
        _chk(L.eicos_batch_create(pat.n, pat.m, pat.p, pat.l, pat.ncones, _ip(q) if pat.ncones else None,
                                  _ip(Gjc) if pat.m > 0 else None, _ip(Gir) if pat.m > 0 else None,
                                  _ip(Ajc) if pat.p > 0 else None, _ip(Air) if pat.p > 0 else None,
                                  self.batch, device, C.byref(h)))
        self._h = h

    # ---- updateData ----
    def update(self, Gpr=None, Apr=None, c=None, h=None, b=None, first: int = 0, count: int | None = None):
        """Host arrays shaped [count, ...]; None keeps the group (reference semantics)."""
        arrs = []
        for a in (Gpr, Apr, c, h, b):
            arrs.append(None if a is None else np.ascontiguousarray(a, dtype=np.float64))
        if count is None:
            count = next((a.shape[0] for a in arrs if a is not None and a.ndim == 2), self.batch)
        pat = self.pat
        for a, w in zip(arrs, (pat.nnzG, pat.nnzA, pat.n, pat.m, pat.p)):
            if a is not None and a.size != count * w:
                raise ValueError(f"array has {a.size} elements, expected {count}x{w}")
        # zero-width groups are passed as NULL-safe dummies
        ptr = [(_dp(a) if (a is not None and a.size) else (_dp(np.zeros(1)) if a is not None else None)) for a in arrs]
        _chk(_lib().eicos_batch_update(self._h, first, count, *ptr))

    def update_solve(self, Gpr=None, Apr=None, c=None, h=None, b=None, x_out=None):
        """updateData + solve in one call (eicos_batch_update_solve): host arrays shaped [batch, ...] (None keeps the group).  With pinned /
        registered arrays the solve kernel pulls every instance's inputs itself (no separate updateData launch) and writes x into a pinned
        `x_out`.  Returns the exit codes."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (Gpr, Apr, c, h, b)]
        pat = self.pat
        for a, w in zip(arrs, (pat.nnzG, pat.nnzA, pat.n, pat.m, pat.p)):
            if a is not None and a.size != self.batch * w:
                raise ValueError(f"array has {a.size} elements, expected {self.batch}x{w}")
        ptr = [(_dp(a) if (a is not None and a.size) else (_dp(np.zeros(1)) if a is not None else None)) for a in arrs]
        if x_out is not None:
            assert x_out.dtype == np.float64 and x_out.flags.c_contiguous and x_out.shape == (self.batch, pat.n)
        codes = np.zeros(self.batch, np.int32)
        _chk(_lib().eicos_batch_update_solve(self._h, *ptr, _dp(x_out) if (x_out is not None and x_out.size) else None, _ip(codes)))
        return codes

    def update_device(self, dG=0, dA=0, dc=0, dh=0, db=0, first: int = 0, count: int | None = None):
        """Raw device pointers (ints, e.g. torch.Tensor.data_ptr()); 0 keeps the group."""
        count = self.batch if count is None else count
        _chk(_lib().eicos_batch_update_device(self._h, first, count, *[C.c_void_p(int(p) or None) for p in (dG, dA, dc, dh, db)]))

    # ---- solve ----
    def solve(self):
        codes = np.zeros(self.batch, np.int32)
        _chk(_lib().eicos_batch_solve(self._h, _ip(codes)))
        return codes

    def solve_async(self):
        _chk(_lib().eicos_batch_solve_async(self._h))

    def sync(self):
        _chk(_lib().eicos_batch_sync(self._h))

    def set_warm_start(self, shift: float):
        """shift > 0: re-solves start from the previous solution (not in the reference; see include/eicos_amd.h)."""
        _chk(_lib().eicos_batch_set_warm_start(self._h, float(shift)))

    def set_dynamic_regularization(self, delta: float, eps: float):
        """delta > 0: ECOS-style dynamic regularisation of the LDL' pivots (not in the reference)."""
        _chk(_lib().eicos_batch_set_dynamic_regularization(self._h, float(delta), float(eps)))

    def set_stream(self, stream_ptr: int):
        _chk(_lib().eicos_batch_set_stream(self._h, C.c_void_p(int(stream_ptr) or None)))

    # ---- results ----
    def solution(self):
        x = np.zeros((self.batch, max(self.pat.n, 1)))
        if self.pat.n:
            x = np.zeros((self.batch, self.pat.n))
            _chk(_lib().eicos_batch_solution(self._h, _dp(x)))
            return x
        return x[:, :0]

    def duals(self):
        pat = self.pat
        y, z, s = np.zeros((self.batch, pat.p)), np.zeros((self.batch, pat.m)), np.zeros((self.batch, pat.m))
        _chk(_lib().eicos_batch_duals(self._h, _dp(y) if pat.p else None, _dp(z) if pat.m else None, _dp(s) if pat.m else None))
        return y, z, s

    def info(self):
        arr = (Info * self.batch)()
        _chk(_lib().eicos_batch_info(self._h, arr))
        return [arr[i].asdict() for i in range(self.batch)]

    def info_arrays(self):
        arr = (Info * self.batch)()
        _chk(_lib().eicos_batch_info(self._h, arr))
        raw = np.frombuffer(arr, dtype=np.dtype([(k, "f8" if t is C.c_double else "i4") for k, t in Info._fields_]))  # (declaration order)
        return {k: raw[k].copy() for k in raw.dtype.names}

    def dims(self) -> dict:
        d = Dims()
        _chk(_lib().eicos_batch_dims(self._h, C.byref(d)))
        return d.asdict()

    def kernel_build(self) -> str:
        """Which compilation of the solve kernel the handle launches: 'default', 'lds-resident', 'w2' (256 VGPRs, <= 2 workgroups per CU) or
        'u-in-lds' (one workgroup per CU, the factor operand array in LDS)."""
        v = _lib().eicos_batch_kernel_build(self._h)
        if v < 0:
            _chk(v)
        return ("default", "lds-resident", "w2", "u-in-lds")[v]

    def last_solve_ms(self) -> float:
        ms = C.c_float()
        _chk(_lib().eicos_batch_last_solve_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def last_update_ms(self) -> float:
        ms = C.c_float()
        _chk(_lib().eicos_batch_last_update_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def ms_history(self, which="solve", cap=64):
        """HIP-event durations (ms) of the most recent solve launches / updateData calls, oldest first (the handle's ring of 64 event pairs; which = "solve" | "update" | "step" = update start -> solve end):
        lets a caller time K back-to-back steps without synchronising inside its loop.  None with a library that predates the call."""
        L = _lib()
        if not hasattr(L, "eicos_batch_ms_history"):
            return None
        buf = (C.c_float * cap)()
        n = L.eicos_batch_ms_history(self._h, {"solve": 0, "update": 1, "step": 2}[which], buf, cap)
        if n < 0:
            _chk(n)
        return [float(buf[i]) for i in range(n)]

    def last_update_path(self) -> str:
        """How the most recent host-pointer / peer updateData moved its inputs (UPDATE_PATHS)."""
        return UPDATE_PATHS[_lib().eicos_batch_last_update_path(self._h)]

    def solution_into(self, x):
        """solution() into a caller-owned [batch, n] float64 array (e.g. a PinnedArray's `.a`: one strided device-to-host copy)."""
        assert x.dtype == np.float64 and x.flags.c_contiguous and x.shape == (self.batch, self.pat.n)
        if self.pat.n:
            _chk(_lib().eicos_batch_solution(self._h, _dp(x)))
        return x

    def debug_factor(self, inst: int = 0):
        d = self.dims()
        D, U = np.zeros(max(d["dim_K"], 1)), np.zeros(max(d["nnzL"], 1))
        _chk(_lib().eicos_debug_factor(self._h, inst, _dp(D), _dp(U)))
        return D[: d["dim_K"]], U[: d["nnzL"]]

    TRACE_COLS = ("pcost", "dcost", "gap", "pres", "dres", "kapovert", "mu", "step", "sigma", "tau", "kap", "nitref3")

    def debug_trace(self, inst: int = 0, iters: int | None = None):
        """Per-iteration history [iters+1, 12] of instance `inst` (needs batch <= resident workgroups)."""
        out = np.zeros((102, 12))
        _chk(_lib().eicos_debug_trace(self._h, inst, _dp(out)))
        return out if iters is None else out[: iters + 1]

    def debug_kkt(self, inst: int = 0):
        """Upper triangle of the instance's KKT matrix as the factorisation reads it: (rows, cols, vals)."""
        d = self.dims()
        r, c, v = np.zeros(max(d["nnzK"], 1), np.int32), np.zeros(max(d["nnzK"], 1), np.int32), np.zeros(max(d["nnzK"], 1))
        _chk(_lib().eicos_debug_kkt(self._h, inst, _ip(r), _ip(c), _dp(v)))
        return r[: d["nnzK"]], c[: d["nnzK"]], v[: d["nnzK"]]

    def debug_scalings(self, s, z, inst: int = 0):
        """The solver's updateScalings + updateKKTScalings stage for (s, z): (ran, scaling block of K)."""
        pat = self.pat
        nV = pat.l + int(sum(3 * int(q) + 1 for q in pat.q))
        s, z = np.ascontiguousarray(s, np.float64), np.ascontiguousarray(z, np.float64)
        V, ran = np.zeros(max(nV, 1)), np.zeros(1, np.int32)
        _chk(_lib().eicos_debug_scalings(self._h, inst, _dp(s), _dp(z), _dp(V), _ip(ran)))
        return bool(ran[0]), V[:nV]

    def debug_pattern(self):
        d = self.dims()
        perm, Lp, Li = np.zeros(max(d["dim_K"], 1), np.int32), np.zeros(d["dim_K"] + 1, np.int32), np.zeros(max(d["nnzL"], 1), np.int32)
        _chk(_lib().eicos_debug_pattern(self._h, _ip(perm), _ip(Lp), _ip(Li)))
        return perm[: d["dim_K"]], Lp, Li[: d["nnzL"]]

    def close(self):
        if getattr(self, "_h", None):
            _lib().eicos_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _mchk(rc):
    if rc != 0:
        raise RuntimeError(f"eicos_amd error {rc}: {_lib().eicos_multi_last_error().decode()}")


class MultiBatchSolver:
    """One sparsity pattern, `batch` instances in contiguous shards over `device_ids` (eicos_multi_* of include/eicos_amd.h):
    ONE process drives every listed GPU, one handle + stream per list entry, no collective.  Arrays are [batch, ...] in global
    instance order; a device may be listed more than once (its shards run concurrently on that GPU)."""

    def __init__(self, pat, batch: int, device_ids):
        L = _lib()
        self.pat, self.batch = pat, int(batch)
        self._keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (pat.q, pat.Gjc, pat.Gir, pat.Ajc, pat.Air)]
        q, Gjc, Gir, Ajc, Air = self._keep
        dev = np.ascontiguousarray(device_ids, dtype=np.int32)
        h = C.c_void_p()
        _mchk(L.eicos_multi_create(pat.n, pat.m, pat.p, pat.l, pat.ncones, _ip(q) if pat.ncones else None,
                                   _ip(Gjc) if pat.m > 0 else None, _ip(Gir) if pat.m > 0 else None,
                                   _ip(Ajc) if pat.p > 0 else None, _ip(Air) if pat.p > 0 else None,
                                   self.batch, _ip(dev), len(dev), C.byref(h)))
        self._h = h

    def update(self, Gpr=None, Apr=None, c=None, h=None, b=None, first: int = 0, count: int | None = None):
        """Host arrays shaped [count, ...]; None keeps the group (reference semantics)."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (Gpr, Apr, c, h, b)]
        if count is None:
            count = next((a.shape[0] for a in arrs if a is not None and a.ndim == 2), self.batch)
        ptr = [(_dp(a) if (a is not None and a.size) else (_dp(np.zeros(1)) if a is not None else None)) for a in arrs]
        _mchk(_lib().eicos_multi_update(self._h, first, count, *ptr))

    def update_device(self, src_device: int, dG=0, dA=0, dc=0, dh=0, db=0, first: int = 0, count: int | None = None):
        """Raw pointers into the HBM of GPU `src_device` (arrays [count, ...]); 0 keeps the group."""
        count = self.batch if count is None else count
        _mchk(_lib().eicos_multi_update_device(self._h, int(src_device), first, count, *[C.c_void_p(int(p) or None) for p in (dG, dA, dc, dh, db)]))

    def solve(self):
        codes = np.zeros(self.batch, np.int32)
        _mchk(_lib().eicos_multi_solve(self._h, _ip(codes)))
        return codes

    def solve_async(self):
        _mchk(_lib().eicos_multi_solve_async(self._h))

    def sync(self):
        _mchk(_lib().eicos_multi_sync(self._h))

    def solution(self):
        x = np.zeros((self.batch, self.pat.n))
        if self.pat.n:
            _mchk(_lib().eicos_multi_solution(self._h, _dp(x)))
        return x

    def duals(self):
        pat = self.pat
        y, z, s = np.zeros((self.batch, pat.p)), np.zeros((self.batch, pat.m)), np.zeros((self.batch, pat.m))
        _mchk(_lib().eicos_multi_duals(self._h, _dp(y) if pat.p else None, _dp(z) if pat.m else None, _dp(s) if pat.m else None))
        return y, z, s

    def info_arrays(self):
        arr = (Info * self.batch)()
        _mchk(_lib().eicos_multi_info(self._h, arr))
        raw = np.frombuffer(arr, dtype=np.dtype([(k, "f8" if t is C.c_double else "i4") for k, t in Info._fields_]))
        return {k: raw[k].copy() for k in raw.dtype.names}

    def shards(self):
        """[(first, count, device)] of every shard."""
        out = []
        for s in range(_lib().eicos_multi_num_shards(self._h)):
            hh, f, c, d = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
            _mchk(_lib().eicos_multi_shard(self._h, s, C.byref(hh), C.byref(f), C.byref(c), C.byref(d)))
            out.append((f.value, c.value, d.value))
        return out

    def shard_update_device(self, s: int, dG=0, dA=0, dc=0, dh=0, db=0):
        """updateData of shard s from raw pointers into the HBM of THAT shard's GPU (arrays [count of the shard, ...]): inputs resident
        on every device -- no copy at all (eicos_batch_update_device on the shard's handle)."""
        hh, c = C.c_void_p(), C.c_int()
        _mchk(_lib().eicos_multi_shard(self._h, s, C.byref(hh), None, C.byref(c), None))
        _chk(_lib().eicos_batch_update_device(hh, 0, c.value, *[C.c_void_p(int(p) or None) for p in (dG, dA, dc, dh, db)]))

    def shard_last_update(self, s: int):
        """(path, HIP-event ms) of shard s's most recent updateData."""
        hh = C.c_void_p()
        _mchk(_lib().eicos_multi_shard(self._h, s, C.byref(hh), None, None, None))
        ms = C.c_float()
        _chk(_lib().eicos_batch_last_update_ms(hh, C.byref(ms)))
        return UPDATE_PATHS[_lib().eicos_batch_last_update_path(hh)], float(ms.value)

    def shard_dims(self, s: int = 0) -> dict:
        hh = C.c_void_p()
        _mchk(_lib().eicos_multi_shard(self._h, s, C.byref(hh), None, None, None))
        d = Dims()
        _chk(_lib().eicos_batch_dims(hh, C.byref(d)))
        return d.asdict()

    def last_solve_ms(self):
        """(max over the shards, [per shard]) of the most recent solve's HIP-event duration."""
        n = _lib().eicos_multi_num_shards(self._h)
        mx, per = C.c_float(), (C.c_float * n)()
        _mchk(_lib().eicos_multi_last_solve_ms(self._h, C.byref(mx), per))
        return float(mx.value), [float(v) for v in per]

    def set_warm_start(self, shift: float):
        _mchk(_lib().eicos_multi_set_warm_start(self._h, float(shift)))

    def set_dynamic_regularization(self, delta: float, eps: float):
        _mchk(_lib().eicos_multi_set_dynamic_regularization(self._h, float(delta), float(eps)))

    def close(self):
        if getattr(self, "_h", None):
            _lib().eicos_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_check_tiles(pat, seed: int = 1, order_mode: int = -1, hybrid: bool = False):
    """Host-only check of the tile (dense-front) plan, or of the hybrid plan (scalar programs + tiles on the top block
    of the tree; residual -10 = the pattern does not qualify) -- no GPU: returns (relative residual, stats dict)."""
    q, Gjc, Gir, Ajc, Air = [np.ascontiguousarray(a, dtype=np.int32) for a in (pat.q, pat.Gjc, pat.Gir, pat.Ajc, pat.Air)]
    st = np.zeros(8, np.int32)
    fn = _lib().eicos_debug_host_check_hybrid if hybrid else _lib().eicos_debug_host_check_tiles
    r = fn(pat.n, pat.m, pat.p, pat.ncones, _ip(q) if pat.ncones else None,
                                            _ip(Gjc) if pat.m else None, _ip(Gir) if pat.m else None,
                                            _ip(Ajc) if pat.p else None, _ip(Air) if pat.p else None, seed, order_mode, _ip(st))
    keys = ("dim_K", "nnzK", "nnzL", "block_levels", "tile_pairs", "order_mode", "blocks", "tiles")
    return float(r), dict(zip(keys, (int(v) for v in st)))


def host_check(pat, seed: int = 1, order_mode: int = -1):
    """Host-only check of the symbolic analysis (no GPU): returns (relative residual, stats dict)."""
    q, Gjc, Gir, Ajc, Air = [np.ascontiguousarray(a, dtype=np.int32) for a in (pat.q, pat.Gjc, pat.Gir, pat.Ajc, pat.Air)]
    st = np.zeros(8, np.int32)
    r = _lib().eicos_debug_host_check(pat.n, pat.m, pat.p, pat.ncones, _ip(q) if pat.ncones else None,
                                      _ip(Gjc) if pat.m else None, _ip(Gir) if pat.m else None,
                                      _ip(Ajc) if pat.p else None, _ip(Air) if pat.p else None, seed, order_mode, _ip(st))
    keys = ("dim_K", "nnzK", "nnzL", "nlevels", "factor_pairs", "order_mode", "max_row", "max_col")
    return float(r), dict(zip(keys, (int(v) for v in st)))
