"""ctypes binding of libdesc_amd.so -- the C ABI declared in include/desc_amd.h.

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible when a solver handle is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DESC_AMD_LIB", os.path.join(HERE, "libdesc_amd.so"))   # override: diagnostic builds only

DESC_OK = 0
STEP_CONSTANT, STEP_PIECEWISE, STEP_HYBRID = 0, 1, 2
BUILD_HOST, BUILD_DEVICE = 0, 1

# every symbol include/desc_amd.h declares (tests check that the library exports them)
EXPORTS = [
    "desc_last_error", "desc_version", "desc_device_count", "desc_problem_upload", "desc_problem_free",
    "desc_spectral_run_dev", "desc_gcw_run_dev", "desc_cemp_run_dev", "desc_refine_run_dev", "desc_pgd_create_dev",
    "desc_structure_build", "desc_structure_import", "desc_structure_get", "desc_structure_sizes",
    "desc_structure_host_exports", "desc_structure_free",
    "desc_sample_key", "desc_params_default",
    "desc_pgd_create", "desc_pgd_destroy", "desc_pgd_run", "desc_pgd_run_traced", "desc_pgd_reset", "desc_pgd_iterate",
    "desc_pgd_iterate_timed", "desc_pgd_sync", "desc_pgd_download", "desc_pgd_get_s0",
    "desc_pgd_sizes", "desc_pgd_layout_stats", "desc_pgd_kernel_name", "desc_pgd_solve", "desc_selftest_group_sum",
    "desc_pgd_create_shard", "desc_pgd_shard_info", "desc_pgd_shard_bind", "desc_pgd_shard_colsum", "desc_pgd_shard_sweep",
    "desc_pgd_shard_finish", "desc_pgd_shard_objective", "desc_pgd_shard_set_collectives", "desc_pgd_shard_start",
    "desc_pgd_shard_iterate", "desc_pgd_shard_run", "desc_pgd_stopped", "desc_device_synchronize", "desc_memcpy_d2h", "desc_memcpy_h2d", "desc_debug_band_plan", "desc_debug_spmm_variants", "desc_debug_wg_clock", "desc_debug_wg_plan", "desc_debug_last_sweep", "desc_debug_shard_layout", "desc_trim_memory", "desc_spectral_run", "desc_cemp_run", "desc_refine_run",
    "desc_marshal_edges", "desc_marshal_rij",
]

I32P = C.POINTER(C.c_int32)
I64P = C.POINTER(C.c_int64)
F64P = C.POINTER(C.c_double)


class Problem(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("ind_i", I32P), ("ind_j", I32P), ("rij", F64P)]


class StructureView(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("m_pos", C.c_int64), ("m_cycle", C.c_int64),
                ("n_sample", C.c_int32), ("max_cnt", C.c_int32),
                ("codeg", I32P), ("pos_edge", I32P), ("cum_ind", I64P),
                ("k", I32P), ("e_jk", I32P), ("e_ki", I32P), ("ikj", I32P), ("jki", I32P)]


class StructureInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("m_pos", C.c_int64), ("m_cycle", C.c_int64),
                ("n_sample", C.c_int32), ("max_cnt", C.c_int32), ("built_where", C.c_int32),
                ("host_resident", C.c_int32), ("ms_build", C.c_double)]


class Params(C.Structure):
    _fields_ = [("iters", C.c_int32), ("step_kind", C.c_int32), ("lr", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("decay_interval", C.c_double),
                ("hybrid_strategy", C.c_int32), ("t0", C.c_int32), ("patience", C.c_int32),
                ("stop_tol", C.c_double), ("n_sample_min", C.c_int32), ("seed", C.c_uint64),
                ("verbose", C.c_int32), ("device", C.c_int32), ("build_where", C.c_int32),
                ("check_every", C.c_int32), ("progress", C.c_void_p), ("progress_user", C.c_void_p)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_double, C.c_double)


class Result(C.Structure):
    _fields_ = [("s_vec", F64P), ("obj_trace", F64P), ("avg_change_trace", F64P), ("w", F64P),
                ("adam_m", F64P), ("adam_v", F64P), ("iters_run", C.c_int32), ("t_end", C.c_int32),
                ("ms_structure", C.c_double), ("ms_upload", C.c_double), ("ms_cycle_d", C.c_double),
                ("ms_pgd", C.c_double), ("ms_total", C.c_double)]


class ShardInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("t_len", C.c_int64), ("t_part", C.c_int64), ("slice_len", C.c_int64),
                ("seg_lo", C.c_int64), ("seg_hi", C.c_int64), ("cyc_lo", C.c_int64), ("cyc_hi", C.c_int64),
                ("m_pos", C.c_int64), ("m_cycle", C.c_int64), ("xparts", C.c_int32), ("reserved", C.c_int32)]


RS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)     # ncclReduceScatter
AG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)              # ncclAllGather


class Collectives(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("reduce_scatter", C.c_void_p), ("all_gather", C.c_void_p)]


class SpectralInfo(C.Structure):
    _fields_ = [("iters", C.c_int32), ("products", C.c_int32), ("converged", C.c_int32), ("reserved", C.c_int32), ("residual", C.c_double),
                ("eigenvalues", C.c_double * 3), ("ms_total", C.c_double)]


class RefineInfo(C.Structure):
    _fields_ = [("iters", C.c_int32), ("cg_iters", C.c_int32), ("verbose", C.c_int32), ("cg_unconverged", C.c_int32),
                ("score", C.c_double), ("ms_total", C.c_double), ("cg_residual", C.c_double)]


ERR_INVALID, ERR_HIP, ERR_TOO_LARGE, ERR_STATE = -1, -2, -3, -4


class DescError(RuntimeError):
    """Carries the C return code (``code``): callers branch on it, not on the message text."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


_lib = None


def load():
    """Load libdesc_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DescError(
            f"{LIB_PATH} is missing: build it with `python -m desc_amd.build` (hipcc, gfx950). "
            "The DESC_PGD hot path has no CPU fallback.")
    # PyTorch bundles a ROCm runtime with the same sonames as /opt/rocm; a process can hold only
    # one copy and torch needs its own.  Callers that use torch next to this library (the
    # multi-GPU driver) import torch first, or set DESC_AMD_PRELOAD_TORCH=1.
    if os.environ.get("DESC_AMD_PRELOAD_TORCH") == "1":
        import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.desc_last_error.restype = C.c_char_p
    L.desc_version.restype = C.c_char_p
    L.desc_pgd_kernel_name.restype = C.c_char_p
    L.desc_pgd_kernel_name.argtypes = [C.c_void_p]
    L.desc_sample_key.restype = C.c_uint64
    L.desc_sample_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.desc_params_default.argtypes = [C.POINTER(Params)]
    L.desc_params_default.restype = None
    L.desc_structure_build.argtypes = [C.POINTER(Problem), C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                       C.POINTER(C.c_void_p)]
    L.desc_structure_import.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32, I32P, I64P, I32P, I32P,
                                        I32P, I32P, I32P, C.POINTER(C.c_void_p)]
    L.desc_structure_get.argtypes = [C.c_void_p, C.POINTER(StructureView)]
    L.desc_structure_sizes.argtypes = [C.c_void_p, C.POINTER(StructureInfo)]
    L.desc_structure_host_exports.restype = C.c_int64
    L.desc_structure_host_exports.argtypes = []
    L.desc_structure_free.argtypes = [C.c_void_p]
    L.desc_structure_free.restype = None
    L.desc_pgd_create.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.desc_pgd_destroy.argtypes = [C.c_void_p]
    L.desc_pgd_destroy.restype = None
    L.desc_pgd_run.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Result)]
    L.desc_pgd_run_traced.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), F64P, C.c_double, C.c_int32, F64P, F64P, C.POINTER(Result)]
    L.desc_pgd_reset.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.desc_pgd_iterate.argtypes = [C.c_void_p, C.c_int32]
    L.desc_pgd_iterate_timed.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.desc_pgd_sync.argtypes = [C.c_void_p]
    L.desc_pgd_download.argtypes = [C.c_void_p, C.POINTER(Result)]
    L.desc_pgd_get_s0.argtypes = [C.c_void_p, F64P]
    L.desc_pgd_sizes.argtypes = [C.c_void_p, I64P, I64P, I64P, I32P]
    L.desc_pgd_solve.argtypes = [C.POINTER(Problem), C.POINTER(Params), C.POINTER(Result)]
    L.desc_marshal_edges.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, I32P, I32P, I64P, I32P]
    L.desc_marshal_rij.argtypes = [F64P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, I64P, F64P]
    L.desc_pgd_create_shard.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.desc_pgd_shard_info.argtypes = [C.c_void_p, C.POINTER(ShardInfo)]
    L.desc_pgd_shard_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.desc_pgd_shard_colsum.argtypes = [C.c_void_p]
    L.desc_pgd_shard_sweep.argtypes = [C.c_void_p]
    L.desc_pgd_shard_finish.argtypes = [C.c_void_p, C.c_int32]
    L.desc_pgd_shard_objective.argtypes = [C.c_void_p, C.c_int32]
    L.desc_pgd_stopped.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.desc_pgd_shard_set_collectives.argtypes = [C.c_void_p, C.POINTER(Collectives)]
    L.desc_pgd_shard_start.argtypes = [C.c_void_p, C.POINTER(Params)]
    L.desc_pgd_shard_iterate.argtypes = [C.c_void_p, C.c_int32]
    L.desc_pgd_shard_run.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(Result)]
    L.desc_debug_band_plan.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_int32, C.c_int32, C.c_int32, I64P]
    L.desc_debug_spmm_variants.argtypes = [C.c_void_p, C.c_int32, F64P]
    L.desc_debug_wg_clock.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int32]
    L.desc_debug_wg_plan.argtypes = [C.c_void_p, I64P, C.c_int32]
    L.desc_pgd_layout_stats.argtypes = [C.c_void_p, I64P, C.c_int32]
    L.desc_debug_last_sweep.restype = C.c_char_p
    L.desc_debug_last_sweep.argtypes = [C.c_void_p]
    L.desc_debug_shard_layout.argtypes = [C.c_void_p, I32P, I32P, I32P, I32P]
    L.desc_trim_memory.restype = C.c_int64
    L.desc_trim_memory.argtypes = []
    L.desc_device_synchronize.argtypes = [C.c_int32]
    L.desc_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.desc_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.desc_spectral_run.argtypes = [C.POINTER(Problem), F64P, C.c_int32, C.c_double, C.c_int32, C.c_int32, F64P,
                                    C.POINTER(SpectralInfo)]
    L.desc_problem_upload.argtypes = [C.POINTER(Problem), C.c_int32, C.POINTER(C.c_void_p)]
    L.desc_problem_free.argtypes = [C.c_void_p]
    L.desc_problem_free.restype = None
    L.desc_spectral_run_dev.argtypes = [C.c_void_p, F64P, C.c_int32, C.c_double, C.c_int32, F64P, C.POINTER(SpectralInfo)]
    L.desc_gcw_run_dev.argtypes = [C.c_void_p, F64P, C.c_double, C.c_int32, F64P, C.POINTER(SpectralInfo)]
    L.desc_cemp_run_dev.argtypes = [C.c_void_p, F64P, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, F64P, C.POINTER(C.c_double)]
    L.desc_refine_run_dev.argtypes = [C.c_void_p, F64P, F64P, C.c_double, C.c_int32, F64P, C.POINTER(RefineInfo)]
    L.desc_pgd_create_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.desc_cemp_run.argtypes = [C.POINTER(Problem), F64P, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, F64P,
                                C.POINTER(C.c_double)]
    L.desc_refine_run.argtypes = [C.POINTER(Problem), F64P, F64P, C.c_double, C.c_int32, C.c_int32, F64P, C.POINTER(RefineInfo)]
    _lib = L
    return L


# ---- DESC_DEBUG_GUARD=1 (diagnostics; tests/conftest.py turns it on): every buffer this module hands to the library to be
# written is fenced by 64 bytes of guard words on both sides, and every fence still alive is verified after each native call --
# a write past a caller buffer (or a late write into an older one) is reported at the call that made it.
GUARD = os.environ.get("DESC_DEBUG_GUARD") == "1"
_FENCE = 64
_FENCE_BYTE = 0xA5
_fences = []          # weak references to the fenced base arrays


def out_buffer(count, dtype=np.float64):
    """Zeroed output buffer for the library to fill (count >= 1 elements)."""
    count = max(int(count), 1)
    if not GUARD:
        return np.zeros(count, dtype=dtype)
    import weakref
    nbytes = count * np.dtype(dtype).itemsize
    base = np.zeros(nbytes + 2 * _FENCE, dtype=np.uint8)
    base[:_FENCE] = _FENCE_BYTE
    base[_FENCE + nbytes:] = _FENCE_BYTE
    _fences.append(weakref.ref(base))
    return base[_FENCE:_FENCE + nbytes].view(dtype)


def verify_guards():
    """Check the fences of every live guarded buffer; raises DescError(ERR_STATE) naming the damaged side."""
    if not _fences:
        return
    live = []
    for r in _fences:
        base = r()
        if base is None:
            continue
        live.append(r)
        lo, hi = base[:_FENCE], base[base.size - _FENCE:]
        if (lo != _FENCE_BYTE).any() or (hi != _FENCE_BYTE).any():
            side = "before" if (lo != _FENCE_BYTE).any() else "after"
            raise DescError(f"DESC_DEBUG_GUARD: native code wrote {side} a caller buffer of {base.size - 2 * _FENCE} bytes", ERR_STATE)
    _fences[:] = live


def check(rc):
    if GUARD:
        verify_guards()
    if rc != DESC_OK:
        raise DescError(f"desc_amd error {rc}: {load().desc_last_error().decode()}", rc)


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def default_params():
    p = Params()
    load().desc_params_default(C.byref(p))
    return p


class ProblemArrays:
    """Keeps the NumPy buffers behind a desc_problem alive."""

    def __init__(self, n, ind_i, ind_j, rij=None):
        self.ind_i = np.ascontiguousarray(ind_i, dtype=np.int32)
        self.ind_j = np.ascontiguousarray(ind_j, dtype=np.int32)
        self.rij = None if rij is None else np.ascontiguousarray(rij, dtype=np.float64).reshape(-1)
        m = self.ind_i.shape[0]
        self.n, self.m = int(n), int(m)
        if self.rij is not None and self.rij.shape[0] != 9 * m:
            raise ValueError("rij must hold m*9 doubles")
        self.c = Problem(int(n), m, ptr(self.ind_i, I32P), ptr(self.ind_j, I32P), ptr(self.rij, F64P))


class DeviceProblem:
    """Owner of a desc_device_problem*: Ind / RijMat / CSR index resident in HBM, shared by every stage of DESC()."""

    def __init__(self, prob: ProblemArrays, device=0):
        h = C.c_void_p()
        check(load().desc_problem_upload(C.byref(prob.c), device, C.byref(h)))
        self.handle, self.prob, self.device = h, prob, device
        self.n, self.m = prob.n, prob.m

    def free(self):
        if self.handle:
            load().desc_problem_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _view_arrays(v: StructureView):
    def arr(p, count, dt):
        if count == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(p, shape=(count,)).copy()

    return dict(n=v.n, m=v.m, m_pos=v.m_pos, m_cycle=v.m_cycle, n_sample=v.n_sample, max_cnt=v.max_cnt,
                codeg=arr(v.codeg, v.m, np.int32), pos_edge=arr(v.pos_edge, v.m_pos, np.int32),
                cum_ind=arr(v.cum_ind, v.m_pos + 1, np.int64), k=arr(v.k, v.m_cycle, np.int32),
                e_jk=arr(v.e_jk, v.m_cycle, np.int32), e_ki=arr(v.e_ki, v.m_cycle, np.int32),
                ikj=arr(v.ikj, v.m_cycle, np.int32), jki=arr(v.jki, v.m_cycle, np.int32))


class Structure:
    """Owner of a desc_structure*."""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def build(cls, prob: ProblemArrays, n_sample_min=30, seed=0, where=BUILD_HOST, device=0):
        h = C.c_void_p()
        check(load().desc_structure_build(C.byref(prob.c), n_sample_min, seed, where, device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, n, m, n_sample, pos_edge, cum_ind, k, e_jk, e_ki, ikj, jki):
        a = [np.ascontiguousarray(x, dtype=np.int32) for x in (pos_edge, k, e_jk, e_ki, ikj, jki)]
        cum = np.ascontiguousarray(cum_ind, dtype=np.int64)
        h = C.c_void_p()
        check(load().desc_structure_import(int(n), int(m), a[0].shape[0], int(n_sample), ptr(a[0], I32P),
                                           ptr(cum, I64P), ptr(a[1], I32P), ptr(a[2], I32P), ptr(a[3], I32P),
                                           ptr(a[4], I32P), ptr(a[5], I32P), C.byref(h)))
        return cls(h)

    def arrays(self):
        v = StructureView()
        check(load().desc_structure_get(self.handle, C.byref(v)))
        return _view_arrays(v)

    def sizes(self):
        """O(1): never exports a device-built structure to the host (desc_structure_sizes)."""
        v = StructureInfo()
        check(load().desc_structure_sizes(self.handle, C.byref(v)))
        return dict(n=v.n, m=v.m, m_pos=v.m_pos, m_cycle=v.m_cycle, n_sample=v.n_sample, max_cnt=v.max_cnt,
                    built_where=v.built_where, host_resident=bool(v.host_resident), ms_build=v.ms_build)

    def free(self):
        if self.handle:
            load().desc_structure_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Solver:
    """Owner of a desc_pgd* (problem + structure resident in HBM)."""

    def __init__(self, prob, structure: Structure, device=0, rank=0, world=1):
        h = C.c_void_p()
        if isinstance(prob, DeviceProblem):        # rotations and edge list already in HBM
            check(load().desc_pgd_create_dev(prob.handle, structure.handle, rank, world, C.byref(h)))
        else:
            check(load().desc_pgd_create_shard(C.byref(prob.c), structure.handle, device, rank, world, C.byref(h)))
        self.handle = h
        m, mp, mc, mx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        check(load().desc_pgd_sizes(h, C.byref(m), C.byref(mp), C.byref(mc), C.byref(mx)))
        self.m, self.m_pos, self.m_cycle, self.max_cnt = m.value, mp.value, mc.value, mx.value

    def kernel_name(self):
        return load().desc_pgd_kernel_name(self.handle).decode()

    def layout_stats(self):
        out = np.zeros(6, dtype=np.int64)
        k = load().desc_pgd_layout_stats(self.handle, ptr(out, I64P), 6)
        if k < 0:
            check(k)
        return dict(zip(("colsum_entries", "pieces", "bands", "piece_row_entries", "cycles", "segments"), (int(x) for x in out[:k])))

    def last_sweep(self):
        """Name + template arguments of the sweep kernel launched last (diagnostics)."""
        return load().desc_debug_last_sweep(self.handle).decode()

    def shard_layout(self):
        """The exchange layout (diagnostics): xpos, spos (2m each), xt and slot_ab ((owned segments, 2) each)."""
        info = self.shard_info()
        nsl = int(info.seg_hi - info.seg_lo)
        xpos, spos = out_buffer(2 * self.m, np.int32), out_buffer(2 * self.m, np.int32)
        xt, sab = out_buffer(2 * nsl, np.int32), out_buffer(2 * nsl, np.int32)
        check(load().desc_debug_shard_layout(self.handle, ptr(xpos, I32P), ptr(spos, I32P), ptr(xt, I32P), ptr(sab, I32P)))
        return dict(xpos=xpos[:2 * self.m], spos=spos[:2 * self.m], xt=xt[:2 * nsl].reshape(-1, 2), slot_ab=sab[:2 * nsl].reshape(-1, 2))

    def s0(self):
        out = out_buffer(self.m_cycle)
        check(load().desc_pgd_get_s0(self.handle, ptr(out, F64P)))
        return out[:self.m_cycle]

    def _result(self, iters, want_w=False, adam=None):
        bufs = dict(s_vec=out_buffer(self.m), obj=out_buffer(iters), avg=out_buffer(iters))
        r = Result()
        r.s_vec = ptr(bufs["s_vec"], F64P)
        r.obj_trace = ptr(bufs["obj"], F64P)
        r.avg_change_trace = ptr(bufs["avg"], F64P)
        if want_w:
            bufs["w"] = out_buffer(self.m_cycle)
            r.w = ptr(bufs["w"], F64P)
        if adam is not None:
            # the library reads AND writes m_cycle doubles through these pointers (HybridGradient.m_t / v_t)
            for a in adam:
                if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable
                        and (a.size == self.m_cycle or self.m_cycle == 0)):
                    raise ValueError(f"Adam state must be two writable contiguous float64 arrays of m_cycle = {self.m_cycle} entries")
            bufs["adam_m"], bufs["adam_v"] = adam
            r.adam_m = ptr(bufs["adam_m"], F64P)
            r.adam_v = ptr(bufs["adam_v"], F64P)
        return r, bufs

    def _pack(self, r, bufs):
        it = r.iters_run
        out = dict(S_vec=bufs["s_vec"][:self.m], obj=bufs["obj"][:it], avg=bufs["avg"][:it], iters_run=it,
                   t_end=r.t_end, ms_upload=r.ms_upload, ms_cycle_d=r.ms_cycle_d, ms_pgd=r.ms_pgd,
                   ms_total=r.ms_total, ms_structure=r.ms_structure)
        if "w" in bufs:
            out["w"] = bufs["w"][:self.m_cycle]
        if "adam_m" in bufs:
            out["adam_m"], out["adam_v"] = bufs["adam_m"], bufs["adam_v"]
        return out

    def run(self, params: Params, want_w=False, adam=None):
        r, bufs = self._result(params.iters, want_w, adam)
        check(load().desc_pgd_run(self.handle, C.byref(params), C.byref(r)))
        return self._pack(r, bufs)

    def run_traced(self, params: Params, dprob, err_vec, gcw_tol=1e-13, gcw_max_iters=500, adam=None):
        """desc_pgd_run_traced (params.make_plots = true, DESC_PGD.m:235-239): the run plus svec_errors and the GCW estimate
        of every iteration, R_est_all (iters_run, 3, 3, n)."""
        r, bufs = self._result(params.iters, adam=adam)
        n = dprob.n
        ev = np.ascontiguousarray(err_vec, dtype=np.float64).reshape(-1)
        if ev.size != self.m:
            raise ValueError("err_vec must have m entries")
        se = out_buffer(params.iters); Rall = out_buffer(max(params.iters, 1) * 9 * max(n, 1))
        check(load().desc_pgd_run_traced(self.handle, dprob.handle, C.byref(params), ptr(ev, F64P), gcw_tol, gcw_max_iters,
                                         ptr(se, F64P), ptr(Rall, F64P), C.byref(r)))
        out = self._pack(r, bufs)
        k = out["iters_run"]
        out["svec_errors"] = se[:k]
        out["R_est_all"] = Rall[:k * 9 * n].reshape(k, 9 * n).reshape((k, n, 3, 3)).transpose(0, 3, 2, 1)     # (t, r, c, node) from 3 x 3 x n column-major
        return out

    def reset(self, params: Params):
        self._iters_cap = params.iters
        check(load().desc_pgd_reset(self.handle, C.byref(params)))

    def iterate(self, n):
        check(load().desc_pgd_iterate(self.handle, n))

    def iterate_timed(self, n, per_kernel=False):
        ms, mk = C.c_float(), C.c_float()
        check(load().desc_pgd_iterate_timed(self.handle, n, C.byref(ms), C.byref(mk) if per_kernel else None))
        return ms.value, (mk.value if per_kernel else None)

    def sync(self):
        check(load().desc_pgd_sync(self.handle))

    def download(self, want_w=False):
        r, bufs = self._result(getattr(self, "_iters_cap", 1), want_w)
        check(load().desc_pgd_download(self.handle, C.byref(r)))
        return self._pack(r, bufs)

    # ---- multi-GPU pieces (see desc_amd/sharded.py)
    def shard_info(self):
        info = ShardInfo()
        check(load().desc_pgd_shard_info(self.handle, C.byref(info)))
        return info

    def shard_bind(self, t_send_ptr, t_recv_ptr, sall_ptr, stream_ptr=None):
        check(load().desc_pgd_shard_bind(self.handle, t_send_ptr, t_recv_ptr, sall_ptr, stream_ptr))

    def shard_colsum(self):
        check(load().desc_pgd_shard_colsum(self.handle))

    def shard_sweep(self):
        check(load().desc_pgd_shard_sweep(self.handle))

    def shard_finish(self, initial=0):
        check(load().desc_pgd_shard_finish(self.handle, initial))

    def shard_objective(self, phase):
        check(load().desc_pgd_shard_objective(self.handle, phase))

    # fused protocol: whole iterations enqueued from C (collectives = function pointers, e.g. RCCL's)
    def shard_set_collectives(self, comm=None, reduce_scatter=None, all_gather=None):
        c = Collectives(comm, C.cast(reduce_scatter, C.c_void_p) if reduce_scatter is not None else None,
                        C.cast(all_gather, C.c_void_p) if all_gather is not None else None)
        self._coll_keepalive = (reduce_scatter, all_gather)
        check(load().desc_pgd_shard_set_collectives(self.handle, C.byref(c)))

    def shard_start(self, params: Params):
        self._iters_cap = params.iters
        check(load().desc_pgd_shard_start(self.handle, C.byref(params)))

    def shard_iterate(self, n):
        check(load().desc_pgd_shard_iterate(self.handle, n))

    def shard_run(self, params: Params):
        r, bufs = self._result(params.iters)
        check(load().desc_pgd_shard_run(self.handle, C.byref(params), C.byref(r)))
        return self._pack(r, bufs)

    def stopped(self):
        f = C.c_int32()
        check(load().desc_pgd_stopped(self.handle, C.byref(f)))
        return bool(f.value)

    def destroy(self):
        if self.handle:
            load().desc_pgd_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


DTYPE_CODES = {np.dtype(np.float64): 0, np.dtype(np.int64): 1, np.dtype(np.int32): 2}      # DESC_DTYPE_*


def marshal_edges_native(Ind):
    """desc_marshal_edges: m x 2 array of 1-based node ids (float64 / int64 / int32, any strides) -> (n, ind_i, ind_j, sorted)."""
    if Ind.dtype not in DTYPE_CODES:
        Ind = Ind.astype(np.int64 if np.issubdtype(Ind.dtype, np.integer) else np.float64)
    m = Ind.shape[0]
    ii = np.empty(m, dtype=np.int32)
    jj = np.empty(m, dtype=np.int32)
    n, srt = C.c_int64(0), C.c_int32(0)
    item = Ind.dtype.itemsize
    check(load().desc_marshal_edges(C.c_void_p(Ind.ctypes.data), DTYPE_CODES[Ind.dtype], m, Ind.strides[0] // item, Ind.strides[1] // item,
                                    ptr(ii, I32P), ptr(jj, I32P), C.byref(n), C.byref(srt)))
    return int(n.value), ii, jj, bool(srt.value)


def marshal_rij_native(R, perm=None):
    """desc_marshal_rij: (3, 3, m) float64 array with any strides (+ edge permutation) -> the ABI's (m * 9,) buffer."""
    m = R.shape[2]
    out = np.empty(9 * m, dtype=np.float64)
    pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int64)
    check(load().desc_marshal_rij(ptr(R, F64P), m, R.strides[0] // 8, R.strides[1] // 8, R.strides[2] // 8, ptr(pp, I64P), ptr(out, F64P)))
    return out


def solve(prob: ProblemArrays, params: Params, want_w=False):
    """One-shot desc_pgd_solve: structure build + layout + run + download in one C call."""
    r = Result()
    bufs = dict(s_vec=out_buffer(prob.m), obj=out_buffer(params.iters), avg=out_buffer(params.iters))
    r.s_vec = ptr(bufs["s_vec"], F64P)
    r.obj_trace = ptr(bufs["obj"], F64P)
    r.avg_change_trace = ptr(bufs["avg"], F64P)
    check(load().desc_pgd_solve(C.byref(prob.c), C.byref(params), C.byref(r)))
    it = r.iters_run
    return dict(S_vec=bufs["s_vec"][:prob.m], obj=bufs["obj"][:it], avg=bufs["avg"][:it], iters_run=it, t_end=r.t_end,
                ms_upload=r.ms_upload, ms_cycle_d=r.ms_cycle_d, ms_pgd=r.ms_pgd, ms_total=r.ms_total, ms_structure=r.ms_structure)


def spectral_run(prob, weights=None, normalize_rows=False, tol=1e-13, max_iters=500, device=0):
    """desc_spectral_run[_dev] -> (R (3,3,n) Fortran-ordered, info dict).  prob: ProblemArrays or DeviceProblem."""
    n = prob.n
    R = out_buffer(9 * n)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    if w is not None and w.size != prob.m:
        raise ValueError("weights must have m entries")
    info = SpectralInfo()
    if isinstance(prob, DeviceProblem):
        check(load().desc_spectral_run_dev(prob.handle, ptr(w, F64P), 1 if normalize_rows else 0, tol, max_iters, ptr(R, F64P), C.byref(info)))
    else:
        check(load().desc_spectral_run(C.byref(prob.c), ptr(w, F64P), 1 if normalize_rows else 0, tol, max_iters, device,
                                       ptr(R, F64P), C.byref(info)))
    return R[:9 * n].reshape((3, 3, n), order="F"), dict(iters=info.iters, products=info.products, converged=bool(info.converged), residual=info.residual,
                                                        eigenvalues=list(info.eigenvalues), ms_total=info.ms_total)


def gcw_run(dprob: DeviceProblem, s_vec, tol=1e-13, max_iters=500):
    """desc_gcw_run_dev: GCW with the weights formed on the device from S_vec -> (R (3,3,n), info)."""
    n = dprob.n
    R = out_buffer(9 * n)
    S = np.ascontiguousarray(s_vec, dtype=np.float64)
    if S.size != dprob.m:
        raise ValueError("s_vec must have m entries")
    info = SpectralInfo()
    check(load().desc_gcw_run_dev(dprob.handle, ptr(S, F64P), tol, max_iters, ptr(R, F64P), C.byref(info)))
    return R[:9 * n].reshape((3, 3, n), order="F"), dict(iters=info.iters, products=info.products, converged=bool(info.converged), residual=info.residual,
                                                        eigenvalues=list(info.eigenvalues), ms_total=info.ms_total)


def cemp_run(prob, beta, max_iter, nsample, seed=0, device=0):
    b = np.ascontiguousarray(beta, dtype=np.float64).reshape(-1)
    S = out_buffer(prob.m)
    ms = C.c_double()
    L = load()
    if isinstance(prob, DeviceProblem):
        check(L.desc_cemp_run_dev(prob.handle, ptr(b, F64P), b.shape[0], int(max_iter), int(nsample), int(seed), ptr(S, F64P), C.byref(ms)))
    else:
        check(L.desc_cemp_run(C.byref(prob.c), ptr(b, F64P), b.shape[0], int(max_iter), int(nsample), int(seed), device, ptr(S, F64P),
                              C.byref(ms)))
    return S[:prob.m], ms.value


def refine_run(prob, s_vec, R_init, stop_threshold=1e-3, max_iters=100, device=0, verbose=False):
    """desc_refine_run[_dev] -> (R (3,3,n), info)."""
    n = prob.n
    S = np.ascontiguousarray(s_vec, dtype=np.float64)
    Ri = np.ascontiguousarray(np.asarray(R_init, dtype=np.float64).reshape(-1, order="F"))
    Ro = out_buffer(9 * n)
    if S.size != prob.m or Ri.size != 9 * n:
        raise ValueError("s_vec must have m entries and R_init 3 x 3 x n")
    info = RefineInfo(); info.verbose = 1 if verbose else 0
    L = load()
    if isinstance(prob, DeviceProblem):
        check(L.desc_refine_run_dev(prob.handle, ptr(S, F64P), ptr(Ri, F64P), stop_threshold, max_iters, ptr(Ro, F64P), C.byref(info)))
    else:
        check(L.desc_refine_run(C.byref(prob.c), ptr(S, F64P), ptr(Ri, F64P), stop_threshold, max_iters, device, ptr(Ro, F64P), C.byref(info)))
    return Ro[:9 * n].reshape((3, 3, n), order="F"), dict(iters=info.iters, cg_iters=info.cg_iters, score=info.score, ms_total=info.ms_total,
                                                        cg_unconverged=info.cg_unconverged, cg_residual=info.cg_residual)


def spmm_variants(dprob: DeviceProblem, reps=20):
    """desc_debug_spmm_variants: ms per block-SpMM product, vector-FMA vs v_mfma_f64_4x4x4 form (measurement hook)."""
    out = out_buffer(4)
    check(load().desc_debug_spmm_variants(dprob.handle, int(reps), ptr(out, F64P)))
    return dict(ms_valu=float(out[0]), ms_mfma=float(out[1]), max_abs_diff=float(out[2]), mfma_layout=int(out[3]))


def trim_memory():
    """Give the device blocks the library has parked for reuse back to the driver; returns the bytes released."""
    return int(load().desc_trim_memory())


def host_exports():
    """How often this process exported a device-built structure to host memory (diagnostics)."""
    return int(load().desc_structure_host_exports())


def device_count():
    rc = load().desc_device_count()
    if rc < 0:
        raise DescError(load().desc_last_error().decode())
    return rc
