"""ORACLE -- test infrastructure only (ctypes binding of oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package ``desc_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liboracle.so with gcc (plain C + OpenMP)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "desc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


class _Sizes(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("m_pos", C.c_int64),
                ("m_cycle", C.c_int64), ("n_sample", C.c_int32)]


class _Params(C.Structure):
    _fields_ = [("iters", C.c_int32), ("step_kind", C.c_int32),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("decay_interval", C.c_double),
                ("hybrid_strategy", C.c_int32), ("t0", C.c_int32),
                ("patience", C.c_int32), ("stop_tol", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        # a GPU box exposes every host thread but grants ~16 CPUs: an OpenMP team of
        # 256 spinning threads is pathologically slow there
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _LIB = C.CDLL(so)
        _LIB.oracle_set_threads(int(os.environ.get("ORACLE_THREADS", granted_cpus()["granted"])))
        _LIB.oracle_sample_key.restype = C.c_uint64
        _LIB.oracle_sample_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _LIB.oracle_num_threads.restype = C.c_int
    return _LIB


def granted_cpus():
    """CPUs this process may actually use: the affinity mask, cut down by the cgroup's CPU quota where one is set (a GPU box shows all
    256 host threads in the mask and grants a share of them through the quota).  The OpenMP team of the oracle is sized to `granted`."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:                                            # cgroup v2: "max 100000" or "<quota> <period>"
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                        # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    granted = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return dict(granted=granted, affinity=affinity, cgroup_quota=quota, host_cpus=os.cpu_count())


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


_M64 = (1 << 64) - 1


def _mix64(x):
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & _M64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & _M64
    x ^= x >> 31
    return x


def sample_key(seed, edge, k):
    """Pure-Python twin of oracle_sample_key (0-based edge id and node id)."""
    a = _mix64((seed ^ (((edge + 1) * 0x9E3779B97F4A7C15) & _M64)) & _M64)
    return _mix64(a ^ (((k + 1) * 0xD1B54A32D192ED03) & _M64))


def keyed_sampler(seed):
    """Sampler for desc_pgd_literal: keeps the n_sample common neighbours with the
    smallest sample_key, returned ascending -- the same deterministic stand-in for
    ``datasample`` (DESC_PGD.m:84) that oracle_build_structure and the product use."""

    def _s(l, IJ, CoInd_ij, n_sample):
        keys = [(sample_key(seed, int(IJ) - 1, int(k) - 1), int(k)) for k in CoInd_ij]
        keys.sort()
        return np.array(sorted(k for _, k in keys[:n_sample]), dtype=np.int64)

    return _s


def build_structure(n, ii, jj, seed=0, n_sample_min=30):
    """DESC_PGD.m:19-54,79-127 sparse.  ii, jj: 0-based int32 endpoints (i<j, sorted)."""
    L = lib()
    ii = np.ascontiguousarray(ii, dtype=np.int32)
    jj = np.ascontiguousarray(jj, dtype=np.int32)
    m = ii.shape[0]
    sz = _Sizes()
    codeg = np.zeros(max(m, 1), dtype=np.int32)
    I32, I64 = C.c_int32, C.c_int64
    args0 = [C.c_int64(n), C.c_int64(m), _p(ii, I32), _p(jj, I32), C.c_int32(n_sample_min),
             C.c_uint64(seed), C.byref(sz)]
    L.oracle_build_structure(*args0, _p(codeg, I32), None, None, None, None, None, None, None)
    mp, mc = sz.m_pos, sz.m_cycle
    pos_edge = np.zeros(max(mp, 1), dtype=np.int32)
    cum_ind = np.zeros(mp + 1, dtype=np.int64)
    kk, e_jk, e_ki, ikj, jki = (np.zeros(max(mc, 1), dtype=np.int32) for _ in range(5))
    L.oracle_build_structure(*args0, _p(codeg, I32), _p(pos_edge, I32), _p(cum_ind, I64),
                             _p(kk, I32), _p(e_jk, I32), _p(e_ki, I32), _p(ikj, I32), _p(jki, I32))
    return dict(n=n, m=m, m_pos=mp, m_cycle=mc, n_sample=sz.n_sample, codeg=codeg[:m],
                pos_edge=pos_edge[:mp], cum_ind=cum_ind, k=kk[:mc], e_jk=e_jk[:mc],
                e_ki=e_ki[:mc], ikj=ikj[:mc], jki=jki[:mc])


def cycle_d(ii, jj, rij, st):
    """DESC_PGD.m:129-147.  rij: (m,9) float64, MATLAB block order."""
    L = lib()
    ii = np.ascontiguousarray(ii, dtype=np.int32)
    jj = np.ascontiguousarray(jj, dtype=np.int32)
    rij = np.ascontiguousarray(rij, dtype=np.float64)
    S0 = np.zeros(max(st["m_cycle"], 1), dtype=np.float64)
    I32, I64, F64 = C.c_int32, C.c_int64, C.c_double
    L.oracle_cycle_d(C.c_int64(st["m_pos"]), _p(ii, I32), _p(jj, I32), _p(rij, F64),
                     _p(np.ascontiguousarray(st["pos_edge"]), I32), _p(st["cum_ind"], I64),
                     _p(np.ascontiguousarray(st["k"]), I32), _p(np.ascontiguousarray(st["e_jk"]), I32),
                     _p(np.ascontiguousarray(st["e_ki"]), I32), _p(S0, F64))
    return S0[:st["m_cycle"]]


def pgd_run(st, S0, iters, step_kind=0, lr=0.01, beta1=0.9, beta2=0.999, decay_interval=25,
            hybrid_strategy=0, t0=0, patience=30, stop_tol=1e-5, adam_m=None, adam_v=None):
    """DESC_PGD.m:148-261 given the structure.  Returns dict(S_vec, w, obj, avg, iters_run)."""
    L = lib()
    m, mp, mc = st["m"], st["m_pos"], st["m_cycle"]
    S_vec = np.zeros(max(m, 1), dtype=np.float64)
    w = np.zeros(max(mc, 1), dtype=np.float64)
    obj = np.zeros(max(iters, 1), dtype=np.float64)
    avg = np.zeros(max(iters, 1), dtype=np.float64)
    p = _Params(iters, step_kind, lr, beta1, beta2, float(decay_interval), hybrid_strategy, t0,
                patience, stop_tol)
    I32, I64, F64 = C.c_int32, C.c_int64, C.c_double
    S0c = np.ascontiguousarray(S0, dtype=np.float64)
    if S0c.shape[0] == 0:
        S0c = np.zeros(1)
    arrs = [np.ascontiguousarray(st[k]) if st[k].shape[0] else np.zeros(1, dtype=np.int32)
            for k in ("pos_edge", "e_jk", "e_ki", "ikj", "jki")]
    L.oracle_pgd_run.restype = C.c_int
    it = L.oracle_pgd_run(C.c_int64(m), C.c_int64(mp), _p(arrs[0], I32), _p(st["cum_ind"], I64),
                          _p(arrs[1], I32), _p(arrs[2], I32), _p(arrs[3], I32), _p(arrs[4], I32),
                          _p(S0c, F64), C.byref(p), _p(S_vec, F64), _p(w, F64), _p(obj, F64),
                          _p(avg, F64), _p(adam_m, F64), _p(adam_v, F64))
    return dict(S_vec=S_vec[:m], w=w[:mc], obj=obj[:it], avg=avg[:it], iters_run=it)


def pgd_run_ld(st, S0, iters, step_kind=0, lr=0.01, decay_interval=25, t0=0, patience=30, stop_tol=1e-5):
    """oracle_pgd_run_ld: the same loop carried in long double from start to end (constant / piecewise step), a yardstick
    for round-off amplification.  Returns dict(S_vec, w, obj, iters_run), rounded to double once at the end."""
    L = lib()
    m, mp, mc = st["m"], st["m_pos"], st["m_cycle"]
    S_vec = np.zeros(max(m, 1)); w = np.zeros(max(mc, 1)); obj = np.zeros(max(iters, 1))
    p = _Params(iters, step_kind, lr, 0.9, 0.999, float(decay_interval), 0, t0, patience, stop_tol)
    I32, I64, F64 = C.c_int32, C.c_int64, C.c_double
    S0c = np.ascontiguousarray(S0, dtype=np.float64)
    if S0c.shape[0] == 0:
        S0c = np.zeros(1)
    arrs = [np.ascontiguousarray(st[k]) if st[k].shape[0] else np.zeros(1, dtype=np.int32)
            for k in ("pos_edge", "e_jk", "e_ki", "ikj", "jki")]
    L.oracle_pgd_run_ld.restype = C.c_int
    it = L.oracle_pgd_run_ld(C.c_int64(m), C.c_int64(mp), _p(arrs[0], I32), _p(st["cum_ind"], I64),
                             _p(arrs[1], I32), _p(arrs[2], I32), _p(arrs[3], I32), _p(arrs[4], I32),
                             _p(S0c, F64), C.byref(p), _p(S_vec, F64), _p(w, F64), _p(obj, F64))
    if it < 0:
        raise ValueError("oracle_pgd_run_ld: constant / piecewise step only")
    return dict(S_vec=S_vec[:m], w=w[:mc], obj=obj[:it], iters_run=it)


def num_threads():
    return lib().oracle_num_threads()
