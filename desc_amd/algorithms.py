"""Host-side mirror of the reference's solver entry points.

The reference is MATLAB and no MATLAB/Octave/MEX toolchain exists in the build
environment, so the host layer above the C ABI is written in Python with the
reference's names, argument meaning and conventions:

    S_vec = DESC_PGD(Ind, RijMat, params)        Algorithms/DESC_PGD.m:14

``Ind`` is ``m x 2`` with 1-based node ids ``i<j``; ``RijMat`` is ``3 x 3 x m``;
``params`` is a struct-like object (attributes or dict keys) with the fields the
reference reads: ``iters`` (:170), ``Gradient`` (:207), ``make_plots`` (:235) and,
when plotting, ``ErrVec`` / ``R_orig`` (:236-238).  ``learning_rate`` is accepted
and ignored, as in the reference (:169 is commented out).  Optional fields that do
not exist in the reference: ``seed`` (cycle-sampling seed; MATLAB uses its global
RNG), ``device``, ``verbose``, ``build_where``.

All numerical work of the hot path happens in libdesc_amd.so on the GPU; this file
only marshals arguments.  MATLAB wrappers with the same signatures and the MEX shim
over the same ABI are in matlab/.
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _lib
from .stepsize import ConstantStepSize, HybridGradient, PiecewiseStepSize


def _get(params, name, default=None):
    if isinstance(params, dict):
        return params.get(name, default)
    return getattr(params, name, default)


def _set(params, name, value):
    if isinstance(params, dict):
        params[name] = value
    else:
        try:
            setattr(params, name, value)
        except Exception:
            pass


def marshal_edges(Ind, RijMat=None):
    """MATLAB arrays -> C-ABI arrays.

    Returns (n, ind_i, ind_j, rij, perm): 0-based int32 endpoints sorted by (i,j),
    rij as an (m*9,) float64 buffer in the reference's memory order
    (element (r,c,l) at 9*l + r + 3*c), and ``perm`` such that row ``perm[t]`` of the
    caller's ``Ind`` is the t-th sorted edge (``None`` when already sorted).  The
    reference silently requires sorted input (DESC_PGD.m:5,31-34); unsorted input is
    sorted here and outputs are returned in the caller's order."""
    Ind = np.asarray(Ind)
    if Ind.ndim != 2 or Ind.shape[1] != 2:
        raise ValueError("Ind must be m x 2")
    m = Ind.shape[0]
    if Ind.dtype.kind not in "iuf":
        raise ValueError("Ind must hold integer node ids")
    try:        # one threaded pass of the library (desc_marshal_edges): checks, 0-based int32 endpoints, n = max(Ind(:)) (DESC_PGD.m:21), order
        n, ii, jj, is_sorted = _lib.marshal_edges_native(Ind)
    except _lib.DescError as e:
        if e.code != _lib.ERR_INVALID:
            raise
        raise ValueError(str(e).split(": ", 1)[-1]) from None
    perm = None
    if not is_sorted:
        perm = np.lexsort((jj, ii))
        ii, jj = ii[perm], jj[perm]
        if m > 1 and np.any((ii[1:] == ii[:-1]) & (jj[1:] == jj[:-1])):
            raise ValueError("Ind lists an edge twice")
    rij = None
    if RijMat is not None:
        R = np.asarray(RijMat, dtype=np.float64)
        if R.shape != (3, 3, m):
            raise ValueError("RijMat must be 3 x 3 x m")
        if perm is None and R.flags.f_contiguous:
            rij = R.reshape(-1, order="F")            # MATLAB memory order of a 3x3xm array, r + 3c + 9l: the ABI's own, no copy
        else:
            if any(st % 8 for st in R.strides):
                R = np.ascontiguousarray(R)
            rij = _lib.marshal_rij_native(R, perm)    # any other strides (NumPy's C order) and / or the edge permutation: one threaded pass
    return n, ii, jj, rij, perm


def gradient_to_params(G, p: _lib.Params):
    """Translate a params.Gradient plugin object into the flat C struct."""
    if isinstance(G, ConstantStepSize):
        p.step_kind = _lib.STEP_CONSTANT
        p.lr = G.learning_rate
        p.t0 = 0
    elif isinstance(G, PiecewiseStepSize):
        p.step_kind = _lib.STEP_PIECEWISE
        p.lr = G.learning_rate
        p.decay_interval = float(G.decay_interval)
        p.t0 = int(G.t)
    elif isinstance(G, HybridGradient):
        p.step_kind = _lib.STEP_HYBRID
        p.lr = G.lr
        p.beta1, p.beta2 = G.beta_1, G.beta_2
        p.decay_interval = float(G.decay_interval)
        p.hybrid_strategy = int(G.strategy)
        p.t0 = int(G.t)
    else:
        raise TypeError("params.Gradient must be a ConstantStepSize, PiecewiseStepSize or HybridGradient "
                        "object (Utils/*.m); arbitrary GetStep callbacks cannot run inside the HIP sweep")


def make_c_params(params):
    p = _lib.default_params()
    iters = _get(params, "iters")
    if iters is None:
        raise ValueError("params.iters is required (DESC_PGD.m:170)")
    p.iters = int(iters)
    G = _get(params, "Gradient")
    if G is None:
        raise ValueError("params.Gradient is required (DESC_PGD.m:207)")
    gradient_to_params(G, p)
    p.seed = int(_get(params, "seed", 0))
    p.device = int(_get(params, "device", 0))
    p.verbose = 1 if _get(params, "verbose", True) else 0
    p.build_where = int(_get(params, "build_where", _lib.BUILD_DEVICE))
    return p, G


def DESC_PGD(Ind, RijMat, params, return_info=False, _marshalled=None):
    """[S_vec] = DESC_PGD(Ind, RijMat, params) -- Algorithms/DESC_PGD.m:14.

    Returns the estimated corruption level of every edge (length-m vector in the
    caller's edge order).  With ``return_info`` also a dict with the objective and
    average-change traces, iteration count, timings and structure sizes.
    ``_marshalled`` (internal, used by DESC()): (perm, ProblemArrays, DeviceProblem or a callable that returns it) already prepared."""
    p, G = make_c_params(params)
    make_plots = bool(_get(params, "make_plots", False))
    if make_plots and (_get(params, "ErrVec") is None or _get(params, "R_orig") is None):
        raise ValueError("params.make_plots=true reads params.ErrVec and params.R_orig (DESC_PGD.m:236-238)")
    if _marshalled is None:
        n, ii, jj, rij, perm = marshal_edges(Ind, RijMat)
        if ii.shape[0] == 0:
            raise ValueError("empty edge list")
        prob = _lib.ProblemArrays(n, ii, jj, rij)
        dprob = prob
    else:
        perm, prob, dprob = _marshalled
    verbose = bool(p.verbose)
    cb = None
    if verbose:        # the reference's per-iteration line (DESC_PGD.m:241), streamed by the library while the loop runs
        def _line(user, it, avg, obj):
            print("iter %d: average change in S_vec %f, objective value: %f" % (it, avg, obj), flush=True)
        cb = _lib.PROGRESS_FN(_line)
        p.progress = C.cast(cb, C.c_void_p)
        p.verbose = 0
    hybrid_state = isinstance(G, HybridGradient) and G.strategy == 0       # carries m_cycle-long moment vectors in and out
    if _marshalled is None and not make_plots and not return_info and not hybrid_state:
        # the reference's own signature, S_vec = DESC_PGD(Ind, RijMat, params): ONE C call (desc_pgd_solve), in which the rotations go up
        # while the structure is built (C4: 13 ms hidden)
        if verbose:
            print("compute R cycle")                      # DESC_PGD.m:132
            print("S0Mat")                                # :145
            print("Initialization completed!")            # :160
            print("Reweighting Procedure Started ...")    # :162
        out = _lib.solve(prob, p)
        if isinstance(G, (PiecewiseStepSize, HybridGradient)):
            G.t = int(out["t_end"])
        if perm is None:
            return out["S_vec"]
        S_vec = np.empty_like(out["S_vec"])
        S_vec[perm] = out["S_vec"]
        return S_vec
    try:
        st = _lib.Structure.build(prob, p.n_sample_min, p.seed, p.build_where, p.device)
    except _lib.DescError as e:
        # the device builder refuses graphs whose bitmaps / per-edge staging exceed its budget
        if p.build_where != _lib.BUILD_DEVICE or e.code != _lib.ERR_TOO_LARGE:
            raise
        st = _lib.Structure.build(prob, p.n_sample_min, p.seed, _lib.BUILD_HOST, p.device)
    try:
        sizes = st.sizes()                    # O(1): the structure stays on the device
        ms_structure = sizes.pop("ms_build")
        if callable(dprob):
            dprob = dprob()                       # DESC(): the device problem has been going up on a helper thread meanwhile
        solver = _lib.Solver(dprob, st, p.device)
    finally:
        st.free()
    try:
        if verbose:
            print("compute R cycle")                      # DESC_PGD.m:132
            print("S0Mat")                                # :145
            print("Initialization completed!")            # :160
            print("Reweighting Procedure Started ...")    # :162
        adam = None
        if isinstance(G, HybridGradient) and G.strategy == 0:
            mt = G.m_t if (G.t > 0 and G.m_t is not None) else np.zeros(solver.m_cycle)
            vt = G.v_t if (G.t > 0 and G.v_t is not None) else np.zeros(solver.m_cycle)
            if mt.shape[0] != solver.m_cycle:
                raise ValueError("HybridGradient state has a different length than this problem's cycle vector")
            adam = (np.ascontiguousarray(mt, dtype=np.float64).copy(), np.ascontiguousarray(vt, dtype=np.float64).copy())
        if not make_plots:
            out = solver.run(p, adam=adam)
        else:
            out = _run_with_plots(solver, p, params, prob, dprob, perm, verbose, adam)
    finally:
        solver.destroy()
    # plugin state after the run (handle-object semantics)
    if isinstance(G, (PiecewiseStepSize, HybridGradient)):
        G.t = int(out["t_end"])
    if adam is not None:
        G.m_t, G.v_t = out["adam_m"], out["adam_v"]
    S_sorted = out["S_vec"]
    if perm is not None:
        S_vec = np.empty_like(S_sorted)
        S_vec[perm] = S_sorted
    else:
        S_vec = S_sorted
    if return_info:
        info = dict(out)
        info.update(sizes)
        info["ms_structure"] = ms_structure
        return S_vec, info
    return S_vec


def _run_with_plots(solver, p, params, prob, dprob, perm, verbose, adam=None):
    """params.make_plots = true (DESC_PGD.m:235-239): after every iteration the error of S_vec against params.ErrVec and
    the rotation error of GCW(S_vec) against params.R_orig (GlobalSOdCorrectRight = the alignment of Rotation_Alignment).
    A composition of device rows: one sweep, one S_vec download and one GCW eigen-solve per iteration."""
    own = None
    if not isinstance(dprob, _lib.DeviceProblem):
        own = dprob = _lib.DeviceProblem(prob, p.device)
    ErrVec = np.asarray(_get(params, "ErrVec"), dtype=np.float64).reshape(-1)
    R_orig = np.asarray(_get(params, "R_orig"), dtype=np.float64)
    if perm is not None:
        ErrVec = ErrVec[perm]
    try:
        out = solver.run_traced(p, dprob, ErrVec, adam=adam)                     # desc_pgd_run_traced: :236-237 for every iteration
    finally:
        if own is not None:
            own.free()
    mse_means, mse_medians = [], []
    for R_est in out.pop("R_est_all"):
        _, _, mean_e, med_e = Rotation_Alignment(R_est, R_orig)                                  # :238
        mse_means.append(mean_e); mse_medians.append(med_e)
    svec_errors = out["svec_errors"]
    k = out["iters_run"]
    out["svec_errors"] = np.array(svec_errors[:k]); out["MSE_means"] = np.array(mse_means[:k]); out["MSE_medians"] = np.array(mse_medians[:k])
    return out


def plot_convergence(info, path=None):
    """The 2 x 2 figure of DESC.m:315-344 from the traces of a make_plots run (matplotlib, if it is installed)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return None
    fig, ax = plt.subplots(2, 2, figsize=(10, 8))
    for a, key, title, yl in ((ax[0, 0], "svec_errors", "Convergence of Corruption Estimate Vector (S_vec, sampled)", "Average distance to true corruption"),
                              (ax[0, 1], "obj", "Convergence of Objective Function (sampled)", "Value of Objective Function"),
                              (ax[1, 0], "MSE_means", "Convergence of Rotation Estimate, Mean (sampled)", "Mean Error in R estimate (degrees)"),
                              (ax[1, 1], "MSE_medians", "Convergence of Rotation Estimate, Median (sampled)", "Median Error in R estimate (degrees)")):
        a.plot(np.arange(1, len(info[key]) + 1), info[key]); a.set_title(title, fontsize=9); a.set_xlabel("Iteration number"); a.set_ylabel(yl)
    fig.tight_layout()
    if path:
        fig.savefig(path)
    return fig


def Spectral(Ind, RijMat, device=0, return_info=False):
    """R_est = Spectral(Ind, RijMat) -- Algorithms/Spectral.m:15.  3 x 3 x n rotations, defined
    up to one global right rotation (use Rotation_Alignment to compare)."""
    n, ii, jj, rij, perm = marshal_edges(Ind, RijMat)
    prob = _lib.ProblemArrays(n, ii, jj, rij)
    R, info = _lib.spectral_run(prob, None, False, device=device)
    return (R, info) if return_info else R


def GCW(Ind, AdjMat, RijMat, S_vec, device=0, return_info=False):
    """R_est = GCW(Ind, AdjMat, RijMat, S_vec) -- Utils/GCW.m:1.  ``AdjMat`` is accepted for
    signature compatibility and not used (it is implied by ``Ind``)."""
    n, ii, jj, rij, perm = marshal_edges(Ind, RijMat)
    S = np.asarray(S_vec, dtype=np.float64).reshape(-1)
    if S.shape[0] != ii.shape[0]:
        raise ValueError("S_vec must have one entry per edge")
    if perm is not None:
        S = S[perm]
    w = 1.0 / (S * np.sqrt(S) + 1e-8)                         # GCW.m:20: SVec.^(1.5) (sqrt form: 5x cheaper than pow; S >= 0)
    prob = _lib.ProblemArrays(n, ii, jj, rij)
    R, info = _lib.spectral_run(prob, w, True, device=device)
    return (R, info) if return_info else R


def CEMP(Ind, RijMat, CEMP_parameters, return_info=False):
    """SVec = CEMP(Ind, RijMat, CEMP_parameters) -- Algorithms/CEMP.m:24.  Fields read:
    ``max_iter``, ``reweighting``, ``nsample`` (``gcw_beta`` is ignored, as in the reference);
    optional ``seed`` / ``device`` (not in the reference)."""
    n, ii, jj, rij, perm = marshal_edges(Ind, RijMat)
    prob = _lib.ProblemArrays(n, ii, jj, rij)
    beta = np.atleast_1d(np.asarray(_get(CEMP_parameters, "reweighting"), dtype=np.float64))
    if _get(CEMP_parameters, "verbose", False):
        for line in ("sampling 3-cycles", "Sampling Finished!", "Initializing", "Initialization completed!",
                     "Reweighting Procedure Started ..."):           # CEMP.m:45,67,68,104,105
            print(line)
    S, ms = _lib.cemp_run(prob, beta, int(_get(CEMP_parameters, "max_iter")), int(_get(CEMP_parameters, "nsample")),
                          int(_get(CEMP_parameters, "seed", 0)), int(_get(CEMP_parameters, "device", 0)))
    if perm is not None:
        out = np.empty_like(S); out[perm] = S; S = out
    return (S, dict(ms_total=ms)) if return_info else S


def DESC(Ind, RijMat, params, return_info=False):
    """[R_est, R_init, S_vec] = DESC(Ind, RijMat, params) -- Algorithms/DESC.m:14 (the call of
    Demo/compare_algorithms.m:72): DESC_PGD (:16-261) -> GCW initialisation (:263) -> reweighted
    Lie-algebraic refinement (:265-313).  All three stages run on the GPU."""
    n, ii, jj, rij, perm = marshal_edges(Ind, RijMat)
    if ii.shape[0] == 0:
        raise ValueError("empty edge list")
    prob = _lib.ProblemArrays(n, ii, jj, rij)
    device = int(_get(params, "device", 0))
    # Ind / RijMat / CSR index go to HBM once for all three stages -- on a helper thread, while DESC_PGD builds the cycle structure (which needs
    # only the edge list): DESC_PGD asks for the device problem when it creates the solver
    import threading
    box = {}

    def _upload():
        try:
            box["dp"] = _lib.DeviceProblem(prob, device)
        except BaseException as e:      # noqa: BLE001  (re-raised on the calling thread)
            box["err"] = e

    th = threading.Thread(target=_upload)
    th.start()
    if os.environ.get("DESC_DEBUG_SERIAL_UPLOAD") == "1":       # A/B: the upload first, then the structure (as before round 4)
        th.join()

    def _dprob():
        th.join()
        if "err" in box:
            raise box["err"]
        return box["dp"]

    dprob = None
    try:
        S_vec, info = DESC_PGD(Ind, RijMat, params, return_info=True, _marshalled=(perm, prob, _dprob))
        dprob = _dprob()
        S_sorted = S_vec if perm is None else S_vec[perm]
        R_init, ginfo = _lib.gcw_run(dprob, S_sorted)                    # GCW.m:9-36, weights (GCW.m:20) formed on the device
        verbose = bool(_get(params, "verbose", True))
        if verbose:
            print("Rotation Initialized!"); print("Start DESC refinement ...")                # DESC.m:283-284
        R_est, rinfo = _lib.refine_run(dprob, S_sorted, R_init, verbose=verbose)
        if verbose:
            print("DONE!")                                                                    # DESC.m:313
    finally:
        th.join()
        if "dp" in box:
            box["dp"].free()
    if return_info:
        return R_est, R_init, S_vec, dict(pgd=info, gcw=ginfo, refine=rinfo)
    return R_est, R_init, S_vec


def Rotation_Alignment(R_est, R_gt):
    """[R_out, R_align, mean_error, median_error] = Rotation_Alignment(R_est, R_gt)
    -- Utils/Rotation_Alignment.m:13-38 (evaluation helper: host NumPy, O(n))."""
    R_est = np.asarray(R_est, dtype=np.float64); R_gt = np.asarray(R_gt, dtype=np.float64)
    d, n = R_gt.shape[0], R_gt.shape[2]
    A = np.einsum("abk,ack->bc", R_est, R_gt)                 # sum_k R_est_k' R_gt_k
    U1, _, V1t = np.linalg.svd(A)
    D = np.eye(d); D[-1, -1] = np.linalg.det(U1 @ V1t)
    R_align = U1 @ D @ V1t
    R_out = np.einsum("abk,bc->ack", R_est, R_align)
    tr = np.einsum("abk,abk->k", R_gt, R_out)
    x = (tr - 1.0) / 2.0
    err = np.where(np.abs(x) <= 1, np.arccos(np.clip(x, -1, 1)), np.abs(np.arccos(x.astype(complex)))) / np.pi * 180
    return R_out, R_align, float(np.mean(err)), float(np.median(err))
