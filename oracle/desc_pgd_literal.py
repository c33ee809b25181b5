"""ORACLE -- test infrastructure only.  Never imported by the product path.

Literal NumPy restatement of the reference's ``Algorithms/DESC_PGD.m`` (the same
text is inlined at ``Algorithms/DESC.m:16-261``).  It keeps the reference's
dense arrays (``AdjMat``, ``IndMat``, ``RijMat4d``, ``IJK_Mat``, ``IKJ_appears``)
and its variable names so that each block can be read side by side with the
``.m`` text; it is therefore only usable at small ``n``.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors and no
MATLAB/Octave interpreter exists in the build environment, so this restatement
cannot be checked against an execution of the reference.  It is pinned only by
(i) hand-derived known-answer cases (tests/test_oracle.py), (ii) invariants
and a finite-difference gradient check, and (iii) agreement with the independent
sparse C restatement in oracle/desc_oracle.c.

All indices inside this file are 1-based *values* stored in 0-based NumPy arrays,
exactly as MATLAB would hold them; array position p (0-based) corresponds to
MATLAB position p+1.  Line numbers cite /root/reference/Algorithms/DESC_PGD.m.
"""
from __future__ import annotations

import math
import numpy as np


# --------------------------------------------------------------------------
# MATLAB builtins restated
# --------------------------------------------------------------------------
def matlab_abs_acos(x):
    """abs(acos(x)) with MATLAB's complex extension outside [-1, 1].

    DESC_PGD.m:147 evaluates ``abs(acos((R_trace-1)./2))``.  For real x > 1 MATLAB
    returns acos(x) = 0 + 1i*acosh(x) (modulus acosh(x)); for x < -1 it returns
    pi - 1i*acosh(-x) (modulus hypot(pi, acosh(-x))).
    """
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    inside = np.abs(x) <= 1.0
    out[inside] = np.arccos(x[inside])
    hi = x > 1.0
    out[hi] = np.arccosh(x[hi])
    lo = x < -1.0
    out[lo] = np.hypot(np.pi, np.arccosh(-x[lo]))
    nan = np.isnan(x)
    out[nan] = np.nan
    return out


def matlab_median(v):
    """median() of a vector; median([]) is NaN (DESC_PGD.m:43)."""
    v = np.asarray(v, dtype=np.float64)
    if v.size == 0:
        return float("nan")
    return float(np.median(v))


def matlab_max_ignore_nan(a, b):
    """max(a, b) for scalars: MATLAB's max ignores NaN operands."""
    if math.isnan(a):
        return b
    if math.isnan(b):
        return a
    return max(a, b)


# --------------------------------------------------------------------------
# Step-size plugins (Utils/ConstantStepSize.m, PiecewiseStepSize.m,
# HybridGradient.m) restated as tiny stateful classes.
# --------------------------------------------------------------------------
class ConstantStepSize:
    """Utils/ConstantStepSize.m:9-11."""

    def __init__(self, learning_rate):
        self.learning_rate = float(learning_rate)

    def GetStep(self, grad):
        return -self.learning_rate * grad


class PiecewiseStepSize:
    """Utils/PiecewiseStepSize.m:8-18 (counter ``t`` persists across calls)."""

    def __init__(self, learning_rate, decay_interval):
        self.learning_rate = float(learning_rate)
        self.decay_interval = decay_interval
        self.t = 0

    def GetStep(self, grad):
        self.t += 1
        step_size = self.learning_rate / (math.trunc(self.t / self.decay_interval) + 1)
        return -step_size * grad


class HybridGradient:
    """Utils/HybridGradient.m:13-52 (Adam, switchable to a decayed plain step)."""

    def __init__(self, lr, beta_1, beta_2, decay_interval):
        self.lr = float(lr)
        self.beta_1 = float(beta_1)
        self.beta_2 = float(beta_2)
        self.decay_interval = decay_interval
        self.t = 0
        self.strategy = 0
        self.m_t = None
        self.v_t = None

    def GetStep(self, grad):
        if self.t == 0:                                    # :24-27
            self.m_t = np.zeros_like(grad)
            self.v_t = np.zeros_like(grad)
        step = None
        if self.strategy == 0:                             # :28-35
            self.t += 1
            self.m_t = (self.beta_1 * self.m_t) + (1 - self.beta_1) * grad
            self.v_t = (self.beta_2 * self.v_t) + (1 - self.beta_2) * (grad ** 2)
            corr_m_t = self.m_t / (1 - self.beta_1 ** self.t)
            corr_v_t = self.v_t / (1 - self.beta_2 ** self.t)
            step = -self.lr * corr_m_t / (np.sqrt(corr_v_t) + 10 ** (-8))
        if self.strategy == 1:                             # :36-41
            self.t += 1
            step_size = 100 * (self.lr / (math.trunc(self.t / self.decay_interval) + 1))
            step = -step_size * grad
        return step

    def stopAdam(self):
        self.strategy = 1
        return self


# --------------------------------------------------------------------------
# The literal restatement
# --------------------------------------------------------------------------
def project_simplex_literal(w_new):
    """DESC_PGD.m:215-224: sort, first i with sum(w(i:end)-w(i)) < 1, threshold T, clamp."""
    w_new = np.asarray(w_new, dtype=np.float64)
    nsample = w_new.shape[0]
    w = np.sort(w_new)                               # :215
    Ti = 0
    for i in range(1, nsample + 1):                  # :217-222
        if np.sum(w[i - 1:] - w[i - 1]) < 1:
            Ti = i
            break
    T = w[Ti - 1] - (1 - np.sum(w[Ti - 1:] - w[Ti - 1])) / len(w[Ti - 1:])   # :223
    return np.maximum(w_new - T, 0)                  # :224


def default_sampler(rng):
    """datasample(CoInd_ij, n_sample, 'Replace', false) with a NumPy generator:
    a uniformly random subset in random order (DESC_PGD.m:84)."""

    def _sample(l, IJ, CoInd_ij, n_sample):
        return rng.choice(CoInd_ij, size=n_sample, replace=False)

    return _sample


def desc_pgd_literal(Ind, RijMat, iters, Gradient, sampler=None, verbose=False,
                     patience=30, return_state=False, forced_lists=None):
    """Literal restatement of DESC_PGD.m:19-261.

    Ind      (m,2) integer array, 1-based node ids, i<j, sorted as the reference
             requires (DESC_PGD.m:5).
    RijMat   (3,3,m) float64, RijMat[:,:,l] as in MATLAB.
    iters    params.iters (:170).
    Gradient object with GetStep(grad) (:207).
    sampler  callable (l, IJ, CoInd_ij, n_sample) -> array of n_sample distinct
             entries of CoInd_ij (1-based node ids) standing in for ``datasample``.
             ``l`` is the 1-based index among edges with cycles, ``IJ`` the 1-based
             edge index.
    forced_lists  test hook, not in the reference: {IJ: kept third vertices (1-based)} replaces
             the outcome of :83-85 for the listed edges (any non-empty subset of the common
             neighbours), so that hand-sized graphs can have absent mirror cycles.  The reference
             itself only samples edges with >= 30 common neighbours.

    Returns S_vec (m,) and, if return_state, a dict with the structure arrays and
    per-iteration traces.
    """
    Ind = np.asarray(Ind, dtype=np.int64)
    RijMat = np.asarray(RijMat, dtype=np.float64)
    Ind_i = Ind[:, 0]                                            # :19
    Ind_j = Ind[:, 1]                                            # :20
    n = int(Ind.max())                                           # :21
    m = Ind_i.shape[0]                                           # :22
    AdjMat = np.zeros((n, n))                                    # :23-24
    AdjMat[Ind_i - 1, Ind_j - 1] = 1
    AdjMat = AdjMat + AdjMat.T

    CoDeg = (AdjMat @ AdjMat) * AdjMat                           # :29
    CoDeg[(CoDeg == 0) & (AdjMat > 0)] = -1                      # :30
    CoDeg_low = np.tril(CoDeg, -1)                               # :31
    CoDeg_vec = CoDeg_low.flatten(order="F")                     # :32 column-major (:)
    CoDeg_vec = CoDeg_vec[CoDeg_vec != 0]                        # :34
    assert CoDeg_vec.shape[0] == m, "Ind must list every edge once with i<j"

    CoDeg_pos_ind = np.flatnonzero(CoDeg_vec > 0) + 1            # :36 (1-based)
    CoDeg_vec_pos = CoDeg_vec[CoDeg_pos_ind - 1]                 # :37

    n_sample = matlab_max_ignore_nan(                            # :43
        math.ceil(matlab_median(CoDeg_vec_pos) / 4) if CoDeg_vec_pos.size else float("nan"), 30)
    n_sample = int(n_sample)

    CoDeg_vec_pos_sampled = np.minimum(CoDeg_vec_pos, n_sample).astype(np.int64)   # :45
    if forced_lists:
        for t, IJ in enumerate(CoDeg_pos_ind):
            if int(IJ) in forced_lists:
                CoDeg_vec_pos_sampled[t] = len(forced_lists[int(IJ)])
    cum_ind = np.concatenate([[0], np.cumsum(CoDeg_vec_pos_sampled)]).astype(np.int64)  # :49
    m_pos = CoDeg_pos_ind.shape[0]                               # :50
    m_cycle = int(cum_ind[-1])                                   # :51

    CoDeg_pos_ind_long = np.zeros(m, dtype=np.int64)             # :53-54
    CoDeg_pos_ind_long[CoDeg_pos_ind - 1] = np.arange(1, m_pos + 1)

    Ind_ij = np.zeros(m_cycle, dtype=np.int64)                   # :56-58
    Ind_jk = np.zeros(m_cycle, dtype=np.int64)
    Ind_ki = np.zeros(m_cycle, dtype=np.int64)

    RijMat4d = np.zeros((3, 3, n, n))                            # :60
    IndMat = np.zeros((n, n), dtype=np.int64)
    for l in range(1, m + 1):                                    # :63-69
        i = Ind_i[l - 1]; j = Ind_j[l - 1]
        RijMat4d[:, :, i - 1, j - 1] = RijMat[:, :, l - 1]
        RijMat4d[:, :, j - 1, i - 1] = RijMat[:, :, l - 1].T
        IndMat[i - 1, j - 1] = l
        IndMat[j - 1, i - 1] = l

    Rjk0Mat = np.zeros((3, 3, m_cycle))                          # :71-75
    Rki0Mat = np.zeros((3, 3, m_cycle))
    IJK = np.zeros(m_cycle, dtype=np.int64)
    IKJ = np.zeros(m_cycle, dtype=np.int64)
    JKI = np.zeros(m_cycle, dtype=np.int64)

    IJK_Mat = np.zeros((n, m_pos), dtype=np.int64)               # :77

    if sampler is None:
        sampler = default_sampler(np.random.default_rng(0))

    for l in range(1, m_pos + 1):                                # :79-96
        IJ = CoDeg_pos_ind[l - 1]
        i = Ind_i[IJ - 1]; j = Ind_j[IJ - 1]
        CoInd_ij = np.flatnonzero(AdjMat[:, i - 1] * AdjMat[:, j - 1]) + 1     # :82
        if forced_lists and int(IJ) in forced_lists:
            kept = np.asarray(forced_lists[int(IJ)], dtype=np.int64)
            assert kept.size and np.all(np.isin(kept, CoInd_ij))
            CoInd_ij = kept
        elif CoInd_ij.shape[0] >= n_sample:                      # :83
            CoInd_ij = np.asarray(sampler(l, IJ, CoInd_ij, n_sample), dtype=np.int64)   # :84
            assert CoInd_ij.shape[0] == n_sample
        lo, hi = cum_ind[l - 1], cum_ind[l]                      # range (lo+1):hi
        Ind_ij[lo:hi] = IJ                                       # :86
        Ind_jk[lo:hi] = IndMat[j - 1, CoInd_ij - 1]              # :87
        Ind_ki[lo:hi] = IndMat[CoInd_ij - 1, i - 1]              # :88
        Rjk0Mat[:, :, lo:hi] = RijMat4d[:, :, j - 1, CoInd_ij - 1]   # :89
        Rki0Mat[:, :, lo:hi] = RijMat4d[:, :, CoInd_ij - 1, i - 1]   # :91
        IJK[lo:hi] = CoInd_ij                                    # :93
        IJK_Mat[0:CoDeg_vec_pos_sampled[l - 1], l - 1] = CoInd_ij    # :94

    IKJ_appears = np.zeros((n, m_pos), dtype=bool)               # :100
    JKI_appears = np.zeros((n, m_pos), dtype=bool)               # :102
    for l in range(1, m_pos + 1):                                # :103-127
        IJ = CoDeg_pos_ind[l - 1]
        i = Ind_i[IJ - 1]; j = Ind_j[IJ - 1]
        lo, hi = cum_ind[l - 1], cum_ind[l]
        cnt = hi - lo
        IK = CoDeg_pos_ind_long[IndMat[i - 1, IJK[lo:hi] - 1] - 1]      # :106
        range_l = np.arange(lo + 1, hi + 1)                      # :107 (1-based cycle ids)
        IK_cum = cum_ind[IK - 1]                                 # :110  cum_ind(IK)
        eq = (IJK_Mat[:, IK - 1] == j)                           # n x cnt
        # [J_ind,~] = find(...): column-major scan -> row index of each hit, by column
        cols, rows = np.nonzero(eq.T)                            # sorted by column then row
        J_ind = rows + 1                                         # :111
        IKJ_appears[0:cnt, l - 1] = eq.any(axis=0)               # :113
        mask = IKJ_appears[:, l - 1]
        # range_l(mask): logical mask longer than range_l, true entries all < cnt
        assert not mask[cnt:].any()
        sel = np.flatnonzero(mask[:cnt])
        IKJ[range_l[sel] - 1] = IK_cum[sel] + J_ind              # :116

        JK = CoDeg_pos_ind_long[IndMat[j - 1, IJK[lo:hi] - 1] - 1]      # :119
        JK_cum = cum_ind[JK - 1]                                 # :122
        eq = (IJK_Mat[:, JK - 1] == i)
        cols, rows = np.nonzero(eq.T)
        I_ind = rows + 1                                         # :123
        JKI_appears[0:cnt, l - 1] = eq.any(axis=0)               # :124
        mask = JKI_appears[:, l - 1]
        sel = np.flatnonzero(mask[:cnt])
        JKI[range_l[sel] - 1] = JK_cum[sel] + I_ind              # :125

    Rij0Mat = RijMat[:, :, Ind_ij - 1]                           # :129

    R_cycle0 = np.zeros((3, 3, m_cycle))                         # :133-143
    R_cycle = np.zeros((3, 3, m_cycle))
    for j in range(3):
        R_cycle0 = R_cycle0 + Rij0Mat[:, j:j + 1, :] * Rjk0Mat[j:j + 1, :, :]
    for j in range(3):
        R_cycle = R_cycle + R_cycle0[:, j:j + 1, :] * Rki0Mat[j:j + 1, :, :]

    R_trace = (R_cycle[0, 0, :] + R_cycle[1, 1, :] + R_cycle[2, 2, :]).reshape(m_cycle)   # :146
    S0_long = matlab_abs_acos((R_trace - 1) / 2) / np.pi         # :147
    S_vec = np.ones(m)                                           # :148

    wijk = np.ones(m_cycle)                                      # :151
    for l in range(1, m_pos + 1):                                # :152-157
        IJ = CoDeg_pos_ind[l - 1]
        lo, hi = cum_ind[l - 1], cum_ind[l]
        weight = wijk[lo:hi]
        wijk[lo:hi] = weight / np.sum(weight)
        S_vec[IJ - 1] = wijk[lo:hi] @ S0_long[lo:hi]

    sum_ikj = np.zeros(m_cycle)                                  # :164
    sum_jki = np.zeros(m_cycle)                                  # :165
    S_vec_last = S_vec.copy()                                    # :167
    learning_iters = iters                                       # :170
    rm = 1; proj = 1                                             # :171-172
    obj_vals = []
    avg_changes = []
    misses = 0                                                   # :181
    iters_run = 0
    for it in range(1, learning_iters + 1):                      # :182
        iters_run = it
        for l in range(1, m_pos + 1):                            # :185-191
            lo, hi = cum_ind[l - 1], cum_ind[l]
            cnt = hi - lo
            m1 = np.flatnonzero(IKJ_appears[:cnt, l - 1])
            m2 = np.flatnonzero(JKI_appears[:cnt, l - 1])
            # scalar RHS broadcast to the masked positions only (:189-190)
            sum_ikj[lo + m1] = np.sum(wijk[IKJ[lo + m1] - 1])
            sum_jki[lo + m2] = np.sum(wijk[JKI[lo + m2] - 1])

        grad_long = S_vec[Ind_jk - 1] + S_vec[Ind_ki - 1] + (sum_ikj + sum_jki) * S0_long   # :193

        for l in range(1, m_pos + 1):                            # :195-204
            nsample = CoDeg_vec_pos_sampled[l - 1]
            lo, hi = cum_ind[l - 1], cum_ind[l]
            grad = grad_long[lo:hi]
            nv = np.ones(nsample) / (nsample ** 0.5)             # :199
            if rm == 1:
                grad = grad - (grad @ nv) * nv                   # :201
            grad_long[lo:hi] = grad

        wijk = wijk + Gradient.GetStep(grad_long)                # :207
        for l in range(1, m_pos + 1):                            # :208-230
            IJ = CoDeg_pos_ind[l - 1]
            nsample = CoDeg_vec_pos_sampled[l - 1]
            lo, hi = cum_ind[l - 1], cum_ind[l]
            w_new = wijk[lo:hi].copy()
            if proj == 1:
                wijk[lo:hi] = project_simplex_literal(w_new)     # :215-224
            else:
                wijk[lo:hi] = wijk[lo:hi] / np.sum(wijk[lo:hi])
            S_vec[IJ - 1] = wijk[lo:hi] @ S0_long[lo:hi]         # :229

        average_change = float(np.mean(np.abs(S_vec - S_vec_last)))          # :232
        obj_vals.append(float(wijk @ (S_vec[Ind_jk - 1] + S_vec[Ind_ki - 1])))   # :233
        avg_changes.append(average_change)
        if verbose:                                              # :241
            print('iter %d: average change in S_vec %f, objective value: %f' %
                  (it, average_change, obj_vals[-1]))
        if it > 1 and obj_vals[-2] - obj_vals[-1] < 10 ** (-5):  # :243
            misses += 1
            if misses >= patience:                               # :245
                break
        else:
            misses = 0                                           # :255
        S_vec_last = S_vec.copy()                                # :257

    if not return_state:
        return S_vec
    state = dict(
        n=n, m=m, n_sample=n_sample, m_pos=m_pos, m_cycle=m_cycle,
        CoDeg_vec=CoDeg_vec.astype(np.int64), CoDeg_pos_ind=CoDeg_pos_ind,
        cum_ind=cum_ind, Ind_ij=Ind_ij, Ind_jk=Ind_jk, Ind_ki=Ind_ki, IJK=IJK,
        IKJ=IKJ, JKI=JKI, S0_long=S0_long, wijk=wijk,
        obj_vals=np.array(obj_vals), avg_changes=np.array(avg_changes),
        iters_run=iters_run,
    )
    return S_vec, state
