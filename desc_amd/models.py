"""Synthetic SO(3) synchronisation problems with the distributions of the
reference's ``Models/Uniform_Topology.m`` and ``Models/Nonuniform_Topology.m``.

MATLAB's RNG streams cannot be matched, so these are *seeded re-statements of the
same distributions* (NumPy ``default_rng``), not bit-copies of a MATLAB run.  The
output struct has the reference's field names (``Uniform_Topology.m:104-109``):
``Ind`` (m x 2, 1-based, i<j, sorted (1,2),(1,3),...,(2,3),...), ``RijMat``
(3 x 3 x m), ``Rij_orig``, ``R_orig`` (3 x 3 x n), ``ErrVec`` (m,), ``AdjMat``.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np


def _project_so3(Q):
    """U*diag(1,1,det(U*V'))*V' for a stack Q (...,3,3) (Uniform_Topology.m:42-44)."""
    U, _, Vt = np.linalg.svd(Q)
    d = np.linalg.det(U @ Vt)
    U = U.copy()
    U[..., :, 2] *= d[..., None]
    return U @ Vt


def _haar(rng, count):
    return _project_so3(rng.standard_normal((count, 3, 3)))


def _abs_acos_ext(x):
    out = np.empty_like(x)
    inside = np.abs(x) <= 1
    out[inside] = np.arccos(x[inside])
    out[x > 1] = np.arccosh(x[x > 1])
    out[x < -1] = np.hypot(np.pi, np.arccosh(-x[x < -1]))
    return out


def _er_graph(rng, n, p):
    """G = tril(rand(n)<p,-1); [Ind_j,Ind_i] = find(G==1)  (Uniform_Topology.m:29-34).

    Returns 1-based (Ind_i, Ind_j) with Ind_i < Ind_j sorted by (Ind_i, Ind_j),
    generated row-block-wise so n = 10^4 does not need an n x n float matrix."""
    ii, jj = [], []
    for c in range(n - 1):                      # column c of the lower triangle: rows r > c
        rows = np.flatnonzero(rng.random(n - 1 - c) < p) + c + 1
        if rows.size:
            ii.append(np.full(rows.size, c, dtype=np.int64))
            jj.append(rows.astype(np.int64))
    if not ii:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return np.concatenate(ii) + 1, np.concatenate(jj) + 1


class _Model(SimpleNamespace):
    @property
    def AdjMat(self):
        n = self.n
        A = np.zeros((n, n))
        A[self.Ind[:, 0] - 1, self.Ind[:, 1] - 1] = 1
        return A + A.T


def _finish(n, Ind_i, Ind_j, Rm, Rij_orig, R_orig, extra=None):
    m = Ind_i.shape[0]
    # ErrVec = abs(acos((trace(Rij_orig' * RijMat) - 1)/2))/pi  (Uniform_Topology.m:94-101)
    tr = np.einsum("lrc,lrc->l", Rij_orig, Rm) if m else np.zeros(0)
    ErrVec = _abs_acos_ext((tr - 1.0) / 2.0) / np.pi
    out = _Model(n=n, Ind=np.stack([Ind_i, Ind_j], axis=1),
                 RijMat=np.asfortranarray(np.transpose(Rm, (1, 2, 0))),
                 Rij_orig=np.asfortranarray(np.transpose(Rij_orig, (1, 2, 0))),
                 R_orig=np.asfortranarray(np.transpose(R_orig, (1, 2, 0))),
                 ErrVec=ErrVec)
    if extra:
        for k, v in extra.items():
            setattr(out, k, v)
    return out


def Uniform_Topology(n, p, q, sigma, model="uniform", seed=0):
    """Reference: Models/Uniform_Topology.m:24-110.

    ``model``: 'uniform' or 'self-consistent'.  As in the reference (``:76,83``) any
    string other than 'uniform' selects the self-consistent branch."""
    rng = np.random.default_rng(seed)
    Ind_i, Ind_j = _er_graph(rng, n, p)
    m = Ind_i.shape[0]
    R_orig = _haar(rng, n)                                           # :40-45
    Rij_orig = R_orig[Ind_i - 1] @ np.transpose(R_orig[Ind_j - 1], (0, 2, 1))   # :48-51
    Rm = Rij_orig.copy()
    noiseIndLog = rng.random(m) >= q                                 # :53
    noiseInd = np.flatnonzero(noiseIndLog)
    corrInd = np.flatnonzero(~noiseIndLog)
    if noiseInd.size:
        Rm[noiseInd] = _project_so3(Rm[noiseInd] + sigma * rng.standard_normal((noiseInd.size, 3, 3)))  # :58-65
    R_corr = _haar(rng, n)                                           # :69-74
    if corrInd.size:
        if model == "uniform":                                       # :76-82
            Rm[corrInd] = _haar(rng, corrInd.size)
        else:                                                        # :84-90
            Q = R_corr[Ind_i[corrInd] - 1] @ np.transpose(R_corr[Ind_j[corrInd] - 1], (0, 2, 1)) \
                + sigma * rng.standard_normal((corrInd.size, 3, 3))
            Rm[corrInd] = _project_so3(Q)
    return _finish(n, Ind_i, Ind_j, Rm, Rij_orig, R_orig, dict(corrupted=~noiseIndLog))


def Nonuniform_Topology(n, p, p_node_crpt, p_edge_crpt, sigma_in, sigma_out, crpt_type="uniform", seed=0):
    """Reference: Models/Nonuniform_Topology.m:26-156 ('uniform' | 'self-consistent' | 'adv')."""
    rng = np.random.default_rng(seed)
    Ind_i, Ind_j = _er_graph(rng, n, p)
    m = Ind_i.shape[0]
    R_orig = _haar(rng, n)
    Rij_orig = R_orig[Ind_i - 1] @ np.transpose(R_orig[Ind_j - 1], (0, 2, 1))
    Rm = Rij_orig.copy()
    node_crpt = rng.permutation(n)[: int(np.floor(n * p_node_crpt))] + 1       # :60-62
    crptInd = np.zeros(m, dtype=bool)
    R_crpt = _haar(rng, n)                                                     # :66-71
    # incident edge lists: Ind_full(Ind_full(:,1)==i, 2) lists first the neighbours j with
    # (j,i) = (Ind_j,Ind_i) rows i.e. smaller neighbours, then the larger ones (:37,77)
    order_lo = np.argsort(Ind_j, kind="stable")
    lo_ptr = np.searchsorted(Ind_j[order_lo], np.arange(1, n + 2))
    hi_ptr = np.searchsorted(Ind_i, np.arange(1, n + 2))
    for i in node_crpt:                                                        # :76-118
        e_lo = order_lo[lo_ptr[i - 1]:lo_ptr[i]]          # edges (x, i), x < i   -> IndMat(i,x) = -k
        e_hi = np.arange(hi_ptr[i - 1], hi_ptr[i])        # edges (i, x), x > i   -> IndMat(i,x) = +k
        cand_e = np.concatenate([e_lo, e_hi])
        cand_sign = np.concatenate([-np.ones(e_lo.size), np.ones(e_hi.size)])
        perm = rng.permutation(cand_e.size)[: int(np.floor(p_edge_crpt * cand_e.size))]   # :79-81
        for t in perm:                                                         # :84-116
            k = cand_e[t]; pos = cand_sign[t] > 0
            jn = (Ind_j[k] if pos else Ind_i[k])
            crptInd[k] = True
            R0 = _haar(rng, 1)[0]
            if crpt_type == "uniform":
                M = R0
            elif crpt_type == "self-consistent":
                M = R_crpt[i - 1] @ R_crpt[jn - 1].T
            elif crpt_type == "adv":
                M = R_crpt[i - 1] @ R_orig[jn - 1].T
            else:
                continue
            Rm[k] = M if pos else M.T
    noise = ~crptInd                                                           # :121-127
    Rm[noise] = Rm[noise] + sigma_in * rng.standard_normal((int(noise.sum()), 3, 3))
    Rm[crptInd] = Rm[crptInd] + sigma_out * rng.standard_normal((int(crptInd.sum()), 3, 3))
    if m:
        Rm = _project_so3(Rm)                                                  # :133-137
    return _finish(n, Ind_i, Ind_j, Rm, Rij_orig, R_orig, dict(corrupted=crptInd))
