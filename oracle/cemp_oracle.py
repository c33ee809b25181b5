"""ORACLE -- test infrastructure only.  NumPy restatement of the reference's
``Algorithms/CEMP.m:24-132`` (small n).  PARITY UNPINNED by the reference (no fixtures; MATLAB
cannot run here).  ``datasample(..., nsample)`` (with replacement, MATLAB global RNG, CEMP.m:64)
is replaced by the deterministic rule shared with the product: the t-th sample of edge l is
``CoInd[sample_key(seed, l, t) mod codeg]`` (CoInd ascending)."""
import numpy as np

from oracle.desc_pgd_literal import matlab_abs_acos
from oracle.oracle import sample_key


def cemp_oracle(Ind, RijMat, max_iter, reweighting, nsample, seed=0):
    Ind = np.asarray(Ind, dtype=np.int64)
    T = int(max_iter)
    beta_cemp = list(np.asarray(reweighting, dtype=np.float64).reshape(-1))
    if len(beta_cemp) < T:                                              # :30-34
        beta_cemp = beta_cemp + [beta_cemp[-1]] * (T - len(beta_cemp))
    Ind_i, Ind_j = Ind[:, 0], Ind[:, 1]
    n = int(Ind.max()); m = Ind.shape[0]
    AdjMat = np.zeros((n, n)); AdjMat[Ind_i - 1, Ind_j - 1] = 1; AdjMat = AdjMat + AdjMat.T     # :41-42
    CoDeg = (AdjMat @ AdjMat) * AdjMat                                  # :49
    IndPosbin = np.array([CoDeg[Ind_i[l] - 1, Ind_j[l] - 1] > 0 for l in range(m)])              # :50-58
    IndPos = np.flatnonzero(IndPosbin)
    CoIndMat = np.zeros((nsample, m), dtype=np.int64)
    for l in IndPos:                                                    # :62-65
        i, j = Ind_i[l], Ind_j[l]
        co = np.flatnonzero(AdjMat[:, i - 1] * AdjMat[:, j - 1]) + 1
        CoIndMat[:, l] = [co[sample_key(seed, int(l), t) % len(co)] for t in range(nsample)]
    RijMat4d = np.zeros((3, 3, n, n)); IndMat = np.zeros((n, n), dtype=np.int64)
    for l in range(m):                                                  # :71-77
        i, j = Ind_i[l], Ind_j[l]
        RijMat4d[:, :, i - 1, j - 1] = RijMat[:, :, l]
        RijMat4d[:, :, j - 1, i - 1] = RijMat[:, :, l].T
        IndMat[i - 1, j - 1] = l + 1; IndMat[j - 1, i - 1] = -(l + 1)
    S0Mat = np.zeros((nsample, m))
    for l in IndPos:                                                    # :80-101
        i, j = Ind_i[l], Ind_j[l]
        for s in range(nsample):
            k = CoIndMat[s, l]
            Rc = RijMat[:, :, l] @ RijMat4d[:, :, j - 1, k - 1] @ RijMat4d[:, :, k - 1, i - 1]
            S0Mat[s, l] = matlab_abs_acos(np.array([(np.trace(Rc) - 1) / 2]))[0] / np.pi
    SVec = S0Mat.mean(axis=0)                                           # :102
    SVec[~IndPosbin] = 1                                                # :103
    for it in range(T):                                                 # :107-128
        beta = beta_cemp[it]
        Ski = np.zeros((nsample, m)); Sjk = np.zeros((nsample, m))
        for l in IndPos:
            i, j = Ind_i[l], Ind_j[l]
            Ski[:, l] = SVec[np.abs(IndMat[i - 1, CoIndMat[:, l] - 1]) - 1]
            Sjk[:, l] = SVec[np.abs(IndMat[j - 1, CoIndMat[:, l] - 1]) - 1]
        WeightMat = np.exp(-beta * (Ski + Sjk))
        WeightMat = WeightMat / WeightMat.sum(axis=0)
        SVec = (WeightMat * S0Mat).sum(axis=0)
        SVec[~IndPosbin] = 1
    return SVec
