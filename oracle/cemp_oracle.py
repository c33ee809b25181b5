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


def _mix64_v(x):
    x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
    x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def cemp_oracle_batched(Ind, RijMat, max_iter, reweighting, nsample, seed=0, chunk=4096):
    """The same restatement of CEMP.m:24-132 with the per-edge loops batched over chunks of edges, for full-size graphs
    (C2: 2.5e5 edges x 50 samples).  Array shapes, reduction axes and the arithmetic of every line are those of
    ``cemp_oracle`` (S0Mat is nsample x m, sums run down its columns); tests/test_oracle_next_rows.py checks the two
    agree on small graphs."""
    Ind = np.asarray(Ind, dtype=np.int64)
    T = int(max_iter)
    beta_cemp = list(np.asarray(reweighting, dtype=np.float64).reshape(-1))
    if len(beta_cemp) < T:                                              # :30-34
        beta_cemp = beta_cemp + [beta_cemp[-1]] * (T - len(beta_cemp))
    Ind_i, Ind_j = Ind[:, 0] - 1, Ind[:, 1] - 1
    n = int(Ind.max()); m = Ind.shape[0]
    A = np.zeros((n, n), dtype=bool); A[Ind_i, Ind_j] = True; A |= A.T   # :41-42
    eid = np.full((n, n), -1, dtype=np.int64); eid[Ind_i, Ind_j] = np.arange(m); eid[Ind_j, Ind_i] = np.arange(m)   # |IndMat| - 1  (:75-76)
    R = np.ascontiguousarray(np.transpose(np.asarray(RijMat, dtype=np.float64), (2, 0, 1)))      # (m, 3, 3)
    S0Mat = np.zeros((nsample, m)); Eki = np.zeros((nsample, m), dtype=np.int64); Ejk = np.zeros((nsample, m), dtype=np.int64)
    IndPosbin = np.zeros(m, dtype=bool)
    tt = np.arange(nsample, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        for a in range(0, m, chunk):
            b = min(m, a + chunk)
            i, j = Ind_i[a:b], Ind_j[a:b]
            common = A[i] & A[j]                                        # :49, :63 (ascending k)
            codeg = common.sum(axis=1)
            pos = codeg > 0
            IndPosbin[a:b] = pos
            if not pos.any():
                continue
            rows, ks = np.nonzero(common)
            start = np.concatenate([[0], np.cumsum(codeg)])[:-1]
            l = np.arange(a, b, dtype=np.uint64)[None, :]
            key = _mix64_v(_mix64_v(np.uint64(seed) ^ ((l + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))) ^ ((tt + np.uint64(1)) * np.uint64(0xD1B54A32D192ED03)))
            idx = (key % np.maximum(codeg, 1).astype(np.uint64)[None, :]).astype(np.int64)       # :64 with the keyed stand-in
            k = ks[np.minimum(start[None, :] + idx, len(ks) - 1)]                               # (nsample, chunk)
            k[:, ~pos] = 0
            ejk, eki = eid[j[None, :], k], eid[i[None, :], k]
            Rjk = np.where((j[None, :] < k)[..., None, None], R[ejk], np.transpose(R[ejk], (0, 1, 3, 2)))    # RijMat4d(:,:,j,k)  (:73-74)
            Rki = np.where((k < i[None, :])[..., None, None], R[eki], np.transpose(R[eki], (0, 1, 3, 2)))    # RijMat4d(:,:,k,i)
            Rc = np.matmul(np.matmul(R[a:b][None], Rjk), Rki)                                     # :84-96
            tr = Rc[..., 0, 0] + Rc[..., 1, 1] + Rc[..., 2, 2]
            s0 = matlab_abs_acos(((tr - 1) / 2).reshape(-1)).reshape(tr.shape) / np.pi
            s0[:, ~pos] = 0
            S0Mat[:, a:b] = s0; Eki[:, a:b] = np.where(pos[None, :], eki, 0); Ejk[:, a:b] = np.where(pos[None, :], ejk, 0)
    SVec = S0Mat.mean(axis=0)                                           # :102
    SVec[~IndPosbin] = 1                                                # :103
    for it in range(T):                                                 # :107-128
        W = np.exp(-beta_cemp[it] * (SVec[Eki] + SVec[Ejk]))
        W = W / W.sum(axis=0)
        SVec = (W * S0Mat).sum(axis=0)
        SVec[~IndPosbin] = 1
    return SVec
