"""ORACLE -- test infrastructure only.  NumPy restatement of the DESC refinement tail
(``Algorithms/DESC.m:265-313``, ``Utils/Weighted_LAA.m``, ``Build_Amatrix.m``, ``R2Q.m``,
``q2R.m``), dense, small n.  PARITY UNPINNED by the reference (no fixtures; MATLAB cannot
run here).  MATLAB's sparse-QR backslash is restated with ``numpy.linalg.lstsq``; ``quantile``
with NumPy's 'hazen' method (MATLAB's definition)."""
import numpy as np


def R2Q(Rot):
    """Utils/R2Q.m:7-14; Rot: 3 x 3 x N -> N x 4."""
    q = np.stack([Rot[0, 0] + Rot[1, 1] + Rot[2, 2] - 1, Rot[2, 1] - Rot[1, 2], Rot[0, 2] - Rot[2, 0], Rot[1, 0] - Rot[0, 1]], axis=1) / 2
    q[:, 0] = np.sqrt((q[:, 0] + 1) / 2)
    q[:, 1:4] = (q[:, 1:4] / q[:, 0:1]) / 2
    return q


def q2R(q):
    """Utils/q2R.m."""
    c2 = q[0]
    if abs(abs(c2) - 1) > 1e-12:
        s2 = np.linalg.norm(q[1:4]); s = 2 * s2 * c2; c = 2 * c2 * c2 - 1; n = q[1:4] / s2; cc = 1 - c
        n1, n2, n3 = n
        return np.array([[c + n1 * n1 * cc, n1 * n2 * cc - n3 * s, n3 * n1 * cc + n2 * s],
                         [n1 * n2 * cc + n3 * s, c + n2 * n2 * cc, n2 * n3 * cc - n1 * s],
                         [n3 * n1 * cc - n2 * s, n2 * n3 * cc + n1 * s, c + n3 * n3 * cc]])
    return np.eye(3)


def Build_Amatrix(I):
    """Utils/Build_Amatrix.m:6-13; I: 2 x m (1-based) -> dense m x (N-1)."""
    m = I.shape[1]; N = int(I.max())
    A = np.zeros((m, N - 1))
    for e in range(m):
        i, j = I[0, e], I[1, e]
        if i != 1: A[e, i - 2] = -1
        if j != 1: A[e, j - 2] = 1
    return A


def qmul(a, b):
    """row-wise quaternion product a*b as written in Weighted_LAA.m:11-13."""
    return np.concatenate([(a[:, 0:1] * b[:, 0:1] - np.sum(a[:, 1:4] * b[:, 1:4], axis=1, keepdims=True)),
                           a[:, 0:1] * b[:, 1:4] + b[:, 0:1] * a[:, 1:4] +
                           np.stack([a[:, 2] * b[:, 3] - a[:, 3] * b[:, 2], a[:, 3] * b[:, 1] - a[:, 1] * b[:, 3], a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]], axis=1)], axis=1)


def Weighted_LAA(I, Q, QQ, Amatrix, Weights):
    """Utils/Weighted_LAA.m:4-51."""
    N = int(I.max())
    i = I[0] - 1; j = I[1] - 1
    w = qmul(QQ, Q[i])                                                             # :11-13
    Qj = Q[j]
    w = np.concatenate([(-Qj[:, 0:1] * w[:, 0:1] - np.sum(Qj[:, 1:4] * w[:, 1:4], axis=1, keepdims=True)),     # :16-18
                        -Qj[:, 0:1] * w[:, 1:4] + w[:, 0:1] * Qj[:, 1:4] +
                        np.stack([Qj[:, 2] * w[:, 3] - Qj[:, 3] * w[:, 2], Qj[:, 3] * w[:, 1] - Qj[:, 1] * w[:, 3], Qj[:, 1] * w[:, 2] - Qj[:, 2] * w[:, 1]], axis=1)], axis=1)
    s2 = np.sqrt(np.sum(w[:, 1:4] ** 2, axis=1))
    w[:, 0] = 2 * np.arctan2(s2, w[:, 0])
    w[w[:, 0] < -np.pi, 0] += 2 * np.pi
    w[w[:, 0] >= np.pi, 0] -= 2 * np.pi
    with np.errstate(divide="ignore", invalid="ignore"):
        B = w[:, 1:4] * (w[:, 0] / s2)[:, None]
    B[np.isnan(B)] = 0                                                            # :35
    W = np.zeros((N, 4)); W[0] = [1, 0, 0, 0]
    W[1:, 1:4] = np.linalg.lstsq(Weights[:, None] * Amatrix, Weights[:, None] * B, rcond=None)[0]      # :38
    score = np.sum(np.sqrt(np.sum(W[1:, 1:4] ** 2, axis=1))) / N                  # :40
    theta = np.sqrt(np.sum(W[:, 1:4] ** 2, axis=1))
    W[:, 0] = np.cos(theta / 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        W[:, 1:4] = W[:, 1:4] * (np.sin(theta / 2) / theta)[:, None]
    W[np.isnan(W)] = 0
    Q = qmul(Q, W)                                                                # :48-50
    return Q, W, B, score


def desc_refine_oracle(Ind, RijMat, S_vec, R_init, stop_threshold=1e-3, maxIters=100):
    """Algorithms/DESC.m:265-313 given S_vec and R_init (= GCW output)."""
    Ind = np.asarray(Ind); S_vec = np.asarray(S_vec, dtype=np.float64)
    n = int(Ind.max())
    RR = np.transpose(RijMat, (1, 0, 2))                                          # :265
    Ind_T = Ind.T
    Amatrix = Build_Amatrix(Ind_T)
    Q = R2Q(R_init); QQ = R2Q(RR)
    score = np.inf; Iteration = 1
    quant_ratio = 1.0; quant_ratio_min = 0.8
    thresh = np.quantile(S_vec, quant_ratio, method="hazen")                      # :276
    Weights = 1.0 / (S_vec ** 0.75)
    weight_max = 1e4; weight_min = 1e-4
    Weights[Weights > weight_max] = weight_max
    Weights[S_vec > thresh] = weight_min
    while score > stop_threshold and Iteration < maxIters:                        # :287
        lam = 1 / (Iteration + 1)
        Q, W, B, score = Weighted_LAA(Ind_T, Q, QQ, Amatrix, Weights)
        E = Amatrix @ W[1:, 1:4] - B
        ResVec = np.sqrt(np.sum(E ** 2, axis=1)) / np.pi
        RSVec = (1 - lam) * ResVec + lam * S_vec
        Weights = 1.0 / (RSVec ** 0.75)
        quant_ratio = max(quant_ratio_min, quant_ratio - 0.05)
        thresh = np.quantile(RSVec, quant_ratio, method="hazen")
        Weights[Weights > weight_max] = weight_max
        Weights[RSVec > thresh] = weight_min
        Iteration += 1
    R_est = np.zeros((3, 3, n))
    for i in range(n):
        R_est[:, :, i] = q2R(Q[i])
    return R_est, Iteration - 1, score
