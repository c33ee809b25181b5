"""ORACLE -- test infrastructure only.  Dense NumPy restatements of the reference's
``Algorithms/Spectral.m``, ``Utils/GCW.m`` and ``Utils/Rotation_Alignment.m`` (small n).

PARITY UNPINNED by the reference (no fixtures, MATLAB cannot run here).  MATLAB's
``eigs(A,3,'la')`` is restated with LAPACK (``numpy.linalg.eigh`` for the symmetric
Spectral matrix, ``scipy.linalg.eig`` for GCW's row-normalised, non-symmetric one): same
eigenvectors up to sign, unit 2-norm columns; sign and basis ambiguities cancel in the
gauge-invariant comparison through Rotation_Alignment.
"""
import numpy as np
import scipy.linalg


def _blk(Ind, RijMat, n):
    """Rij_blk of Spectral.m:27-33 / GCW.m:9-15."""
    d = 3
    B = np.zeros((n * d, n * d))
    for k in range(Ind.shape[0]):
        i, j = int(Ind[k, 0]) - 1, int(Ind[k, 1]) - 1
        B[d * i:d * i + d, d * j:d * j + d] = RijMat[:, :, k]
    return B + B.T


def _project(V, n):
    """sign fix + per-node SVD projection (Spectral.m:39-46, GCW.m:28-35)."""
    d = 3
    V = V.copy()
    V[:, 0] = V[:, 0] * np.sign(np.linalg.det(V[0:d, :]))
    R = np.zeros((d, d, n))
    for i in range(n):
        Ur, _, Vt = np.linalg.svd(V[d * i:d * i + d, :])
        S0 = np.diag([1.0, 1.0, np.linalg.det(Ur @ Vt)])
        R[:, :, i] = Ur @ S0 @ Vt
    return R


def spectral_oracle(Ind, RijMat):
    """Algorithms/Spectral.m:15-47."""
    Ind = np.asarray(Ind)
    n = int(Ind.max())
    B = _blk(Ind, RijMat, n)
    lam, vec = np.linalg.eigh(B)
    V = vec[:, ::-1][:, :3]                        # eigs(..., 3, 'la'): descending
    return _project(V, n)


def gcw_oracle(Ind, RijMat, S_vec):
    """Utils/GCW.m:1-38 (AdjMat is implied by Ind)."""
    Ind = np.asarray(Ind)
    n = int(Ind.max())
    B = _blk(Ind, RijMat, n)
    A = np.zeros((n, n)); Sm = np.zeros((n, n))
    A[Ind[:, 0] - 1, Ind[:, 1] - 1] = 1; A = A + A.T
    Sm[Ind[:, 0] - 1, Ind[:, 1] - 1] = S_vec; Sm = Sm + Sm.T
    W = (1.0 / (Sm ** 1.5 + 1e-8)) * A                       # :20
    W = np.diag(1.0 / W.sum(axis=1)) @ W                     # :21
    RijW = B * np.kron(W, np.ones((3, 3)))                   # :22-23
    lam, vec = scipy.linalg.eig(RijW)
    order = np.argsort(-lam.real)[:3]                        # 'la'
    V = np.real(vec[:, order])
    V = V / np.linalg.norm(V, axis=0)
    return _project(V, n)


def rotation_alignment(R_est, R_gt):
    """Utils/Rotation_Alignment.m:13-38 -> (R_out, R_align, mean_error_deg, median_error_deg)."""
    d, n = R_gt.shape[0], R_gt.shape[2]
    A = np.zeros((d, d))
    for k in range(n):
        A = A + R_est[:, :, k].T @ R_gt[:, :, k]
    U1, _, V1t = np.linalg.svd(A)
    D = np.eye(d); D[-1, -1] = np.linalg.det(U1 @ V1t)
    R_align = U1 @ D @ V1t
    R_out = np.zeros_like(R_est)
    err = np.zeros(n)
    for k in range(n):
        R_out[:, :, k] = R_est[:, :, k] @ R_align
        tr = np.trace(R_gt[:, :, k] @ R_out[:, :, k].T)
        x = (tr - 1) / 2
        err[k] = abs(np.arccos(np.clip(x, -1, 1))) / np.pi * 180 if abs(x) <= 1 else abs(np.arccos(complex(x))) / np.pi * 180
    return R_out, R_align, float(err.mean()), float(np.median(err))
