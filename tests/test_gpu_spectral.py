"""GPU parity of the Spectral / GCW eigen-solve (next row f-1) against the dense LAPACK oracle.
Rotations are gauge-free only up to one global right rotation, so they are compared after
Rotation_Alignment (Utils/Rotation_Alignment.m); tolerance 1e-8 degrees-equivalent
(subspace iteration to relative residual 1e-12 vs LAPACK eigenvectors)."""
import numpy as np
import pytest

from desc_amd import GCW, Rotation_Alignment, Spectral
from desc_amd.models import Nonuniform_Topology, Uniform_Topology
from oracle.spectral_oracle import gcw_oracle, rotation_alignment, spectral_oracle

pytestmark = pytest.mark.gpu


def aligned_diff(R, R_ref):
    R_out, _, mean_deg, _ = rotation_alignment(R, R_ref)
    return float(np.abs(R_out - R_ref).max()), mean_deg


@pytest.mark.parametrize("gen", [lambda: Uniform_Topology(40, 0.5, 0.2, 0.1, "uniform", seed=1),
                                 lambda: Uniform_Topology(100, 0.5, 0.2, 0.1, "uniform", seed=2),
                                 lambda: Uniform_Topology(150, 0.3, 0.4, 0.2, "self-consistent", seed=3),
                                 lambda: Nonuniform_Topology(90, 0.4, 0.5, 0.5, 0.1, 0.1, "uniform", seed=4)])
def test_spectral_matches_dense_oracle(gen):
    mo = gen()
    R, info = Spectral(mo.Ind, mo.RijMat, return_info=True)
    assert info["converged"], info
    R_ref = spectral_oracle(mo.Ind, mo.RijMat)
    diff, _ = aligned_diff(R, R_ref)
    assert diff < 1e-8, (diff, info)
    # valid rotations
    Rm = np.transpose(R, (2, 0, 1))
    assert np.abs(Rm @ np.transpose(Rm, (0, 2, 1)) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(Rm) - 1).max() < 1e-12
    # same accuracy against the ground truth as the reference construction
    e1 = Rotation_Alignment(R, mo.R_orig)[2]; e2 = rotation_alignment(R_ref, mo.R_orig)[2]
    assert abs(e1 - e2) < 1e-6


@pytest.mark.parametrize("seed,kind", [(5, "pgd"), (6, "pgd"), (7, "noisy_truth")])
def test_gcw_matches_dense_oracle(oracle, seed, kind):
    """S_vec as DESC uses it (the PGD output, DESC.m:263) and a harsher synthetic one whose
    weights 1/(s^1.5+1e-8) span 4-5 orders of magnitude (clustered top of the spectrum)."""
    from desc_amd.algorithms import marshal_edges
    mo = Uniform_Topology(80, 0.5, 0.25, 0.1, "uniform", seed=seed)
    if kind == "pgd":
        nn, ii, jj, rij, _ = marshal_edges(mo.Ind, mo.RijMat)
        st = oracle.build_structure(nn, ii, jj, seed=0)
        S = oracle.pgd_run(st, oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st), 100, lr=0.01)["S_vec"]
    else:
        rng = np.random.default_rng(seed)
        S = np.clip(mo.ErrVec + 0.02 * rng.standard_normal(mo.ErrVec.shape), 2e-3, 1)
    R, info = GCW(mo.Ind, mo.AdjMat, mo.RijMat, S, return_info=True)
    assert info["converged"], info
    R_ref = gcw_oracle(mo.Ind, mo.RijMat, S)
    diff, _ = aligned_diff(R, R_ref)
    assert diff < 1e-8, (diff, info)
    assert Rotation_Alignment(R, mo.R_orig)[2] < Rotation_Alignment(Spectral(mo.Ind, mo.RijMat), mo.R_orig)[2]


def test_rotation_alignment_restatements_agree():
    mo = Uniform_Topology(30, 0.6, 0.2, 0.1, "uniform", seed=7)
    R = spectral_oracle(mo.Ind, mo.RijMat)
    a = Rotation_Alignment(R, mo.R_orig); b = rotation_alignment(R, mo.R_orig)
    assert np.abs(a[0] - b[0]).max() < 1e-13 and abs(a[2] - b[2]) < 1e-10 and abs(a[3] - b[3]) < 1e-10


def test_gcw_on_device_problem_matches_dense_oracle(oracle):
    """desc_gcw_run_dev (what DESC() calls): weights 1/(S^1.5 + 1e-8) and weighted degrees formed on the device from S_vec, on a
    device-resident problem shared with a Spectral call -- against the dense restatement of GCW.m and against the host-weights path."""
    from desc_amd import _lib
    from desc_amd.algorithms import marshal_edges
    mo = Uniform_Topology(90, 0.5, 0.25, 0.1, "uniform", seed=8)
    nn, ii, jj, rij, _ = marshal_edges(mo.Ind, mo.RijMat)
    st = oracle.build_structure(nn, ii, jj, seed=0)
    S = oracle.pgd_run(st, oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st), 100, lr=0.01)["S_vec"]
    dp = _lib.DeviceProblem(_lib.ProblemArrays(nn, ii, jj, rij), 0)
    R, info = _lib.gcw_run(dp, S)
    Rs, sinfo = _lib.spectral_run(dp)                               # the same resident problem serves Spectral
    dp.free()
    assert info["converged"] and sinfo["converged"]
    R_ref = gcw_oracle(mo.Ind, mo.RijMat, S)
    assert aligned_diff(R, R_ref)[0] < 1e-8
    assert aligned_diff(Rs, spectral_oracle(mo.Ind, mo.RijMat))[0] < 1e-8
    R_host = GCW(mo.Ind, mo.AdjMat, mo.RijMat, S)
    assert aligned_diff(R, R_host)[0] < 1e-10


def test_block_spmm_forms_agree():
    """The block SpMM of the connection matrix (Spectral.m:27-37 as a sparse product) in its two device forms -- vector FMA
    (production) and v_mfma_f64_4x4x4 (measurement variant) -- on the same operand: equal to summation-order
    round-off; the MFMA operand layout the variant is written for must be the one the unit-vector probe finds on this GPU."""
    from desc_amd import _lib
    from desc_amd.algorithms import marshal_edges
    mo = Uniform_Topology(300, 0.4, 0.2, 0.1, "uniform", seed=5)
    n, ii, jj, rij, _ = marshal_edges(mo.Ind, mo.RijMat)
    dp = _lib.DeviceProblem(_lib.ProblemArrays(n, ii, jj, rij), 0)
    try:
        v = _lib.spmm_variants(dp, reps=3)
    finally:
        dp.free()
    assert v["mfma_layout"] == 1 and v["ms_valu"] > 0 and v["ms_mfma"] > 0
    assert v["max_abs_diff"] < 1e-11
