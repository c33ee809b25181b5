"""GPU parity of CEMP (next row f-2) against the NumPy restatement of Algorithms/CEMP.m.
Tolerance 1e-12 on SVec (f64; differences: summation order and libm exp/acos rounding)."""
import numpy as np
import pytest

from desc_amd import CEMP
from desc_amd.models import Uniform_Topology
from oracle.cemp_oracle import cemp_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p,nsample,seed", [(40, 0.5, 50, 1), (70, 0.3, 20, 2), (25, 0.6, 80, 3)])
def test_cemp_matches_oracle(n, p, nsample, seed):
    mo = Uniform_Topology(n, p, 0.2, 0.1, "uniform", seed=seed)
    params = dict(max_iter=6, reweighting=2.0 ** np.arange(6), nsample=nsample, seed=seed)       # Demo/compare_algorithms.m:26-28
    S = CEMP(mo.Ind, mo.RijMat, params)
    ref = cemp_oracle(mo.Ind, mo.RijMat, 6, params["reweighting"], nsample, seed=seed)
    assert np.abs(S - ref).max() < 1e-12
    assert np.mean(np.abs(S - mo.ErrVec)) < 0.06


def test_cemp_short_beta_vector_and_no_cycles():
    mo = Uniform_Topology(30, 0.5, 0.2, 0.1, "uniform", seed=4)
    S = CEMP(mo.Ind, mo.RijMat, dict(max_iter=5, reweighting=[1.0, 4.0], nsample=30))
    ref = cemp_oracle(mo.Ind, mo.RijMat, 5, [1.0, 4.0], 30)
    assert np.abs(S - ref).max() < 1e-12
    Ind = np.array([[1, 2], [2, 3], [3, 4]])
    R = np.repeat(np.eye(3)[:, :, None], 3, axis=2)
    assert np.array_equal(CEMP(Ind, R, dict(max_iter=3, reweighting=[1.0], nsample=10)), np.ones(3))


@pytest.mark.parametrize("bi,jb", [(1, 32), (3, 32), (7, 50), (16, 40)])
def test_cemp_tile_shapes_equal_plain_rounds(bi, jb, monkeypatch):
    """The tile kernel of the rounds (CSR-aligned S, band rows in the LDS, j-block-major tiles) against the plain wave-per-edge kernel on the same
    samples, for tile shapes whose bands straddle the j-blocks' ends: the same arithmetic in the same order, bit for bit (CEMP.m:107-128)."""
    mo = Uniform_Topology(260, 0.5, 0.2, 0.1, "uniform", seed=9)
    params = dict(max_iter=6, reweighting=2.0 ** np.arange(6), nsample=50, seed=5)
    monkeypatch.setenv("DESC_DEBUG_CEMP_TILES", "0")
    plain = CEMP(mo.Ind, mo.RijMat, params)
    monkeypatch.setenv("DESC_DEBUG_CEMP_TILES", "1")
    monkeypatch.setenv("DESC_DEBUG_CEMP_BI", str(bi))
    monkeypatch.setenv("DESC_DEBUG_CEMP_JB", str(jb))
    tiles = CEMP(mo.Ind, mo.RijMat, params)
    assert np.array_equal(plain, tiles)
