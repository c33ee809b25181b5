"""CPU tests of the ORACLE itself (no GPU): hand-derived known answers from the .m text
(SURVEY.md 4), invariants, finite-difference gradient, and agreement of the two
independent restatements (dense literal NumPy vs sparse C).

The reference ships no fixtures and cannot be executed here, so parity is UNPINNED by the
reference; these tests are what pins the oracle instead."""
import numpy as np
import pytest

from desc_amd.models import Uniform_Topology, _haar
from oracle.desc_pgd_literal import (ConstantStepSize, HybridGradient, PiecewiseStepSize, desc_pgd_literal,
                                     matlab_abs_acos, project_simplex_literal)
from tests.helpers import make_problem


def cycle_d(R12, R23, R31):
    tr = np.trace(R12 @ R23 @ R31)
    return abs(np.arccos((tr - 1) / 2)) / np.pi


# ------------------------------------------------------------------ known answers
def test_simplex_projection_unit_vectors():
    P = project_simplex_literal
    assert np.allclose(P([0.5, 0.5]), [0.5, 0.5])
    assert np.allclose(P([1.2, -0.2]), [1.0, 0.0])
    assert np.allclose(P([0.4, 0.4, 0.4]), [1 / 3] * 3)
    assert np.allclose(P([7.0]), [1.0])
    assert np.allclose(P([0.2, 0.2, 0.2, 0.2, 0.2]), [0.2] * 5)
    assert np.allclose(P([3.0, 3.0, -1.0]), [0.5, 0.5, 0.0])       # ties
    rng = np.random.default_rng(0)
    for _ in range(50):
        v = rng.standard_normal(rng.integers(1, 40)) * rng.choice([0.01, 1, 10])
        p = P(v)
        assert abs(p.sum() - 1) < 1e-12 and (p >= 0).all()
        # Euclidean projection: p - v is constant on the support, >= that constant off it
        sup = p > 0
        shift = (p - v)[sup]
        assert np.ptp(shift) < 1e-12
        assert ((0 - v)[~sup] >= shift.mean() - 1e-12).all()


def test_abs_acos_complex_extension():
    x = np.array([-3.0, -1.0000001, -1.0, -0.3, 0.0, 0.9, 1.0, 1.0000001, 2.0, 5.5])
    ref = np.abs(np.arccos(x.astype(complex)))
    assert np.allclose(matlab_abs_acos(x), ref, rtol=1e-14, atol=1e-14)


def test_single_triangle():
    """m=3, every edge has one cycle: w == 1, S_vec == d for all three edges, objective 6d
    every iteration, early stop fires at iteration 31 (first miss at it=2, 30th at it=31)."""
    Rs = _haar(np.random.default_rng(1), 3)
    Ind = np.array([[1, 2], [1, 3], [2, 3]])
    Rm = np.stack([Rs[0], Rs[1], Rs[2]], axis=2)
    S, st = desc_pgd_literal(Ind, Rm, 100, ConstantStepSize(0.01), return_state=True)
    d = cycle_d(Rs[0], Rs[2], Rs[1].T)
    assert np.abs(S - d).max() < 1e-15
    assert st["iters_run"] == 31
    assert np.allclose(st["obj_vals"], 6 * d, rtol=1e-15)
    assert np.array_equal(st["wijk"], np.ones(3))
    assert st["n_sample"] == 30 and st["m_cycle"] == 3


def test_tree_has_no_cycles():
    Ind = np.array([[1, 2], [2, 3], [3, 4], [3, 5]])
    Rm = np.repeat(np.eye(3)[:, :, None], 4, axis=2)
    S, st = desc_pgd_literal(Ind, Rm, 100, ConstantStepSize(0.01), return_state=True)
    assert np.array_equal(S, np.ones(4))
    assert st["m_pos"] == 0 and st["n_sample"] == 30 and st["iters_run"] == 31


def test_consistent_graph_gives_zero_and_pendant_edge_stays_one():
    rng = np.random.default_rng(2)
    R = _haar(rng, 6)
    edges = [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4), (4, 5), (5, 6)]   # K4 + a path
    Ind = np.array(edges)
    Rm = np.stack([R[i - 1] @ R[j - 1].T for i, j in edges], axis=2)
    S = desc_pgd_literal(Ind, Rm, 20, ConstantStepSize(0.01))
    assert np.abs(S[:6]).max() < 1e-7          # acos near 1 amplifies rounding: sqrt(eps)
    assert np.array_equal(S[6:], [1.0, 1.0])   # no triangle: never overwritten (DESC_PGD.m:148)


def test_k4_one_corrupted_edge_first_iterations_by_hand():
    """K4, edge (1,2) corrupted: each edge lies in 2 cycles; iteration 1 tabulated from the
    formulas of DESC_PGD.m:185-230 independently of the restatement's loops."""
    rng = np.random.default_rng(3)
    R = _haar(rng, 4)
    edges = [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (3, 4)]
    Rel = {e: R[e[0] - 1] @ R[e[1] - 1].T for e in edges}
    Rel[(1, 2)] = _haar(rng, 1)[0]
    Ind = np.array(edges)
    Rm = np.stack([Rel[e] for e in edges], axis=2)

    def rot(a, b):
        return Rel[(a, b)] if a < b else Rel[(b, a)].T

    def d(a, b, c):
        return cycle_d(rot(a, b), rot(b, c), rot(c, a))

    # cycles per edge (third vertices ascending), all mirrors present (no sampling)
    third = {e: [k for k in (1, 2, 3, 4) if k not in e] for e in edges}
    dd = {(e, k): d(e[0], e[1], k) for e in edges for k in third[e]}
    w = {(e, k): 0.5 for e in edges for k in third[e]}
    S0 = {e: sum(w[(e, k)] * dd[(e, k)] for k in third[e]) for e in edges}

    def ed(a, b):
        return (a, b) if a < b else (b, a)

    lr = 0.01
    newS = {}
    neww = {}
    for e in edges:
        i, j = e
        T1 = sum(w[(ed(i, k), j)] for k in third[e])       # sum_k w(ik;j)
        T2 = sum(w[(ed(j, k), i)] for k in third[e])       # sum_k w(jk;i)
        g = np.array([S0[ed(j, k)] + S0[ed(k, i)] + (T1 + T2) * dd[(e, k)] for k in third[e]])
        g = g - g.mean()
        v = np.array([w[(e, k)] for k in third[e]]) - lr * g
        p = project_simplex_literal(v)
        for k, pv in zip(third[e], p):
            neww[(e, k)] = pv
        newS[e] = float(p @ np.array([dd[(e, k)] for k in third[e]]))
    obj1 = sum(neww[(e, k)] * (newS[ed(e[1], k)] + newS[ed(k, e[0])]) for e in edges for k in third[e])

    S, st = desc_pgd_literal(Ind, Rm, 1, ConstantStepSize(lr), return_state=True)
    assert np.allclose(S, [newS[e] for e in edges], rtol=0, atol=1e-15)
    assert abs(st["obj_vals"][0] - obj1) < 1e-13
    # the corrupted edge gets the largest corruption estimate after a few iterations
    S50 = desc_pgd_literal(Ind, Rm, 50, ConstantStepSize(lr))
    assert np.argmax(S50) == 0


def test_sampling_regime_known_answer_by_hand(oracle):
    """Mirror cycles absent (DESC_PGD.m:113,124 false): the per-edge scalar mirror sums of :189-190 reach
    only the positions whose mirror was sampled.  Two iterations tabulated by hand in tests/kat_sampling.py;
    both restatements must reproduce them, and the literal one must derive the same index structure."""
    from tests import kat_sampling as K
    Ind = np.array(K.EDGES)
    for iters, W, S, objs, avgs in ((1, K.W1, K.S1, [K.OBJ1], [K.AVG1]), (2, K.W2, K.S2, [K.OBJ1, K.OBJ2], [K.AVG1, K.AVG2])):
        Sv, st = desc_pgd_literal(Ind, K.rotations(), iters, ConstantStepSize(K.LR), return_state=True, forced_lists=K.FORCED)
        assert np.abs(st["S0_long"] - K.D).max() < K.TOL
        assert np.array_equal(st["cum_ind"], K.CUM_IND) and np.array_equal(st["IJK"] - 1, K.K)
        assert np.array_equal(st["Ind_jk"] - 1, K.E_JK) and np.array_equal(st["Ind_ki"] - 1, K.E_KI)
        assert np.array_equal(st["IKJ"] - 1, K.IKJ) and np.array_equal(st["JKI"] - 1, K.JKI)
        assert np.abs(st["wijk"] - W).max() < K.TOL and np.abs(Sv - S).max() < K.TOL
        assert np.abs(np.array(st["obj_vals"]) - objs).max() < K.TOL and np.abs(np.array(st["avg_changes"]) - avgs).max() < K.TOL
        sd = K.structure_dict()
        ii = Ind[:, 0].astype(np.int32) - 1; jj = Ind[:, 1].astype(np.int32) - 1
        rij = np.ascontiguousarray(np.transpose(K.rotations(), (2, 1, 0))).reshape(-1, 9)
        d = oracle.cycle_d(ii, jj, rij, sd)
        assert np.abs(d - K.D).max() < K.TOL
        res = oracle.pgd_run(sd, d, iters, lr=K.LR)
        assert np.abs(res["w"] - W).max() < K.TOL and np.abs(res["S_vec"] - S).max() < K.TOL
        assert np.abs(res["obj"] - objs).max() < K.TOL and np.abs(res["avg"] - avgs).max() < K.TOL
    # initial state (:148-157)
    Sv, st = desc_pgd_literal(Ind, K.rotations(), 0, ConstantStepSize(K.LR), return_state=True, forced_lists=K.FORCED)
    assert np.abs(st["wijk"] - K.W0).max() < K.TOL and np.abs(Sv - K.S_INIT).max() < K.TOL
    # first Adam step (HybridGradient.m:23-41): lr against the sign of the gradient, tabulated in kat_sampling.py
    from oracle.desc_pgd_literal import HybridGradient
    H = HybridGradient(K.ADAM_LR, 0.9, 0.999, 10)
    Sv, st = desc_pgd_literal(Ind, K.rotations(), 1, H, return_state=True, forced_lists=K.FORCED)
    assert np.abs(st["wijk"] - K.W1_ADAM).max() < K.TOL and np.abs(Sv - K.S1_ADAM).max() < K.TOL
    assert np.abs(H.m_t - K.ADAM_M1).max() < K.TOL and np.abs(H.v_t - K.ADAM_V1).max() < K.TOL and H.t == 1
    sd = K.structure_dict()
    am = np.zeros(10); av = np.zeros(10)
    res = oracle.pgd_run(sd, K.D, 1, step_kind=2, lr=K.ADAM_LR, beta1=0.9, beta2=0.999, decay_interval=10, adam_m=am, adam_v=av)
    assert np.abs(res["w"] - K.W1_ADAM).max() < K.TOL and np.abs(res["S_vec"] - K.S1_ADAM).max() < K.TOL
    assert np.abs(am - K.ADAM_M1).max() < K.TOL and np.abs(av - K.ADAM_V1).max() < K.TOL


# ------------------------------------------------------------------ invariants / cross-checks
@pytest.mark.parametrize("n,p,seed", [(30, 0.5, 1), (110, 0.6, 2)])
def test_literal_vs_sparse_c_oracle(oracle, n, p, seed):
    """Two independent restatements (dense literal NumPy / sparse C) agree: structure
    bit-exact, numbers to round-off -- in the no-sampling and in the sampling regime."""
    mo, nn, ii, jj, rij = make_problem("uniform", n=n, p=p, q=0.2, sigma=0.1, seed=seed)
    st = oracle.build_structure(nn, ii, jj, seed=7)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    res = oracle.pgd_run(st, S0, 25, lr=0.01)
    S_lit, state = desc_pgd_literal(mo.Ind, mo.RijMat, 25, ConstantStepSize(0.01), sampler=oracle.keyed_sampler(7),
                                    return_state=True)
    assert state["n_sample"] == st["n_sample"]
    assert np.array_equal(state["cum_ind"], st["cum_ind"])
    for a, b in (("IJK", "k"), ("Ind_jk", "e_jk"), ("Ind_ki", "e_ki"), ("IKJ", "ikj"), ("JKI", "jki")):
        assert np.array_equal(state[a] - 1, st[b]), a
    assert np.abs(state["S0_long"] - S0).max() < 1e-15
    assert np.abs(S_lit - res["S_vec"]).max() < 1e-13
    assert np.abs(state["wijk"] - res["w"]).max() < 1e-13
    assert np.allclose(state["obj_vals"], res["obj"], rtol=1e-13)
    if n >= 100:
        assert (st["ikj"] < 0).any()          # sampling regime: some mirrors are absent
    # invariants: segments on the simplex, S in [0,1]
    w = res["w"]
    sums = np.add.reduceat(w, st["cum_ind"][:-1])
    assert np.abs(sums - 1).max() < 1e-12 and (w >= 0).all()
    assert (res["S_vec"] >= 0).all() and (res["S_vec"] <= 1).all()


@pytest.mark.parametrize("kind", ["piecewise", "hybrid_adam", "hybrid_plain"])
def test_step_plugins_literal_vs_c(oracle, kind):
    mo, nn, ii, jj, rij = make_problem("uniform", n=40, p=0.5, q=0.2, sigma=0.1, seed=5)
    st = oracle.build_structure(nn, ii, jj, seed=1)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    if kind == "piecewise":
        G = PiecewiseStepSize(0.05, 4); G.t = 3
        res = oracle.pgd_run(st, S0, 15, step_kind=1, lr=0.05, decay_interval=4, t0=3)
    elif kind == "hybrid_adam":
        G = HybridGradient(0.002, 0.9, 0.999, 10)
        res = oracle.pgd_run(st, S0, 15, step_kind=2, lr=0.002, beta1=0.9, beta2=0.999, decay_interval=10)
    else:
        G = HybridGradient(0.0005, 0.9, 0.999, 5); G.stopAdam(); G.t = 2
        res = oracle.pgd_run(st, S0, 15, step_kind=2, lr=0.0005, decay_interval=5, hybrid_strategy=1, t0=2)
    S_lit = desc_pgd_literal(mo.Ind, mo.RijMat, 15, G, sampler=oracle.keyed_sampler(1))
    assert np.abs(S_lit - res["S_vec"]).max() < 1e-12
    assert G.t == (3 if kind == "piecewise" else 2 if kind == "hybrid_plain" else 0) + 15


def test_gradient_is_gradient_of_objective(oracle):
    """In the no-sampling regime grad_long (DESC_PGD.m:193) is the exact gradient of
    f(w) = sum_c w_c (s_jk + s_ki), s_e = sum_{c in e} w_c d_c: finite differences."""
    mo, nn, ii, jj, rij = make_problem("uniform", n=14, p=0.7, q=0.2, sigma=0.1, seed=8)
    st = oracle.build_structure(nn, ii, jj, seed=0)
    assert (st["ikj"] >= 0).all() and (st["jki"] >= 0).all()
    d = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    rng = np.random.default_rng(0)
    w = rng.random(st["m_cycle"])
    seg = np.repeat(np.arange(st["m_pos"]), np.diff(st["cum_ind"]))

    def f(w):
        S = np.ones(st["m"])
        S[st["pos_edge"]] = np.bincount(seg, w * d, minlength=st["m_pos"])
        return float(w @ (S[st["e_jk"]] + S[st["e_ki"]])), S

    f0, S = f(w)
    T1 = np.bincount(seg, w[st["ikj"]], minlength=st["m_pos"])[seg]
    T2 = np.bincount(seg, w[st["jki"]], minlength=st["m_pos"])[seg]
    grad = S[st["e_jk"]] + S[st["e_ki"]] + (T1 + T2) * d
    for c in rng.choice(st["m_cycle"], 12, replace=False):
        e = np.zeros_like(w); e[c] = 1e-6
        fd = (f(w + e)[0] - f(w - e)[0]) / 2e-6
        assert abs(fd - grad[c]) < 1e-7 * max(1, abs(grad[c]))


def test_statistical_recovery_like_the_demo(oracle):
    """Demo/compare_algorithms.m setting (n=100, p=0.5, q=0.2, sigma=0.1, 100 iterations):
    the estimate tracks the true corruption levels."""
    mo, nn, ii, jj, rij = make_problem("uniform", n=100, p=0.5, q=0.2, sigma=0.1, seed=0)
    st = oracle.build_structure(nn, ii, jj, seed=0)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    res = oracle.pgd_run(st, S0, 100, lr=0.01)
    err = np.mean(np.abs(res["S_vec"] - mo.ErrVec))
    assert err < 0.03
    corrupted = mo.corrupted
    assert res["S_vec"][corrupted].mean() > 3 * res["S_vec"][~corrupted].mean()


def test_long_double_yardstick_follows_the_double_oracle(oracle):
    """oracle_pgd_run_ld (the loop of DESC_PGD.m:148-261 carried in long double) agrees with the double-precision restatement to
    round-off at a contracting step (lr = 0.01), takes the same stop decision, and shows the amplification at lr = 1 that the GPU
    regression test of the round-2 fuzz case relies on (distance growing with the iteration count)."""
    mo, nn, ii, jj, rij = make_problem("uniform", n=70, p=0.6, q=0.2, sigma=0.1, seed=12)
    st = oracle.build_structure(nn, ii, jj, seed=4)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    a, b = oracle.pgd_run(st, S0, 60, lr=0.01), oracle.pgd_run_ld(st, S0, 60, lr=0.01)
    assert a["iters_run"] == b["iters_run"] == 60
    assert np.abs(a["S_vec"] - b["S_vec"]).max() < 1e-13 and np.abs(a["w"] - b["w"]).max() < 1e-13
    assert np.allclose(a["obj"], b["obj"], rtol=1e-13)
    kw = dict(step_kind=1, lr=1.0, decay_interval=7, patience=4, stop_tol=1e-2)
    a, b = oracle.pgd_run(st, S0, 400, **kw), oracle.pgd_run_ld(st, S0, 400, **kw)
    assert a["iters_run"] == b["iters_run"] < 400
    d = [np.abs(oracle.pgd_run(st, S0, t, lr=1.0)["w"] - oracle.pgd_run_ld(st, S0, t, lr=1.0)["w"]).max() for t in (2, 12, 24)]
    assert d[0] < 1e-14 and d[0] < d[1] < d[2]
