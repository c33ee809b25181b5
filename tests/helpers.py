"""Shared helpers for the parity tests."""
import numpy as np

from desc_amd import _lib
from desc_amd.algorithms import marshal_edges
from desc_amd.models import Nonuniform_Topology, Uniform_Topology


def make_problem(kind="uniform", n=60, p=0.5, q=0.2, sigma=0.1, seed=0, **kw):
    if kind == "uniform":
        mo = Uniform_Topology(n, p, q, sigma, kw.get("model", "uniform"), seed=seed)
    else:
        mo = Nonuniform_Topology(n, p, kw.get("p_node_crpt", 0.5), kw.get("p_edge_crpt", 0.5),
                                 kw.get("sigma_in", 0.1), kw.get("sigma_out", 0.1),
                                 kw.get("crpt_type", "self-consistent"), seed=seed)
    nn, ii, jj, rij, perm = marshal_edges(mo.Ind, mo.RijMat)
    assert perm is None
    return mo, nn, ii, jj, rij


def oracle_reference(O, nn, ii, jj, rij, seed, iters, **step):
    st = O.build_structure(nn, ii, jj, seed=seed)
    S0 = O.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    res = O.pgd_run(st, S0, iters, **step)
    return st, S0, res


def c_params(iters, step_kind=0, lr=0.01, beta1=0.9, beta2=0.999, decay_interval=25, hybrid_strategy=0,
             t0=0, patience=30, stop_tol=1e-5, seed=0, check_every=0):
    p = _lib.default_params()
    p.iters = iters; p.step_kind = step_kind; p.lr = lr; p.beta1 = beta1; p.beta2 = beta2
    p.decay_interval = decay_interval; p.hybrid_strategy = hybrid_strategy; p.t0 = t0
    p.patience = patience; p.stop_tol = stop_tol; p.seed = seed; p.check_every = check_every
    return p


STRUCT_KEYS = ("pos_edge", "cum_ind", "k", "e_jk", "e_ki", "ikj", "jki")


def assert_structure_equal(a, b):
    assert a["m_pos"] == b["m_pos"] and a["m_cycle"] == b["m_cycle"] and a["n_sample"] == b["n_sample"]
    for key in STRUCT_KEYS:
        assert np.array_equal(a[key], b[key]), key
