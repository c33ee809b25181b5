#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.

The reference (ColeWyeth/DESC) ships no fixtures and cannot be executed in the build
environment (MATLAB only), so these vectors are produced by the ORACLE's literal NumPy
restatement of Algorithms/DESC_PGD.m (oracle/desc_pgd_literal.py) on seeded synthetic
inputs -- inputs, the deterministic cycle-sampling seed, the resulting structure and the
expected outputs.  They pin the sparse C oracle, the host structure builder and the HIP
path to one another; they do NOT pin any of them to a MATLAB run (parity unpinned).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from desc_amd.models import Nonuniform_Topology, Uniform_Topology  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle.desc_pgd_literal import (ConstantStepSize, HybridGradient, PiecewiseStepSize,  # noqa: E402
                                     desc_pgd_literal)

CASES = {
    # name: (model, sampling seed, iters, step spec)
    "uniform_n24_nosampling": (lambda: Uniform_Topology(24, 0.6, 0.2, 0.1, "uniform", seed=101), 5, 40, ("const", 0.01)),
    "uniform_n90_sampling": (lambda: Uniform_Topology(90, 0.6, 0.25, 0.1, "uniform", seed=102), 6, 30, ("const", 0.01)),
    "selfconsistent_n60_lr1": (lambda: Uniform_Topology(60, 0.5, 0.3, 0.05, "self-consistent", seed=103), 7, 25, ("const", 1.0)),
    "nonuniform_adv_n70_piecewise": (lambda: Nonuniform_Topology(70, 0.5, 0.4, 0.5, 0.1, 0.1, "adv", seed=104), 8, 20,
                                     ("piecewise", 0.05, 6)),
    "uniform_n50_adam": (lambda: Uniform_Topology(50, 0.5, 0.2, 0.1, "uniform", seed=105), 9, 20, ("hybrid", 0.002, 0.9, 0.999, 10)),
}


def make_gradient(spec):
    if spec[0] == "const":
        return ConstantStepSize(spec[1])
    if spec[0] == "piecewise":
        return PiecewiseStepSize(spec[1], spec[2])
    return HybridGradient(spec[1], spec[2], spec[3], spec[4])


def main():
    for name, (gen, seed, iters, spec) in CASES.items():
        mo = gen()
        S, st = desc_pgd_literal(mo.Ind, mo.RijMat, iters, make_gradient(spec), sampler=O.keyed_sampler(seed),
                                 return_state=True)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            Ind=mo.Ind.astype(np.int32), RijMat=np.ascontiguousarray(mo.RijMat), ErrVec=mo.ErrVec,
            sampling_seed=np.int64(seed), iters=np.int64(iters), step=np.array(spec[1:], dtype=np.float64),
            step_kind=np.array({"const": 0, "piecewise": 1, "hybrid": 2}[spec[0]]),
            n_sample=np.int64(st["n_sample"]), pos_edge=(st["CoDeg_pos_ind"] - 1).astype(np.int32),
            cum_ind=st["cum_ind"].astype(np.int64), k=(st["IJK"] - 1).astype(np.int32),
            e_jk=(st["Ind_jk"] - 1).astype(np.int32), e_ki=(st["Ind_ki"] - 1).astype(np.int32),
            ikj=(st["IKJ"] - 1).astype(np.int32), jki=(st["JKI"] - 1).astype(np.int32),
            S0_long=st["S0_long"], S_vec=S, wijk=st["wijk"], obj_vals=st["obj_vals"], avg_changes=st["avg_changes"],
            iters_run=np.int64(st["iters_run"]))
        print(name, "m =", mo.Ind.shape[0], "m_cycle =", st["m_cycle"], "iters_run =", st["iters_run"])


if __name__ == "__main__":
    main()
