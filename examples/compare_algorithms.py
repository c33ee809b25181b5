#!/usr/bin/env python3
"""The reference's demo (Demo/compare_algorithms.m:9-99) through the Python mirror of its MATLAB calls, on the GPU.

Same model (Uniform_Topology n, p, q, sigma, 'uniform'; the demo's defaults n = 100 ... BASELINE configs[0] uses n = 200), same
parameter structs (:25-46), same calls in the same order, rotations aligned with Rotation_Alignment (:75-82) and tabulated (:85-99).
Rows of the reference's table that belong to algorithms outside this library's scope (MPLS, CEMP+MST, IRLS-GM, IRLS-L0.5: SURVEY.md 2)
are left out; CEMP+GCW is the composition CEMP() -> GCW() of the two entry points the library has.

    python examples/compare_algorithms.py [--n 200] [--p 0.5] [--q 0.2] [--sigma 0.1] [--seed 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from desc_amd import CEMP, DESC, GCW, ConstantStepSize, Rotation_Alignment, Spectral, Uniform_Topology  # noqa: E402


def run(n=200, p=0.5, q=0.2, sigma=0.1, seed=0, verbose=True):
    model_out = Uniform_Topology(n, p, q, sigma, "uniform", seed=seed)                  # :13
    Ind, RijMat, ErrVec, R_orig = model_out.Ind, model_out.RijMat, model_out.ErrVec, model_out.R_orig   # :20-23
    CEMP_parameters = dict(max_iter=6, reweighting=[2.0 ** k for k in range(6)], nsample=50, gcw_beta=5)   # :26-29
    lr = 0.01                                                                            # :38-46
    DESC_parameters = dict(iters=100, learning_rate=lr, make_plots=False, Gradient=ConstantStepSize(lr), R_orig=R_orig, ErrVec=ErrVec,
                           verbose=verbose)
    R_SP = Spectral(Ind, RijMat)                                                         # :63
    SVec = CEMP(Ind, RijMat, CEMP_parameters)                                            # :66 (CEMP_GCW.m = CEMP.m + GCW.m)
    R_CEMP_GCW = GCW(Ind, None, RijMat, SVec)
    R_DESC, R_DESC_init, S_vec = DESC(Ind, RijMat, DESC_parameters)                      # :72
    rows = []
    for name, R in (("Spectral", R_SP), ("CEMP+GCW", R_CEMP_GCW), ("DESC_init", R_DESC_init), ("DESC", R_DESC)):
        _, _, mean_error, median_error = Rotation_Alignment(R, R_orig)                   # :75-82
        rows.append((name, float(mean_error), float(median_error)))
    return rows, dict(mean_abs_err_cemp=float(abs(SVec - ErrVec).mean()), mean_abs_err_desc=float(abs(S_vec - ErrVec).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200); ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--q", type=float, default=0.2); ap.add_argument("--sigma", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=0); ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    rows, extra = run(a.n, a.p, a.q, a.sigma, a.seed, verbose=not a.quiet)
    print("\nResults =\n")                                                              # :85-99
    print("    %-12s %-12s %-12s" % ("Algorithms", "MeanError", "MedianError"))
    for name, me, md in rows:
        print("    %-12s %-12.4f %-12.4f" % ('"' + name + '"', me, md))
    print("\n(degrees; corruption levels: mean |SVec - ErrVec| CEMP %.4f, DESC %.4f)" % (extra["mean_abs_err_cemp"], extra["mean_abs_err_desc"]))


if __name__ == "__main__":
    main()
