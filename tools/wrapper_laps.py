"""Where the wall time of the Python wrapper DESC_PGD() goes at a BASELINE workload, next to the bare desc_pgd_solve call (alternating)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import generate                            # noqa: E402
from desc_amd import DESC, DESC_PGD, ConstantStepSize, _lib    # noqa: E402
from desc_amd.algorithms import marshal_edges            # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="C4"); ap.add_argument("--laps", type=int, default=4)
a = ap.parse_args()
mo = generate(a.workload)[0]
p = _lib.default_params(); p.iters = 100; p.lr = 0.01; p.verbose = 0
for lap in range(a.laps):
    t0 = time.perf_counter(); n, ii, jj, rij, perm = marshal_edges(mo.Ind, mo.RijMat); t1 = time.perf_counter()
    prob = _lib.ProblemArrays(n, ii, jj, rij); t2 = time.perf_counter()
    out = _lib.solve(prob, p); t3 = time.perf_counter()
    S = DESC_PGD(mo.Ind, mo.RijMat, dict(iters=100, Gradient=ConstantStepSize(0.01), verbose=False)); t4 = time.perf_counter()
    assert np.array_equal(S, out["S_vec"])
    print("lap %d: marshal %.1f ms, ProblemArrays %.1f ms, solve() %.1f ms (library's own clock %.1f), DESC_PGD() %.1f ms   [Ind %s %s, RijMat %s]" % (
        lap, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, out["ms_total"], (t4 - t3) * 1e3, mo.Ind.dtype, "F" if mo.Ind.flags.f_contiguous else "C",
        "F" if mo.RijMat.flags.f_contiguous else "C"), flush=True)
Rc = np.ascontiguousarray(mo.RijMat)
for lap in range(2):
    t0 = time.perf_counter(); S2 = DESC_PGD(mo.Ind, Rc, dict(iters=100, Gradient=ConstantStepSize(0.01), verbose=False)); t1 = time.perf_counter()
    assert np.array_equal(S2, S)
    print("C-ordered RijMat (NumPy's own order): DESC_PGD() %.1f ms" % ((t1 - t0) * 1e3), flush=True)
for lap in range(4):
    t0 = time.perf_counter(); Re, Ri, Sv = DESC(mo.Ind, mo.RijMat, dict(iters=100, Gradient=ConstantStepSize(0.01), verbose=False)); t1 = time.perf_counter()
    assert np.array_equal(Sv, S)
    print("DESC() = DESC_PGD -> GCW -> refinement on one device-resident problem: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
