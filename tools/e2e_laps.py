#!/usr/bin/env python3
"""DESC_DEBUG_TIMING laps of one warm desc_pgd_solve call (host arrays in -> S_vec out, 100 iterations)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from desc_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
bench.warm_up(_lib)
p = _lib.default_params(); p.iters = 100; p.lr = 0.01; p.patience = (1 << 31) - 1
_lib.solve(prob, p)
os.environ["DESC_DEBUG_TIMING"] = os.environ.get("LAPS_LEVEL", "1")
for rep in range(2):
    t = time.perf_counter(); out = _lib.solve(prob, p)
    print("solve ms %.2f (structure %.2f upload %.2f cycle_d %.2f pgd %.2f total %.2f)" % ((time.perf_counter() - t) * 1e3, out["ms_structure"], out["ms_upload"], out["ms_cycle_d"], out["ms_pgd"], out["ms_total"]), file=sys.stderr)
