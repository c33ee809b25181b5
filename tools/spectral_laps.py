#!/usr/bin/env python3
"""Where Spectral / GCW spend their time on a device-resident problem (DESC_DEBUG_TIMING laps) + the SpMM variants."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["DESC_DEBUG_TIMING"] = "1"
import numpy as np
import bench
from desc_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
bench.warm_up(_lib)
dp = _lib.DeviceProblem(prob, 0)
S = np.clip(mo.ErrVec + 0.01, 0, 1)
for rep in range(2):
    t = time.perf_counter(); _lib.spectral_run(dp); print("spectral ms", (time.perf_counter() - t) * 1e3, file=sys.stderr)
    t = time.perf_counter(); _lib.gcw_run(dp, S); print("gcw ms", (time.perf_counter() - t) * 1e3, file=sys.stderr)
print(json.dumps(_lib.spmm_variants(dp, reps=30)))
dp.free()
