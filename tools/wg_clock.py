#!/usr/bin/env python3
"""Per-workgroup start / end times of one band sweep (DESC_DEBUG_WGCLOCK): how evenly the piece scheduler loads the 256 workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["DESC_DEBUG_WGCLOCK"] = "1"
import ctypes as C
import numpy as np
import bench
from desc_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
bench.warm_up(_lib)
st = _lib.Structure.build(prob, 30, 0, _lib.BUILD_DEVICE, 0)
solver = _lib.Solver(prob, st, 0); st.free()
p = _lib.default_params(); p.iters = 40; p.lr = 0.01; p.patience = (1 << 31) - 1
solver.reset(p); solver.iterate(20); solver.sync()
buf = np.zeros(2 * 1024, dtype=np.uint64)
n = _lib.load().desc_debug_wg_clock(solver.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 1024)
t = buf[:2 * n].reshape(n, 2).astype(np.float64) / 100.0      # microseconds (100 MHz)
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, 1] - t0
dur = end - start
print(f"{wl}: {n} workgroups; start spread {start.max():.1f} us; end: min {end.min():.1f} mean {end.mean():.1f} max {end.max():.1f} us; "
      f"duration: min {dur.min():.1f} p10 {np.percentile(dur, 10):.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
print("ten slowest workgroups (id, end us):", [(int(i), round(float(end[i]), 1)) for i in np.argsort(-end)[:10]])
print("ten fastest workgroups (id, end us):", [(int(i), round(float(end[i]), 1)) for i in np.argsort(end)[:10]])
solver.destroy()
