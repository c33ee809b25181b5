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
cyc = buf[2 * n:3 * n].astype(np.float64)
if cyc.max() > 0:
    mhz = cyc / np.maximum(dur, 1e-9)
    print(f"shader clock over each workgroup's run (cycles / duration): mean {mhz.mean():.0f} MHz, min {mhz.min():.0f}, max {mhz.max():.0f}; by blockIdx % 8: {[round(float(mhz[x::8].mean())) for x in range(8)]}")
print("mean duration by blockIdx % 8 (the XCD a workgroup lands on, round-robin dispatch):", [round(float(dur[x::8].mean()), 1) for x in range(8)])
print("ten slowest workgroups (id, end us):", [(int(i), round(float(end[i]), 1)) for i in np.argsort(-end)[:10]])
print("ten fastest workgroups (id, end us):", [(int(i), round(float(end[i]), 1)) for i in np.argsort(end)[:10]])
plan = np.zeros(4 * 1024, dtype=np.int64)
g = _lib.load().desc_debug_wg_plan(solver.handle, plan.ctypes.data_as(C.POINTER(C.c_int64)), 1024)
if g == n:
    P = plan[:4 * n].reshape(n, 4).astype(np.float64)
    X = np.column_stack([P[:, 0], P[:, 1], P[:, 2], P[:, 3]])
    coef, res, rk, sv = np.linalg.lstsq(X, dur, rcond=None)
    fit = X @ coef
    print("plan per workgroup: cycles %.0f..%.0f, segments %.0f..%.0f, pieces %.0f..%.0f, row entries %.0f..%.0f" % (
        P[:, 0].min(), P[:, 0].max(), P[:, 1].min(), P[:, 1].max(), P[:, 2].min(), P[:, 2].max(), P[:, 3].min(), P[:, 3].max()))
    print("least squares  duration_us = %.3e * cycles + %.3e * segments + %.3e * pieces + %.3e * row_entries;  residual std %.1f us (duration std %.1f us)" % (
        coef[0], coef[1], coef[2], coef[3], float(np.std(dur - fit)), float(np.std(dur))))
    for name, col in (("cycles", 0), ("segments", 1), ("pieces", 2), ("row entries", 3)):
        print("  corr(duration, %s) = %.2f" % (name, float(np.corrcoef(dur, P[:, col])[0, 1])))
solver.destroy()
