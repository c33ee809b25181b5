#!/usr/bin/env python3
"""Per-workgroup start / end times of the band sweep of ONE rank of a world-N run (DESC_DEBUG_WGCLOCK), one exchange part: where the per-rank sweep
(172 us for an eighth of C4 against 126 = one GPU / 8) loses its time.  usage: wg_clock_shard.py [workload] [world] [rank]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["DESC_DEBUG_WGCLOCK"] = "1"
os.environ.setdefault("DESC_SHARD_PARTS", "1")
import ctypes as C
import numpy as np
import torch
import bench
from desc_amd import _lib
from desc_amd.sharded import HipShard
wl = sys.argv[1] if len(sys.argv) > 1 else "C4"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
bench.warm_up(_lib)
st = _lib.Structure.build(prob, 30, 0, _lib.BUILD_DEVICE, 0)
sh = HipShard(prob, st, 0, rank, world); st.free()
p = _lib.default_params(); p.iters = 40; p.lr = 0.01; p.patience = (1 << 31) - 1
with sh.stream_ctx():
    sh.reset(p); sh.finish(1); sh.finish(2)
    for _ in range(6):
        sh.colsum(); sh.sweep(); sh.finish(0)
sh.sync(); torch.cuda.synchronize()
buf = np.zeros(3 * 1024, dtype=np.uint64)
n = _lib.load().desc_debug_wg_clock(sh.solver.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 1024)
t = buf[:2 * n].reshape(n, 2).astype(np.float64) / 100.0
t0 = t[:, 0].min()
start, end = t[:, 0] - t0, t[:, 1] - t0
dur = end - start
print(f"{wl} rank {rank} of {world}: {n} workgroups; start spread {start.max():.1f} us; end: min {end.min():.1f} mean {end.mean():.1f} max {end.max():.1f} us; "
      f"duration: min {dur.min():.1f} p10 {np.percentile(dur, 10):.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
plan = np.zeros(4 * 1024, dtype=np.int64)
g = _lib.load().desc_debug_wg_plan(sh.solver.handle, plan.ctypes.data_as(C.POINTER(C.c_int64)), 1024)
if g == n:
    P = plan[:4 * n].reshape(n, 4).astype(np.float64)
    print("plan per workgroup: cycles %.0f..%.0f (mean %.0f), segments %.0f..%.0f, pieces %.0f..%.0f, row entries %.0f..%.0f" % (
        P[:, 0].min(), P[:, 0].max(), P[:, 0].mean(), P[:, 1].min(), P[:, 1].max(), P[:, 2].min(), P[:, 2].max(), P[:, 3].min(), P[:, 3].max()))
    X = np.column_stack([P[:, 0], P[:, 2], np.ones(n)])
    coef = np.linalg.lstsq(X, dur, rcond=None)[0]
    print("least squares  duration_us = %.3e * cycles + %.2f * pieces + %.1f;  residual std %.1f us" % (coef[0], coef[1], coef[2], float(np.std(dur - X @ coef))))
    print("  one GPU streams %.3e us per cycle per workgroup (125 M cycles, 256 workgroups, 985 us)" % (985.0 / (125e6 / 256)))
    for name, col in (("cycles", 0), ("pieces", 2), ("row entries", 3)):
        print("  corr(duration, %s) = %.2f" % (name, float(np.corrcoef(dur, P[:, col])[0, 1])))
print("ten slowest workgroups (id, start, end us):", [(int(i), round(float(start[i]), 1), round(float(end[i]), 1)) for i in np.argsort(-end)[:10]])
sh.destroy()
