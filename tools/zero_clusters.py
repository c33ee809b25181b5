#!/usr/bin/env python3
"""If a segment's cycles were ordered by their inconsistency S0 (inside each mirror class) instead of by k: how many aligned groups of 4 weights
(= 32-byte sectors of the weight array) would be all-zero after T iterations?  Offline analysis of one run (tools/zero_fraction.py gives the
plain zero fraction).  Usage: tools/zero_clusters.py [C2] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from desc_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
st = _lib.Structure.build(prob, 30, 0, _lib.BUILD_DEVICE, 0)
a = st.arrays()
solver = _lib.Solver(prob, st, 0)
d = solver.s0()
p = _lib.default_params(); p.iters = T; p.lr = 0.01; p.patience = (1 << 31) - 1
w = solver.run(p, want_w=True)["w"]
solver.destroy(); st.free()
cum = a["cum_ind"].astype(np.int64); cnt = np.diff(cum)
seg = np.repeat(np.arange(len(cnt)), cnt)
cls = (a["ikj"] >= 0).astype(np.int64) * 1 + (a["jki"] >= 0).astype(np.int64) * 2        # 1 (ik;j) only, 3 both, 2 (jk;i) only, 0 none
cls_rank = np.array([3, 0, 2, 1])[cls]                                                     # the layout's class order: [ikj only | both | jki only | none]
z = w == 0.0
def sectors(order_key):
    o = np.lexsort((order_key, cls_rank, seg))             # by segment, then class, then the key
    zz = z[o]; s2 = seg[o]
    pos = np.arange(len(zz)) - cum[s2]                      # position inside the segment (segments start on arbitrary sector offsets in memory: take the global index)
    g = np.arange(len(zz)) // 4
    allz = np.bincount(g, weights=zz.astype(np.float64)) == np.bincount(g)
    return allz.mean()
print(f"{wl} after {T} iterations: {z.mean()*100:.1f} % of the weights are zero")
print(f"  all-zero 32-byte sectors, cycles ordered by k inside a class (today): {sectors(a['k'])*100:.1f} %")
print(f"  all-zero 32-byte sectors, cycles ordered by S0 inside a class:        {sectors(d)*100:.1f} %")
print(f"  all-zero 32-byte sectors, cycles ordered by S0 across the segment:    {(lambda o: (np.bincount(np.arange(len(z))//4, weights=z[o].astype(float)) == np.bincount(np.arange(len(z))//4)).mean())(np.lexsort((d, seg)))*100:.1f} %")
