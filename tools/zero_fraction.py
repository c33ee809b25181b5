#!/usr/bin/env python3
"""How many cycle weights are exactly zero after t iterations (the simplex projection clips), and how many of those were zero one and two iterations earlier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from desc_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
st = _lib.Structure.build(prob, 30, 0, _lib.BUILD_DEVICE, 0)
prev = []
for T in (1, 2, 3, 10, 11, 12, 50, 51, 52, 100):
    solver = _lib.Solver(prob, st, 0)
    p = _lib.default_params(); p.iters = T; p.lr = 0.01; p.patience = (1 << 31) - 1
    out = solver.run(p, want_w=True)
    w = out["w"]; z = w == 0.0
    msg = f"{wl} after {T:3d} iterations: {z.mean()*100:5.1f} % of the weights are exactly 0"
    if prev and prev[-1][0] == T - 1:
        msg += f"; zero now and one iteration before: {(z & prev[-1][1]).mean()*100:5.1f} %"
        if len(prev) > 1 and prev[-2][0] == T - 2:
            msg += f"; and two before: {(z & prev[-1][1] & prev[-2][1]).mean()*100:5.1f} %; bitwise unchanged vs two before (any value): {(w == prev[-2][2]).mean()*100:5.1f} %"
    print(msg, flush=True)
    prev.append((T, z, w)); prev = prev[-2:]
    solver.destroy()
st.free()
