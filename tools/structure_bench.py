#!/usr/bin/env python3
"""Times the structure build (a-1..a-3) on host and device for a bench workload."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from desc_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
mo, nn, ii, jj, rij = bench.generate(name)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
out = {}
for where, tag in ((_lib.BUILD_DEVICE, "device"), (_lib.BUILD_DEVICE, "device_again"), (_lib.BUILD_HOST, "host")):
    t = time.perf_counter(); st = _lib.Structure.build(prob, 30, 0, where, 0); out[tag + "_s"] = time.perf_counter() - t
    if tag == "device": a = st.arrays()
    if tag == "host":
        b = st.arrays()
        out["equal"] = all(np.array_equal(a[k], b[k]) for k in ("codeg", "pos_edge", "cum_ind", "k", "e_jk", "e_ki", "ikj", "jki"))
    st.free()
out["sizes"] = dict(m=len(ii), m_cycle=int(a["m_cycle"]), n_sample=int(a["n_sample"]))
print(json.dumps(out))
