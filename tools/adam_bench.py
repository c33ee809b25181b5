#!/usr/bin/env python3
"""Iterations/s of the Adam plugin (HybridGradient strategy 0) on a BASELINE workload: tools/adam_bench.py C4 [steps]."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from desc_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
p = _lib.default_params(); p.iters = K + 5; p.step_kind = 2; p.lr = 0.01; p.hybrid_strategy = 0
st = _lib.Structure.build(prob, p.n_sample_min, p.seed, _lib.BUILD_DEVICE, 0)
s = _lib.Solver(prob, st, 0); st.free()
s.reset(p); s.iterate(5); s.sync()
ms, mk = s.iterate_timed(K, per_kernel=True)
out = s.download()
print(json.dumps({"workload": wl, "lib": os.path.basename(os.environ.get("DESC_AMD_LIB", "libdesc_amd.so")), "kernel": s.kernel_name(), "adam_ms_per_iteration": ms / K,
                  "main_kernel_ms": mk, "m_cycle": s.m_cycle, "GBps_at_104B_per_cycle": 104.0 * s.m_cycle / (ms / K * 1e-3) / 1e9}))
s.destroy()
