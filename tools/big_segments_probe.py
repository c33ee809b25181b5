import sys, time, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from desc_amd import _lib
from tests.helpers import make_problem
for n, p in ((1500, 0.5), (1200, 0.6)):
    mo, nn, ii, jj, rij = make_problem("uniform", n=n, p=p, q=0.3, seed=1)
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    t0 = time.perf_counter(); st = _lib.Structure.build(prob, 30, 0, _lib.BUILD_DEVICE, 0); t1 = time.perf_counter()
    solver = _lib.Solver(prob, st, 0); t2 = time.perf_counter()
    sz = st.sizes(); st.free()
    pp = _lib.default_params(); pp.iters = 60; pp.patience = 1 << 30
    solver.reset(pp); solver.iterate(5); solver.sync()
    ms, mk = solver.iterate_timed(20, per_kernel=True)
    B = 72.0 * solver.m_cycle + 12.0 * solver.m_pos
    print(json.dumps(dict(n=n, p=p, n_sample=sz["n_sample"], m_cycle=solver.m_cycle, kernel=solver.kernel_name(), ms_per_iter=ms / 20, kernel_ms=mk, frac=B / mk / 1e6 / 8000,
                          structure_ms=(t1 - t0) * 1e3, create_ms=(t2 - t1) * 1e3)), flush=True)
    solver.destroy()
