import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
from desc_amd import _lib
bench.warm_up(_lib)
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
mo, nn, ii, jj, rij = bench.generate(name)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
dp = _lib.DeviceProblem(prob, 0)
beta = [1, 2, 4, 8, 16, 32]
for rep in range(3):
    t0 = time.perf_counter(); S, ms = _lib.cemp_run(dp, beta, 6, 50); dt = time.perf_counter() - t0
    print(f"{name} cemp rep {rep}: {dt*1e3:.1f} ms (lib {ms:.1f}) err {np.mean(np.abs(S-mo.ErrVec)):.4f}", file=sys.stderr, flush=True)
dp.free()
