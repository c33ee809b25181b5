import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, bench
from desc_amd import _lib
bench.warm_up(_lib)
for name in sys.argv[1:]:
    mo, nn, ii, jj, rij = bench.generate(name)
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    for rep in range(2):
        p = _lib.default_params(); p.iters = 100; p.lr = 0.01; p.patience = (1 << 31) - 1
        t0 = time.perf_counter(); out = _lib.solve(prob, p); dt = time.perf_counter() - t0
        print(f"{name} rep {rep}: solve {dt*1e3:.1f} ms  structure {out['ms_structure']:.1f} upload {out['ms_upload']:.1f} layout {out['ms_cycle_d']:.1f} pgd {out['ms_pgd']:.1f} total_in_lib {out['ms_total']:.1f}", file=sys.stderr, flush=True)
