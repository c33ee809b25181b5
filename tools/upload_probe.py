import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from desc_amd import _lib
bench.warm_up(_lib)
mo, nn, ii, jj, rij = bench.generate(sys.argv[1] if len(sys.argv) > 1 else "C4")
prob = _lib.ProblemArrays(nn, ii, jj, rij)
for rep in range(4):
    t0 = time.perf_counter(); dp = _lib.DeviceProblem(prob, 0); dt = time.perf_counter() - t0; dp.free()
    print(f"upload rep {rep}: {dt*1e3:.1f} ms (pin={os.environ.get('DESC_UPLOAD_PIN','1')})", file=sys.stderr, flush=True)
