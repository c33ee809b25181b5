"""When does the slow edge-list upload (20 MB that normally take 0.4 ms) happen?  Eight desc_pgd_solve calls in a row, the lap of every call's
edge-list upload with the time since the process started; optionally a pause first (argv[1] seconds) and C2 instead of C4 (argv[2])."""
import os, sys, time, subprocess
T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from desc_amd import _lib
wl = sys.argv[2] if len(sys.argv) > 2 else "C4"
mo, nn, ii, jj, rij = bench.generate(wl)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
bench.warm_up(_lib)
pause = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
time.sleep(pause)
p = _lib.default_params(); p.iters = 20; p.lr = 0.01; p.patience = (1 << 31) - 1
os.environ["DESC_DEBUG_TIMING"] = "1"
for k in range(8):
    sys.stderr.write("CALL %d at %.2f s\n" % (k, time.perf_counter() - T0)); sys.stderr.flush()
    t = time.perf_counter(); _lib.solve(prob, p); dt = time.perf_counter() - t
    sys.stderr.write("CALL %d took %.1f ms\n" % (k, dt * 1e3)); sys.stderr.flush()
