#!/usr/bin/env python3
"""How long the builder's edge-list upload takes over repeated desc_pgd_solve calls (DESC_DEBUG_TIMING laps), with the rotation upload
overlapped (default) and not (DESC_DEBUG_OVERLAP_UPLOAD=0)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import bench
    from desc_amd import _lib
    mo, nn, ii, jj, rij = bench.generate("C4")
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    bench.warm_up(_lib)
    p = _lib.default_params(); p.iters = 20; p.lr = 0.01; p.patience = (1 << 31) - 1
    _lib.solve(prob, p)
    os.environ["DESC_DEBUG_TIMING"] = "1"
    for rep in range(8):
        t = time.perf_counter(); out = _lib.solve(prob, p)
        print("solve ms %.2f structure %.2f" % ((time.perf_counter() - t) * 1e3, out["ms_structure"]), file=sys.stderr)
else:
    for ov in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DESC_DEBUG_OVERLAP_UPLOAD=ov), capture_output=True, text=True)
        print("DESC_DEBUG_OVERLAP_UPLOAD=" + ov)
        for ln in r.stderr.splitlines():
            if "upload Ind" in ln or ln.startswith("solve ms") or "solve structure " in ln:
                print("  ", ln)
