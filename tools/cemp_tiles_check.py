#!/usr/bin/env python3
"""CEMP rounds: the tile kernel (CSR-aligned S, LDS rows) against the plain wave-per-edge kernel on the same samples, for forced tile shapes."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 1:
    from desc_amd import _lib
    from tests.helpers import make_problem
    n, p = int(sys.argv[1]), float(sys.argv[2])
    mo, nn, ii, jj, rij = make_problem("uniform", n=n, p=p, seed=3)
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    S, ms = _lib.cemp_run(prob, [1, 2, 4, 8, 16, 32], 6, 50, seed=1)
    np.save(sys.argv[3], S)
else:
    for n, p in ((120, 0.5), (400, 0.5)):
        outs = {}
        for tag, env in (("plain", dict(DESC_DEBUG_CEMP_TILES="0")), ("tiles", {}), ("jb64", dict(DESC_DEBUG_CEMP_JB="64")), ("jb64_bi1", dict(DESC_DEBUG_CEMP_JB="64", DESC_DEBUG_CEMP_BI="1")),
                         ("jb32_bi3", dict(DESC_DEBUG_CEMP_JB="32", DESC_DEBUG_CEMP_BI="3")), ("jb50_bi7", dict(DESC_DEBUG_CEMP_JB="50", DESC_DEBUG_CEMP_BI="7"))):
            f = f"/tmp/cemp_{tag}.npy"
            subprocess.run([sys.executable, __file__, str(n), str(p), f], env=dict(os.environ, **env), check=True)
            outs[tag] = np.load(f)
        for tag in outs:
            d = np.abs(outs[tag] - outs["plain"])
            print(n, p, tag, "max diff vs plain %.3e, edges off by > 1e-9: %d of %d, first %s" % (d.max(), int((d > 1e-9).sum()), d.size, np.nonzero(d > 1e-9)[0][:8]))
