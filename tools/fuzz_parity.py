#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random graphs / sampling budgets / step plugins, HIP path
(device- or host-built structure; band sweep, k_sweep_node and gather layouts) against the CPU oracle.  Test infrastructure,
like tests/: imports oracle/.  Prints one line per failure and a summary; exit code 1 on any failure."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libdesc_amd.so, see desc_amd/_lib.py)
from desc_amd import _lib as lib
from oracle import oracle as O
from tests.helpers import make_problem, c_params, STRUCT_KEYS
from tests.test_gpu_sharded import _emulate as emulate

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=240)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
t_end = time.time() + args.seconds
n_cases = n_fail = n_sharded = 0
worst = 0.0                 # cases under the 1e-10 bound (constant / piecewise steps)
worst_adam = 0.0            # Adam cases (bound 1e-9 / 1e-8: the update divides by sqrt(v) + 1e-8)
worst_tag = worst_adam_tag = None
t_note = time.time()
while time.time() < t_end:
    if time.time() - t_note > 60:          # a progress line per minute (a silent GPU run is taken to be hung)
        print("... %d cases, %d failures so far" % (n_cases, n_fail), flush=True); t_note = time.time()
    kind = "uniform" if rng.random() < 0.7 else "nonuniform"
    n = int(rng.choice([5, 8, 13, 21, 34, 55, 89, 144, 233, 300]))
    p = float(rng.choice([0.08, 0.15, 0.3, 0.5, 0.7, 0.95]))
    nmin = int(rng.choice([1, 2, 5, 16, 17, 30, 33, 64, 65, 100, 129, 200, 256, 300]))
    sk = int(rng.choice([0, 0, 1, 2]))
    iters = int(rng.choice([1, 2, 7, 40]))
    seed = int(rng.integers(0, 1 << 30))
    lr = float(rng.choice([0.01, 0.1, 1.0]))
    variant = str(rng.choice(["0", "1", "2", "3", "3", "3"]))        # DESC_DEBUG_VARIANT: 0 library's choice, 1 gather layout, 2 k_sweep_node, 3 band sweep (forced on small graphs)
    where = lib.BUILD_DEVICE if rng.random() < 0.8 else lib.BUILD_HOST
    # a fifth of the node-layout cases run SHARDED: world 2 / 3 / 8 emulated on the one card (tests/test_gpu_sharded.py::_emulate), the band
    # sweep's XT instances on a graph cut into many bands (DESC_DEBUG_ROW_CAP) when the band sweep is forced
    world = int(rng.choice([2, 3, 8])) if (variant in ("0", "2", "3") and nmin <= 256 and rng.random() < 0.2) else 1
    row_cap = int(rng.choice([0, 64, 300, 1500])) if world > 1 else 0
    tag = dict(kind=kind, n=n, p=p, nmin=nmin, sk=sk, iters=iters, seed=seed, lr=lr, variant=variant, where=where, world=world, row_cap=row_cap)
    try:
        mo, nn, ii, jj, rij = make_problem(kind, n=n, p=p, seed=seed % 1000)
    except Exception as e:                      # generator refuses degenerate graphs
        continue
    if ii.shape[0] == 0:
        continue
    n_cases += 1
    try:
        st = O.build_structure(nn, ii, jj, seed=seed, n_sample_min=nmin)
        if st["m_pos"] and np.diff(st["cum_ind"]).max() > 64 and variant == "0":
            pass                                 # node layout not applicable: the library falls back to gather itself
        S0 = O.cycle_d(ii, jj, rij.reshape(-1, 9), st)
        step = dict(step_kind=sk, lr=lr, hybrid_strategy=0)
        ref = O.pgd_run(st, S0, iters, **step)
        os.environ["DESC_DEBUG_VARIANT"] = variant
        pr = c_params(iters, seed=seed, **step)
        sharded = world > 1 and st["m_pos"] > 0 and np.diff(st["cum_ind"]).max() <= 256
        if sharded:
            if row_cap:
                os.environ["DESC_DEBUG_ROW_CAP"] = str(row_cap)
            outs, _ = emulate(lib, nn, ii, jj, rij, pr, world, where=where, nmin=nmin)
            n_sharded += 1
            for o in outs[1:]:
                assert np.array_equal(o["S_vec"], outs[0]["S_vec"]) and np.array_equal(o["obj"], outs[0]["obj"]), "ranks disagree"
            out = outs[0]
            out["w"] = ref["w"]                   # per-cycle outputs are not gathered across ranks
        else:
            prob = lib.ProblemArrays(nn, ii, jj, rij)
            dst = lib.Structure.build(prob, nmin, seed, where, 0)
            solver = lib.Solver(prob, dst, 0)
            out = solver.run(pr, want_w=True)
            solver.destroy()
            a = dst.arrays(); dst.free()
            for key in STRUCT_KEYS:
                assert np.array_equal(a[key], st[key]), "structure " + key
        assert out["iters_run"] == ref["iters_run"], "iters_run %d vs %d" % (out["iters_run"], ref["iters_run"])
        # Adam divides by sqrt(v) + 1e-8: round-off is amplified.  Everything else: 1e-10.  lr = 1 amplifies round-off by
        # ~1.3x per sweep (tests/test_gpu_parity.py::test_fuzz_case_945063979_is_roundoff): a case beyond 1e-10 there is judged against
        # the long-double run of the same loop -- it passes if the HIP result is within 4x of the double oracle's own distance to it
        tol = (1e-8 if lr >= 0.1 else 1e-9) if sk == 2 else 1e-10
        e1 = float(np.abs(out["S_vec"] - ref["S_vec"]).max()) if nn else 0.0
        e2 = float(np.abs(out["w"] - ref["w"]).max()) if st["m_cycle"] else 0.0
        if max(e1, e2) > tol and sk != 2 and lr >= 1.0:
            ld = O.pgd_run_ld(st, S0, iters, **{k: v for k, v in step.items() if k != "hybrid_strategy"})
            yard = max(float(np.abs(ref["S_vec"] - ld["S_vec"]).max()), float(np.abs(ref["w"] - ld["w"]).max()))
            eh = max(float(np.abs(out["S_vec"] - ld["S_vec"]).max()), float(np.abs(out["w"] - ld["w"]).max()))
            print("yardstick", tag, "hip-vs-oracle %.3g, oracle-vs-long-double %.3g, hip-vs-long-double %.3g" % (max(e1, e2), yard, eh), flush=True)
            # (the cap on the yardstick itself was 1e-9 until a case with 256-cycle segments showed the double oracle 2.9e-9 from the long-double
            #  run after 40 sweeps at lr = 1, HIP 9.8e-10 from the oracle: tests/test_gpu_parity.py::test_fuzz_case_621930630_is_roundoff)
            assert eh <= 4 * yard and yard <= 1e-8, "values %g %g beyond 4x the round-off yardstick %g" % (e1, e2, yard)
            e1 = e2 = 0.0
        if sk == 2:
            if max(e1, e2) > worst_adam: worst_adam, worst_adam_tag = max(e1, e2), tag
        elif max(e1, e2) > worst: worst, worst_tag = max(e1, e2), tag
        assert e1 <= tol and e2 <= tol, "values %g %g" % (e1, e2)
        assert np.allclose(out["obj"], ref["obj"], rtol=1e-11, atol=1e-9), "objective trace"
    except Exception as e:
        n_fail += 1
        print("FAIL", tag, repr(e)[:300], flush=True)
    finally:
        os.environ.pop("DESC_DEBUG_VARIANT", None)
        os.environ.pop("DESC_DEBUG_ROW_CAP", None)
print("cases %d (of which %d sharded runs) failures %d" % (n_cases, n_sharded, n_fail), flush=True)
print("worst |diff| under the 1e-10 bound (constant / piecewise step; lr = 1 cases beyond it are judged by the yardstick lines above): %.3g  %s" % (worst, worst_tag), flush=True)
print("worst |diff| of the Adam cases (bound 1e-9, 1e-8 at lr >= 0.1): %.3g  %s" % (worst_adam, worst_adam_tag), flush=True)
sys.exit(1 if n_fail else 0)
