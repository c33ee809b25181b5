#!/usr/bin/env python3
"""Diagnostics: time the sweep kernel under ablation masks / grid sizes (one process,
one structure build).  Not part of the product or the bench."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from desc_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--ablate", default="0,1,2,4,8,16,3,15,31")
ap.add_argument("--grids", default="")
ap.add_argument("--variant", default="0")
ap.add_argument("--band", default="0")
ap.add_argument("--warm", type=int, default=40)
args = ap.parse_args()
os.environ["DESC_DEBUG_VARIANT"] = args.variant
os.environ["DESC_DEBUG_BAND"] = args.band
mo, nn, ii, jj, rij = bench.generate(args.workload)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
st = _lib.Structure.build(prob, 30, 0, _lib.BUILD_HOST, 0)
solver = _lib.Solver(prob, st, 0)
st.free()
B = 72.0 * solver.m_cycle + 12.0 * solver.m_pos
def run(tag):
    p = _lib.default_params(); p.iters = 2 * args.steps + args.warm + 10; p.patience = (1 << 31) - 1
    solver.reset(p); solver.iterate(args.warm); solver.sync()
    ms, mk = solver.iterate_timed(args.steps, per_kernel=True)
    print(json.dumps(dict(variant=args.variant, band=args.band, kernel=solver.kernel_name(), tag=tag, kernel_ms=mk, step_ms=ms / args.steps, GBs=B / mk / 1e6, frac=B / mk / 1e6 / 8000)), flush=True)
for ab in args.ablate.split(","):
    os.environ["DESC_DEBUG_ABLATE"] = ab
    run(f"ablate={ab}")
os.environ["DESC_DEBUG_ABLATE"] = "0"
for g in [x for x in args.grids.split(",") if x]:
    os.environ["DESC_DEBUG_GRID"] = g
    run(f"grid={g}")
