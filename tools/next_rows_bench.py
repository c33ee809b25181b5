#!/usr/bin/env python3
"""Times the "next" rows on one GPU for a BASELINE workload: Spectral, GCW, CEMP and the
DESC() refinement tail, each through the C ABI, with accuracy against the synthetic ground truth.
Prints one JSON object.  Not the bench line (that is bench.py)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from desc_amd import _lib
from desc_amd.algorithms import Rotation_Alignment

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
args = ap.parse_args()
mo, nn, ii, jj, rij = bench.generate(args.workload)
prob = _lib.ProblemArrays(nn, ii, jj, rij)
m = prob.m
out = {"workload": bench.describe(args.workload), "n": nn, "m": m}

def rot_err(R):
    _, _, mean_err, med_err = Rotation_Alignment(R, mo.R_orig)
    return {"mean_deg": float(mean_err), "median_deg": float(med_err)}

# warm-up on a small graph, outside the timers: HIP context, code objects, the runtime's staging buffers
wm, wn, wi, wj, wr = bench.generate("C1")
wprob = _lib.ProblemArrays(wn, wi, wj, wr)
wp = _lib.default_params(); wp.iters = 2
wS = _lib.solve(wprob, wp)["S_vec"]
wR, _ = _lib.spectral_run(wprob, weights=1.0 / (wS ** 1.5 + 1e-8), normalize_rows=True)
_lib.spectral_run(wprob); _lib.refine_run(wprob, wS, wR); _lib.cemp_run(wprob, [1, 2], 2, 20)
t0 = time.perf_counter()
R, info = _lib.spectral_run(prob)
dt = time.perf_counter() - t0
# one block-SpMM step streams the 2m blocks (72 B) + block indices (4 B) once and gathers 2m x 144-B operand rows
spmm_bytes = 2 * m * (72 + 4 + 144) + 2 * nn * 144
out["spectral"] = dict(ms_wall=dt * 1e3, ms_lib=info["ms_total"], outer_iters=info["iters"], spmm_products=info["products"], residual=info["residual"],
                       converged=info["converged"], spmm_bytes_per_product=spmm_bytes, **rot_err(R))

p = _lib.default_params(); p.iters = 100; p.lr = 0.01
t0 = time.perf_counter()
pg = _lib.solve(prob, p)
out["desc_pgd_solve"] = dict(ms_wall=(time.perf_counter() - t0) * 1e3, ms_structure=pg["ms_structure"], ms_pgd=pg["ms_pgd"], iters=pg["iters_run"],
                             mean_abs_err_s=float(np.mean(np.abs(pg["S_vec"] - mo.ErrVec))))
S = pg["S_vec"]
t0 = time.perf_counter()
Rg, info = _lib.spectral_run(prob, weights=1.0 / (S ** 1.5 + 1e-8), normalize_rows=True)
out["gcw"] = dict(ms_wall=(time.perf_counter() - t0) * 1e3, ms_lib=info["ms_total"], outer_iters=info["iters"], spmm_products=info["products"],
                  residual=info["residual"], converged=info["converged"], **rot_err(Rg))
t0 = time.perf_counter()
Rr, info = _lib.refine_run(prob, S, Rg)
out["refine"] = dict(ms_wall=(time.perf_counter() - t0) * 1e3, ms_lib=info["ms_total"], iters=info["iters"], cg_iters=info["cg_iters"], score=info["score"], **rot_err(Rr))
beta = [1, 2, 4, 8, 16, 32]            # Demo/compare_algorithms.m:26-28: reweighting 2.^((1:6)-1), nsample 50
t0 = time.perf_counter()
Sc, ms = _lib.cemp_run(prob, beta, len(beta), 50)
out["cemp"] = dict(ms_wall=(time.perf_counter() - t0) * 1e3, ms_lib=ms, rounds=len(beta), nsample=50, mean_abs_err_s=float(np.mean(np.abs(Sc - mo.ErrVec))))
print(json.dumps(out))

# ---- the same stages on one device-resident problem (desc_problem_upload): what DESC() does since round 2
from desc_amd import DESC, DESC_PGD, ConstantStepSize
t0 = time.perf_counter()
dp = _lib.DeviceProblem(prob, 0)
t_up = time.perf_counter() - t0
res = {"upload_ms": t_up * 1e3}
t0 = time.perf_counter(); Rg2, gi = _lib.gcw_run(dp, S); res["gcw_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); Rr2, ri = _lib.refine_run(dp, S, Rg2); res["refine_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); Rs2, si = _lib.spectral_run(dp); res["spectral_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); Sc2, _ = _lib.cemp_run(dp, beta, len(beta), 50); res["cemp_ms"] = (time.perf_counter() - t0) * 1e3
res["max_diff_vs_host_problem_path"] = dict(gcw=float(np.abs(Rotation_Alignment(Rg2, Rg)[0] - Rg).max()), cemp=float(np.abs(Sc2 - Sc).max()))
res["refine_vs_truth"] = rot_err(Rr2)
dp.free()
params = dict(iters=100, learning_rate=0.01, make_plots=False, Gradient=ConstantStepSize(0.01), verbose=False)
t0 = time.perf_counter(); Sp = DESC_PGD(mo.Ind, mo.RijMat, params); res["DESC_PGD_wrapper_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); Re, Ri, Sv = DESC(mo.Ind, mo.RijMat, params); res["DESC_wrapper_ms"] = (time.perf_counter() - t0) * 1e3
res["DESC_minus_DESC_PGD_ms"] = res["DESC_wrapper_ms"] - res["DESC_PGD_wrapper_ms"]
res["desc_pgd_solve_ms"] = out["desc_pgd_solve"]["ms_wall"]
res["DESC_rot_err"] = rot_err(Re)
print(json.dumps({"device_resident_problem": res}))

# ---- rooflines of the next rows' dominant kernels (HBM-bound; peak 8 TB/s as in bench.py).  Algorithmic bytes:
#   block SpMM (Spectral.m:27-37 as a sparse product): per CSR slot the 72-B block + its 4-B column index once, the 144-B operand row
#       (6 vectors x 3 components) gathered, per node row 144 B written                       -> 2m (72 + 4 + 144) + n 144 per product
#   CEMP round (CEMP.m:107-126): per sample S0 8 B + two edge ids 8 B + two S gathers 16 B    -> 32 m_pos nsample + 8 m per round
#   CEMP S0 (CEMP.m:80-101): per sample two 72-B blocks gathered + two ids 8 B + k 4 B + S0 written 8 B -> 164 per sample (+ 72 per edge)
dp = _lib.DeviceProblem(prob, 0)
v = _lib.spmm_variants(dp, reps=30)
dp.free()
roof = {"peak_GBs": 8000.0}
roof["k_bsr_spmm"] = dict(ms=v["ms_valu"], bytes=spmm_bytes, achieved_GBs=spmm_bytes / v["ms_valu"] / 1e6, frac=spmm_bytes / v["ms_valu"] / 1e6 / 8000.0,
                          matrix_bytes_only=2 * m * 72, matrix_GBs=2 * m * 72 / v["ms_valu"] / 1e6)
roof["k_bsr_spmm_mfma_f64_4x4x4"] = dict(ms=v["ms_mfma"], layout_code=v["mfma_layout"], max_abs_diff_vs_valu=v["max_abs_diff"],
                                         achieved_GBs=(spmm_bytes / v["ms_mfma"] / 1e6) if v["ms_mfma"] > 0 else None,
                                         verdict="MFMA form slower" if v["ms_mfma"] > v["ms_valu"] else "MFMA form faster")
print(json.dumps({"next_row_rooflines": roof}))
