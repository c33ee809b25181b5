#!/usr/bin/env python3
"""What ONE GPU can measure about a world-N run (default N = 8): every rank's shard of the workload is created on the one card and a
faithful iteration loop is run over all of them (collectives done by hand between their exchange buffers, as tests/test_gpu_sharded.py
does), each rank's column sums / sweep / unpack bracketed by HIP events.  A rank alone on the card = that rank on its own GPU, so the
per-rank times are what the N GPUs would compute in parallel; the exchange itself is NOT measured here (one card: no xGMI).

    python tools/shard_compute.py --workload C4 --world 8 --steps 10 --warmup 3 > profiles/r04_shard_w8_c4.json

Reports per rank: cycles / segments / pieces, HBM footprint, us per piece of the iteration, bytes sent and received per collective;
max / mean over ranks (load balance).  DESC_PGD.m:185-193 is what the ranks exchange."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from desc_amd import _lib as lib
from desc_amd.sharded import HipShard

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C4")
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()
world, K, W = args.world, args.steps, args.warmup

bench.warm_up(lib)
mo, nn, ii, jj, rij = bench.generate(args.workload)
prob = lib.ProblemArrays(nn, ii, jj, rij)
st = lib.Structure.build(prob, 30, 0, lib.BUILD_DEVICE, 0)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)

# one GPU, unsharded: the reference point
p = lib.default_params(); p.iters = W + K + 4; p.lr = 0.01; p.patience = (1 << 31) - 1; p.seed = 0
solo = lib.Solver(prob, st, 0)
solo.reset(p); solo.iterate(W); solo.sync()
_, ms_pair = solo.iterate_timed(K, per_kernel=True)
solo_name = solo.kernel_name(); solo_lay = solo.layout_stats()
solo.destroy()
lib.trim_memory()

shards, foot = [], []
for r in range(world):
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    shards.append(HipShard(prob, st, 0, r, world, stream=stream))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    foot.append(free0 - free1)
st.free()
L = shards[0].slice_len
Lp = shards[0].info.t_part


def all_gather():
    for r in range(world):
        piece = shards[r].sall.view(world, L)[r].clone()
        for s in shards:
            s.sall.view(world, L)[r].copy_(piece)


def reduce_scatter():                                        # part by part (desc_shard_info.xparts)
    X = shards[0].info.xparts
    for c in range(X):
        tot = torch.zeros(world * Lp, dtype=shards[0].T.dtype, device=dev)
        for s in shards:
            tot += s.T[c * world * Lp:(c + 1) * world * Lp]
        for r, s in enumerate(shards):
            s.T_recv[c * Lp:(c + 1) * Lp].copy_(tot.view(world, Lp)[r])


pieces = ("colsum", "sweep", "unpack")
ev = {k: [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)] for _ in range(world)] for k in pieces}
with torch.cuda.stream(stream):
    for s in shards: s.reset(p)
    for s in shards: s.finish(1)
    all_gather()
    for s in shards: s.finish(2)
    for it in range(W + K):
        k = it - W
        for r, s in enumerate(shards):
            if k >= 0: ev["colsum"][r][k][0].record(stream)
            s.colsum()
            if k >= 0: ev["colsum"][r][k][1].record(stream)
        reduce_scatter()
        for r, s in enumerate(shards):
            if k >= 0: ev["sweep"][r][k][0].record(stream)
            s.sweep()
            if k >= 0: ev["sweep"][r][k][1].record(stream)
        all_gather()
        for r, s in enumerate(shards):
            if k >= 0: ev["unpack"][r][k][0].record(stream)
            s.finish(0)
            if k >= 0: ev["unpack"][r][k][1].record(stream)
    for s in shards: s.objective(0)
    all_gather()
    for s in shards: s.objective(1)
    outs = [s.download() for s in shards]
torch.cuda.synchronize()
us = {k: [float(np.mean([a.elapsed_time(b) for a, b in ev[k][r]])) * 1e3 for r in range(world)] for k in pieces}
# Second timing, one rank at a time WITHOUT the others in between: in the loop above the eight ranks' tables (8 x ~1 GB) evict each other from the
# 256 MB Infinity Cache between a rank's turns, which a rank alone on its own GPU does not suffer.  The exchange buffers keep the last iteration's
# content (the arithmetic is the same; the results of this phase are discarded).
ev2 = {k: [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)] for _ in range(world)] for k in pieces}
p2 = lib.default_params(); p2.iters = W + K + 4; p2.lr = 0.01; p2.patience = (1 << 31) - 1; p2.seed = 0
with torch.cuda.stream(stream):
    for r, s in enumerate(shards):
        s.reset(p2); s.finish(1); s.finish(2)
        for it in range(W + K):
            k = it - W
            for name, call in (("colsum", s.colsum), ("sweep", s.sweep), ("unpack", lambda: s.finish(0))):
                if k >= 0: ev2[name][r][k][0].record(stream)
                call()
                if k >= 0: ev2[name][r][k][1].record(stream)
torch.cuda.synchronize()
us_alone = {k: [float(np.mean([a.elapsed_time(b) for a, b in ev2[k][r]])) * 1e3 for r in range(world)] for k in pieces}
for o in outs[1:]:
    assert np.array_equal(o["S_vec"], outs[0]["S_vec"]) and np.array_equal(o["obj"], outs[0]["obj"])
ranks = []
for r, s in enumerate(shards):
    lay = s.solver.layout_stats()
    ranks.append(dict(rank=r, cycles=int(s.info.cyc_hi - s.info.cyc_lo), segments=int(s.info.seg_hi - s.info.seg_lo), pieces=lay["pieces"],
                      colsum_entries=lay["colsum_entries"], band_row_entries_per_sweep=lay["piece_row_entries"],
                      hbm_bytes=int(foot[r]), sweep_kernel=s.solver.last_sweep(),
                      us_colsum=us["colsum"][r], us_sweep=us["sweep"][r], us_unpack=us["unpack"][r],
                      us_colsum_alone=us_alone["colsum"][r], us_sweep_alone=us_alone["sweep"][r], us_unpack_alone=us_alone["unpack"][r]))
for s in shards: s.destroy()


def stat(key):
    v = np.array([x[key] for x in ranks], dtype=np.float64)
    return dict(max=float(v.max()), mean=float(v.mean()), min=float(v.min()), max_over_mean=float(v.max() / v.mean()))


m = prob.m
out = dict(
    workload=bench.describe(args.workload), world=world, steps=K, warmup=W, n=nn, m=m, m_cycle=int(sum(x["cycles"] for x in ranks)),
    one_gpu=dict(kernel=solo_name, us_kernel_pair=ms_pair * 1e3, layout=solo_lay),
    ranks=ranks,
    balance={k: stat(k) for k in ("cycles", "us_colsum", "us_sweep", "us_unpack", "us_colsum_alone", "us_sweep_alone", "us_unpack_alone", "hbm_bytes")},
    compute_us_max_over_ranks=float(max(x["us_colsum"] + x["us_sweep"] + x["us_unpack"] for x in ranks)),
    compute_us_max_over_ranks_alone=float(max(x["us_colsum_alone"] + x["us_sweep_alone"] + x["us_unpack_alone"] for x in ranks)),
    exchange=dict(
        reduce_scatter=dict(parts=int(shards[0].info.xparts), elements_per_block=int(Lp), bytes_sent_per_rank=int(8 * Lp * shards[0].info.xparts * (world - 1)),
                            bytes_received_per_rank=int(8 * Lp * shards[0].info.xparts * (world - 1)),
                            what="partial mirror sums T1 | T2 (DESC_PGD.m:189-190) of every other rank's edges out, the other ranks' partials of this rank's edges in"),
        all_gather=dict(slice_len=int(L), bytes_sent_per_rank=int(8 * L * (world - 1)), bytes_received_per_rank=int(8 * L * (world - 1)),
                        what="new S of the owned edges (DESC_PGD.m:229) + the workgroup partials of the objective / |dS| sums")),
    checks=dict(ranks_bitwise_equal=True, mean_abs_err_vs_truth=float(np.mean(np.abs(outs[0]["S_vec"] - mo.ErrVec)))),
    note="per-rank times measured on ONE MI355X: us_* in a faithful loop over all ranks (their tables evict each other from the Infinity Cache between turns), "
         "us_*_alone with one rank iterating by itself (closer to that rank on its own GPU); collectives emulated by hand and not timed",
)
print(json.dumps(out, indent=1))
