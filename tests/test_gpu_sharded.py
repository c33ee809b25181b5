"""GPU tests of the multi-GPU path on ONE card: (a) world = 1, 2, 3 emulated in one process
(every rank's shard lives on the same GPU, collectives done by hand between their
exchange buffers), (b) two real processes with torch.distributed (gloo, staged through
host memory) sharing the card.  Both against the single-process oracle."""
import os
import socket

import numpy as np
import pytest

from tests.helpers import c_params, make_problem, oracle_reference

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _emulate(lib, nn, ii, jj, rij, p, world, check_every=5, where=None, nmin=30):
    import torch
    from desc_amd.sharded import HipShard
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    st = lib.Structure.build(prob, nmin, p.seed, lib.BUILD_HOST if where is None else where, 0)
    stream = torch.cuda.Stream(torch.device("cuda", 0))      # one stream for every emulated rank
    shards = [HipShard(prob, st, 0, r, world, stream=stream) for r in range(world)]
    st.free()
    L = shards[0].slice_len
    ctx = torch.cuda.stream(stream)
    ctx.__enter__()

    def all_gather():
        for r in range(world):
            piece = shards[r].sall.view(world, L)[r].clone()
            for s in shards:
                s.sall.view(world, L)[r].copy_(piece)

    def reduce_scatter():                                    # part by part (desc_shard_info.xparts): blocks [c * world, (c + 1) * world) -> block c
        Lp, X = shards[0].info.t_part, shards[0].info.xparts
        for c in range(X):
            tot = torch.zeros(world * Lp, dtype=shards[0].T.dtype, device=shards[0].T.device)
            for s in shards:
                tot += s.T[c * world * Lp:(c + 1) * world * Lp]
            for r, s in enumerate(shards):
                s.T_recv[c * Lp:(c + 1) * Lp].copy_(tot.view(world, Lp)[r])

    for s in shards: s.reset(p)
    for s in shards: s.finish(1)
    all_gather()
    for s in shards: s.finish(2)
    left = p.iters
    while left > 0:
        n = min(left, check_every)
        for _ in range(n):
            for s in shards: s.colsum()
            reduce_scatter()
            for s in shards: s.sweep()
            all_gather()
            for s in shards: s.finish(0)
        left -= n
        flags = [s.stopped() for s in shards]
        assert len(set(flags)) == 1
        if left > 0 and flags[0]:
            break
    for s in shards: s.objective(0)
    all_gather()
    for s in shards: s.objective(1)
    outs = [s.download() for s in shards]
    ctx.__exit__(None, None, None)
    segs = [(s.info.seg_lo, s.info.seg_hi, s.info.cyc_lo, s.info.cyc_hi) for s in shards]
    for o, s in zip(outs, shards):
        o["last_sweep"], o["kernel"] = s.solver.last_sweep(), s.solver.kernel_name()
    for s in shards: s.destroy()
    return outs, segs


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("case", ["const", "sampling_piecewise", "early_stop"])
def test_sharded_emulated_on_one_gpu(lib, oracle, world, case):
    cfg = dict(const=dict(n=60, p=0.5, iters=40, kw=dict(lr=0.01)),
               sampling_piecewise=dict(n=150, p=0.6, iters=25, kw=dict(lr=0.05, step_kind=1, decay_interval=4, t0=1)),
               early_stop=dict(n=40, p=0.5, iters=300, kw=dict(lr=1.0, patience=5, stop_tol=1e-3)))[case]
    mo, nn, ii, jj, rij = make_problem("uniform", n=cfg["n"], p=cfg["p"], q=0.2, sigma=0.1, seed=5)
    st, S0, ref = oracle_reference(oracle, nn, ii, jj, rij, seed=9, iters=cfg["iters"], **cfg["kw"])
    outs, segs = _emulate(lib, nn, ii, jj, rij, c_params(cfg["iters"], seed=9, **cfg["kw"]), world)
    assert segs[0][0] == 0 and segs[-1][1] == st["m_pos"] and segs[-1][3] == st["m_cycle"]
    for a, b in zip(segs[:-1], segs[1:]):
        assert a[1] == b[0] and a[3] == b[2]                 # contiguous, disjoint ownership
    for out in outs:
        assert out["iters_run"] == ref["iters_run"]
        assert np.abs(out["S_vec"] - ref["S_vec"]).max() <= TOL
        assert np.allclose(out["obj"], ref["obj"], rtol=1e-12, atol=1e-9)
        assert np.allclose(out["avg"], ref["avg"], rtol=1e-9, atol=1e-14)
    for out in outs[1:]:
        assert np.array_equal(out["S_vec"], outs[0]["S_vec"]) and np.array_equal(out["obj"], outs[0]["obj"])


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", DESC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from desc_amd import _lib
    from desc_amd.sharded import HipShard, ShardedDriver, TorchComm, init_distributed
    import torch.distributed as dist
    init_distributed()
    mo, nn, ii, jj, rij = make_problem("uniform", n=90, p=0.5, q=0.2, sigma=0.1, seed=6)
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    st = _lib.Structure.build(prob, 30, 2, _lib.BUILD_HOST, 0)
    shard = HipShard(prob, st, 0, rank, world)
    out = ShardedDriver(shard, TorchComm()).run(c_params(30, lr=0.01, seed=2))
    q.put((rank, out["S_vec"], out["obj"], out["iters_run"]))
    dist.barrier(); dist.destroy_process_group()
    shard.destroy()


def test_two_processes_one_gpu_gloo(oracle):
    import torch.multiprocessing as mp
    mo, nn, ii, jj, rij = make_problem("uniform", n=90, p=0.5, q=0.2, sigma=0.1, seed=6)
    st, S0, ref = oracle_reference(oracle, nn, ii, jj, rij, seed=2, iters=30, lr=0.01)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    for rank, S, obj, it in res:
        assert it == ref["iters_run"]
        assert np.abs(S - ref["S_vec"]).max() <= TOL
        assert np.allclose(obj, ref["obj"], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("n,p,nmin", [(150, 0.6, 30), (150, 0.9, 100)])
def test_sharded_emulated_device_built_structure(lib, oracle, world, n, p, nmin):
    """The bench's N > 1 configuration: structure built on the device, laid out per shard in place
    (segment ranges of the ranks, 16 and 32 lanes per segment)."""
    mo, nn, ii, jj, rij = make_problem("uniform", n=n, p=p, q=0.2, sigma=0.1, seed=6)
    st = oracle.build_structure(nn, ii, jj, seed=4, n_sample_min=nmin)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    ref = oracle.pgd_run(st, S0, 30, lr=0.01)
    outs, segs = _emulate(lib, nn, ii, jj, rij, c_params(30, lr=0.01, seed=4), world, where=lib.BUILD_DEVICE, nmin=nmin)
    for o in outs:
        assert o["iters_run"] == ref["iters_run"]
        assert np.abs(o["S_vec"] - ref["S_vec"]).max() <= TOL
        assert np.allclose(o["obj"], ref["obj"], rtol=1e-12, atol=1e-9)


# ---------------------------------------------------------------- fused C protocol
@pytest.mark.parametrize("case", ["const", "sampling_piecewise", "early_stop"])
def test_fused_protocol_one_rank(lib, oracle, case, monkeypatch):
    """desc_pgd_shard_run with world = 1: the whole two-stream iteration loop in C (no collectives) == oracle; then the
    same with the real RCCL entry points of the process forced on a one-rank communicator (plumbing of the function
    pointers: signature, datatype / op codes, stream)."""
    from desc_amd.sharded import NativeShard, RcclComm
    cfg = dict(const=dict(n=60, p=0.5, iters=40, kw=dict(lr=0.01)),
               sampling_piecewise=dict(n=150, p=0.6, iters=25, kw=dict(lr=0.05, step_kind=1, decay_interval=4, t0=1)),
               early_stop=dict(n=40, p=0.5, iters=300, kw=dict(lr=1.0, patience=5, stop_tol=1e-3)))[case]
    mo, nn, ii, jj, rij = make_problem("uniform", n=cfg["n"], p=cfg["p"], q=0.2, sigma=0.1, seed=5)
    st, S0, ref = oracle_reference(oracle, nn, ii, jj, rij, seed=9, iters=cfg["iters"], **cfg["kw"])
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    for mode in ("none", "rccl"):
        hst = lib.Structure.build(prob, 30, 9, lib.BUILD_HOST, 0)
        comm = None
        if mode == "rccl":
            monkeypatch.setenv("DESC_DEBUG_FORCE_COLLECTIVES", "1")
            comm = RcclComm(0, 1, 0)
            assert comm.ok and comm.count == 1          # ncclCommCount of the communicator the library will call into
        shard = NativeShard(prob, hst, 0, 0, 1, comm)
        hst.free()
        out = shard.run(c_params(cfg["iters"], seed=9, check_every=7, **cfg["kw"]))
        shard.destroy()
        if comm is not None:
            comm.destroy()
        assert out["iters_run"] == ref["iters_run"], mode
        assert np.abs(out["S_vec"] - ref["S_vec"]).max() <= TOL
        assert np.allclose(out["obj"], ref["obj"], rtol=1e-12, atol=1e-9)
        assert np.allclose(out["avg"], ref["avg"], rtol=1e-9, atol=1e-14)


def _worker_fused(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", DESC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from desc_amd import _lib
    from desc_amd.sharded import NativeShard, TrampolineComm, init_distributed
    import torch.distributed as dist
    init_distributed()
    mo, nn, ii, jj, rij = make_problem("uniform", n=90, p=0.5, q=0.2, sigma=0.1, seed=6)
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    st = _lib.Structure.build(prob, 30, 2, _lib.BUILD_DEVICE, 0)
    comm = TrampolineComm(device=0)
    shard = NativeShard(prob, st, 0, rank, world, comm)
    out = shard.run(c_params(30, lr=0.01, seed=2, check_every=4))
    q.put((rank, out["S_vec"], out["obj"], out["iters_run"]))
    dist.barrier(); dist.destroy_process_group()
    shard.destroy()


def test_fused_protocol_two_processes_one_gpu(oracle):
    """The fused C loop with a real exchange between two processes (collectives = Python trampolines staged over gloo)."""
    import torch.multiprocessing as mp
    mo, nn, ii, jj, rij = make_problem("uniform", n=90, p=0.5, q=0.2, sigma=0.1, seed=6)
    st, S0, ref = oracle_reference(oracle, nn, ii, jj, rij, seed=2, iters=30, lr=0.01)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fused, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120); assert p.exitcode == 0
    for rank, S, obj, it in res:
        assert it == ref["iters_run"]
        assert np.abs(S - ref["S_vec"]).max() <= TOL
        assert np.allclose(obj, ref["obj"], rtol=1e-12, atol=1e-9)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


# ---------------------------------------------------------------- the kernels a multi-GPU run executes
# A C4 shard runs k_sweep_band<...,XT=true> (exchange positions per segment, S into the all-gather slice), k_xpos / k_xt (the exchange
# layout) and k_unpack_S.  Small graphs take k_sweep_node by default and have ONE band (every rank but the first would be empty): the
# tests below force the band sweep (DESC_DEBUG_VARIANT=3) and shrink the LDS budget of a band (DESC_DEBUG_ROW_CAP) so that the graph is cut
# into dozens of bands and world = 8 gives every rank work.  Coupling under test: DESC_PGD.m:185-193.
BAND_CASES = dict(
    const=dict(n=150, p=0.6, iters=30, kw=dict(lr=0.01), env={}),                                      # sampling regime, <16,4> / <16,2> shapes
    jmajor_tail=dict(n=200, p=0.5, iters=25, kw=dict(lr=0.01), env=dict(DESC_DEBUG_JMAJOR="1", DESC_DEBUG_JBLOCK="24", DESC_DEBUG_TAIL="200")),
    piecewise=dict(n=150, p=0.6, iters=25, kw=dict(lr=0.05, step_kind=1, decay_interval=4, t0=1), env={}),
    adam=dict(n=120, p=0.6, iters=20, kw=dict(lr=0.01, step_kind=2), env={}),
    early_stop=dict(n=60, p=0.5, iters=300, kw=dict(lr=1.0, patience=5, stop_tol=1e-3), env={}),
    short_segments=dict(n=40, p=0.5, iters=30, kw=dict(lr=0.01), env={}),                              # <= 16 cycles: the 1024-thread instance
)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", sorted(BAND_CASES))
def test_sharded_band_sweep_emulated(lib, oracle, world, case, monkeypatch):
    cfg = BAND_CASES[case]
    monkeypatch.setenv("DESC_DEBUG_VARIANT", "3")
    monkeypatch.setenv("DESC_DEBUG_ROW_CAP", str(max(64, int(cfg["n"] * cfg["n"] * cfg["p"] / 24))))     # ~24 bands
    for k, v in cfg["env"].items():
        monkeypatch.setenv(k, v)
    mo, nn, ii, jj, rij = make_problem("uniform", n=cfg["n"], p=cfg["p"], q=0.2, sigma=0.1, seed=8)
    st, S0, ref = oracle_reference(oracle, nn, ii, jj, rij, seed=3, iters=cfg["iters"], **cfg["kw"])
    outs, segs = _emulate(lib, nn, ii, jj, rij, c_params(cfg["iters"], seed=3, **cfg["kw"]), world, where=lib.BUILD_DEVICE)
    assert segs[0][0] == 0 and segs[-1][1] == st["m_pos"] and segs[-1][3] == st["m_cycle"]
    busy = sum(1 for a in segs if a[1] > a[0])
    assert busy >= min(world, 3), segs                         # the small row cap gives (nearly) every rank a range of bands
    tol = 1e-9 if case == "adam" else TOL
    for out, sg in zip(outs, segs):
        assert "band" in out["kernel"]
        if sg[1] > sg[0]:
            assert "k_sweep_band" in out["last_sweep"] and ",XT>" in out["last_sweep"], out["last_sweep"]
        assert out["iters_run"] == ref["iters_run"]
        assert np.abs(out["S_vec"] - ref["S_vec"]).max() <= tol
        assert np.allclose(out["obj"], ref["obj"], rtol=1e-12, atol=1e-9)
        assert np.allclose(out["avg"], ref["avg"], rtol=1e-9, atol=1e-14)
    for out in outs[1:]:
        assert np.array_equal(out["S_vec"], outs[0]["S_vec"]) and np.array_equal(out["obj"], outs[0]["obj"])
    # ... and bitwise what ONE rank computes with the same kernel shapes (fixed-point mirror sums: no order dependence anywhere)
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    dst = lib.Structure.build(prob, 30, 3, lib.BUILD_DEVICE, 0)
    one = _unsharded(lib, prob, dst, c_params(cfg["iters"], seed=3, **cfg["kw"]))
    dst.free()
    assert "one-rank" in one["last_sweep"] and one["iters_run"] == outs[0]["iters_run"]
    assert np.array_equal(one["S_vec"], outs[0]["S_vec"])


def _csr(nn, ii, jj):
    """CSR adjacency (neighbours ascending) with the edge id of every slot, as the library builds it."""
    m = ii.shape[0]
    src = np.concatenate([ii, jj]); dst = np.concatenate([jj, ii]); eid = np.concatenate([np.arange(m), np.arange(m)])
    order = np.lexsort((dst, src))
    rowptr = np.zeros(nn + 1, dtype=np.int64)
    np.add.at(rowptr, src + 1, 1)
    return np.cumsum(rowptr), dst[order], eid[order], src[order]


@pytest.mark.parametrize("world,row_cap", [(2, 700), (3, 700), (8, 700), (8, 0)])
def test_exchange_layout_invariants(lib, world, row_cap, monkeypatch):
    """k_xpos / k_xt: xpos is a bijection of the 2m CSR slots into the owners' parts of the reduce-scatter buffer, {ta, tb} of a segment are
    the xpos of its two slots, spos = (owner of the smaller endpoint, edge number inside the owner's range); the same on every rank.
    row_cap = 0: the default LDS budget, one band -> every rank but the first owns an empty node range."""
    monkeypatch.setenv("DESC_DEBUG_VARIANT", "3")
    if row_cap:
        monkeypatch.setenv("DESC_DEBUG_ROW_CAP", str(row_cap))
    mo, nn, ii, jj, rij = make_problem("uniform", n=130, p=0.55, q=0.2, sigma=0.1, seed=12)
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    st = lib.Structure.build(prob, 30, 1, lib.BUILD_DEVICE, 0)
    m = ii.shape[0]
    rowptr, adj, adj_eid, row_of = _csr(nn, ii, jj)
    layouts, infos = [], []
    for r in range(world):
        sv = lib.Solver(prob, st, 0, r, world)
        layouts.append(sv.shard_layout()); infos.append(sv.shard_info())
        sv.destroy()
    st.free()
    t_part, slice_len, X = infos[0].t_part, infos[0].slice_len, infos[0].xparts
    assert all(i.t_part == t_part and i.slice_len == slice_len and i.xparts == X for i in infos)
    xpos, spos = layouts[0]["xpos"], layouts[0]["spos"]
    for lay in layouts[1:]:                                   # the layout is global: identical on every rank
        assert np.array_equal(lay["xpos"], xpos) and np.array_equal(lay["spos"], spos)
    assert np.unique(xpos).size == 2 * m and xpos.min() >= 0 and xpos.max() < world * X * t_part       # injective
    # block b = part * world + rank: the owner of a slot's column sum = the owner of the SMALLER endpoint of its edge
    block = xpos // t_part
    owner, part = block % world, block // world
    vowner = owner * X + part                                 # virtual owners in node order
    small = np.minimum(row_of, adj)
    for vo in np.unique(vowner):
        mine, others = small[vowner == vo], small[vowner != vo]
        assert not np.any((others >= mine.min()) & (others <= mine.max())), vo                      # node ranges do not interleave
    e_vowner = np.zeros(m, dtype=np.int64); e_vowner[adj_eid] = vowner
    assert np.all(np.diff(e_vowner) >= 0)                     # edge list sorted by (i, j): (rank, part) own consecutive ranges
    e_owner = e_vowner // X
    e_lo = np.searchsorted(e_owner, np.arange(world))         # first edge of every rank
    e_lo_v = np.searchsorted(e_vowner, np.arange(world * X))  # ... of every (rank, part)
    assert np.array_equal(spos, e_owner[adj_eid] * slice_len + (adj_eid - e_lo[e_owner[adj_eid]]))
    # T1 half of a block: edge order; T2 half behind it
    upper = adj > row_of
    assert np.array_equal(xpos[upper], block[upper] * t_part + (adj_eid[upper] - e_lo_v[vowner[upper]]))
    assert np.all(xpos[~upper] - block[~upper] * t_part >= t_part // 2)
    covered = 0
    for r, (lay, info) in enumerate(zip(layouts, infos)):
        nsl = int(info.seg_hi - info.seg_lo)
        assert lay["xt"].shape == (nsl, 2)
        if nsl == 0:
            continue
        sa, sb = lay["slot_ab"][:, 0], lay["slot_ab"][:, 1]
        assert np.all(owner[sa] == r) and np.all(owner[sb] == r) and np.array_equal(block[sa], block[sb])
        assert np.array_equal(lay["xt"][:, 0], xpos[sa] - block[sa] * t_part) and np.array_equal(lay["xt"][:, 1], xpos[sb] - block[sb] * t_part)
        # place in the rank's all-gather slice = place inside the part + the edges of the parts before it
        assert np.array_equal(spos[sa], r * slice_len + lay["xt"][:, 0] + (e_lo_v[vowner[sa]] - e_lo[r])) and np.array_equal(spos[sa], spos[sb])
        covered += nsl
    assert covered == infos[0].m_pos
    if not row_cap:
        assert sum(1 for i in infos if i.seg_hi > i.seg_lo) == 1


def _unsharded(lib, prob, st, p):
    solver = lib.Solver(prob, st, 0)
    out = solver.run(p)
    out["kernel"], out["last_sweep"] = solver.kernel_name(), solver.last_sweep()
    solver.destroy()
    return out


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_c2_full_size_equals_unsharded(lib, world):
    """BASELINE configs[1] (n = 1000, 15.7 M cycles), device-built structure, default kernel choice: every emulated rank runs the XT instance
    of the band sweep on its range of bands; S_vec / objective of every rank within 1e-12 of the one-GPU run and bitwise equal across ranks."""
    import bench
    mo, nn, ii, jj, rij = bench.generate("C2")
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    p = c_params(30, lr=0.01, seed=0)
    st = lib.Structure.build(prob, 30, 0, lib.BUILD_DEVICE, 0)
    one = _unsharded(lib, prob, st, p)
    st.free()
    assert "band" in one["kernel"] and "one-rank" in one["last_sweep"]
    outs, segs = _emulate(lib, nn, ii, jj, rij, p, world, where=lib.BUILD_DEVICE)
    assert all(b > a for a, b, _, _ in segs)                   # every rank owns bands
    cyc = [d - c for _, _, c, d in segs]
    assert max(cyc) <= 1.5 * (sum(cyc) / world), cyc           # whole-band cuts: 28 bands of unequal weight at C2 (C4 has 263: tools/shard_compute.py)
    for out in outs:
        assert "k_sweep_band" in out["last_sweep"] and ",XT>" in out["last_sweep"], out["last_sweep"]
        assert out["iters_run"] == one["iters_run"] == 30
        # the mirror sums are exchanged as fixed-point integers (order-independent): S_vec is BITWISE what one GPU computes; the objective
        # adds the workgroups' partials in another partition
        assert np.array_equal(out["S_vec"], one["S_vec"])
        assert np.allclose(out["obj"], one["obj"], rtol=1e-12, atol=0)
        assert np.allclose(out["avg"], one["avg"], rtol=1e-9, atol=1e-15)
    for out in outs[1:]:
        assert np.array_equal(out["S_vec"], outs[0]["S_vec"]) and np.array_equal(out["obj"], outs[0]["obj"])


def test_fused_protocol_forced_collectives_c2(lib, monkeypatch):
    """The fused two-stream C protocol on the multi-rank code path (exchange layout, XT sweep, real RCCL calls on a one-rank communicator,
    unpack) at C2 against the one-GPU run."""
    import bench
    from desc_amd.sharded import NativeShard, RcclComm
    mo, nn, ii, jj, rij = bench.generate("C2")
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    p = c_params(30, lr=0.01, seed=0, check_every=8)
    st = lib.Structure.build(prob, 30, 0, lib.BUILD_DEVICE, 0)
    one = _unsharded(lib, prob, st, p)
    monkeypatch.setenv("DESC_DEBUG_FORCE_COLLECTIVES", "1")
    comm = RcclComm(0, 1, 0)
    assert comm.ok and comm.count == 1
    shard = NativeShard(prob, st, 0, 0, 1, comm)
    st.free()
    out = shard.run(p)
    name = shard.solver.last_sweep()
    shard.destroy(); comm.destroy()
    assert "k_sweep_band" in name and ",XT>" in name, name
    assert out["iters_run"] == 30
    assert np.array_equal(out["S_vec"], one["S_vec"])
    assert np.allclose(out["obj"], one["obj"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("parts", ["1", "3", "4"])
def test_exchange_parts_other_counts(lib, oracle, parts, monkeypatch):
    """DESC_SHARD_PARTS other than the default 2 (1 = one reduce-scatter as in round 3; 3, 4: parts that need not divide a rank's bands evenly; some
    may be empty): the same bits as one rank, for world 3 on a graph of ~24 bands."""
    monkeypatch.setenv("DESC_DEBUG_VARIANT", "3")
    monkeypatch.setenv("DESC_DEBUG_ROW_CAP", "560")
    monkeypatch.setenv("DESC_SHARD_PARTS", parts)
    mo, nn, ii, jj, rij = make_problem("uniform", n=150, p=0.6, q=0.2, sigma=0.1, seed=8)
    st, S0, ref = oracle_reference(oracle, nn, ii, jj, rij, seed=3, iters=25, lr=0.01)
    outs, segs = _emulate(lib, nn, ii, jj, rij, c_params(25, lr=0.01, seed=3), 3, where=lib.BUILD_DEVICE)
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    dst = lib.Structure.build(prob, 30, 3, lib.BUILD_DEVICE, 0)
    one = _unsharded(lib, prob, dst, c_params(25, lr=0.01, seed=3))
    dst.free()
    for out in outs:
        assert ",XT>" in out["last_sweep"]
        assert np.abs(out["S_vec"] - ref["S_vec"]).max() <= TOL and np.array_equal(out["S_vec"], one["S_vec"])
        assert np.allclose(out["obj"], ref["obj"], rtol=1e-12, atol=1e-9)
