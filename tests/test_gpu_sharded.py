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

    def reduce_scatter():
        Lp = shards[0].info.t_part
        tot = torch.zeros(world * Lp, dtype=torch.float64, device=shards[0].T.device)
        for s in shards:
            tot += s.T[:world * Lp]
        for r, s in enumerate(shards):
            s.T_recv.copy_(tot.view(world, Lp)[r])

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
