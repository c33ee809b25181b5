"""CPU tests of the multi-rank path (no GPU): the driver protocol of desc_amd/sharded.py with
torch.distributed over gloo, world_size 2 and 3, against the single-process oracle.  The
per-rank arithmetic is a NumPy stand-in (tests/numpy_shard.py); what is under test is the
sharding bookkeeping, the exchange protocol and the rank-consistent stop rule."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import c_params, make_problem


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["ORACLE_THREADS"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from desc_amd.sharded import ShardedDriver, TorchComm
    from oracle import oracle as O
    from tests.numpy_shard import NumpyShard
    mo, nn, ii, jj, rij = make_problem("uniform", n=case["n"], p=case["p"], q=0.2, sigma=0.1, seed=case["seed"])
    st = O.build_structure(nn, ii, jj, seed=4)
    S0 = O.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    shard = NumpyShard(st, S0, rank, world)
    drv = ShardedDriver(shard, TorchComm())
    out = drv.run(c_params(case["iters"], **case["kw"]), check_every=case.get("check_every", 4))
    q.put((rank, out["S_vec"], out["obj"], out["avg"], out["iters_run"]))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    dict(n=36, p=0.5, seed=1, iters=12, kw=dict(lr=0.01)),
    dict(n=70, p=0.6, seed=2, iters=8, kw=dict(lr=0.05, step_kind=1, decay_interval=3, t0=2)),      # sampling regime
    dict(n=30, p=0.5, seed=3, iters=200, kw=dict(lr=1.0, patience=4, stop_tol=1e-3), check_every=3),  # early stop
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", CASES, ids=["const", "piecewise_sampling", "early_stop"])
def test_sharded_protocol_gloo(oracle, world, case):
    mo, nn, ii, jj, rij = make_problem("uniform", n=case["n"], p=case["p"], q=0.2, sigma=0.1, seed=case["seed"])
    st = oracle.build_structure(nn, ii, jj, seed=4)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    ref = oracle.pgd_run(st, S0, case["iters"], **case["kw"])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, S, obj, avg, it in res:
        assert it == ref["iters_run"], (rank, it, ref["iters_run"])
        assert np.abs(S - ref["S_vec"]).max() < 1e-12
        assert np.allclose(obj, ref["obj"], rtol=1e-12)
        assert np.allclose(avg, ref["avg"], rtol=1e-9, atol=1e-16)
    # every rank returns bitwise the same answer (scalars are added in rank order everywhere)
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` as ONE process (no WORLD_SIZE in the environment) must start two ranks itself; --dry-launch
    lets them meet over gloo and count themselves without touching a GPU.  A mismatch between --gpus and the ranks that
    actually exist is an error, not an `n_gpus: 1` line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_counted"] == 2 and rec["ok"] is True
    assert rec["problem_shared"] is True and rec["generator_calls"] == 1      # rank 0 generated, rank 1 mapped the same arrays
    # launched with the wrong number of ranks (torchrun-style environment of ONE rank, --gpus 2): refuses
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
