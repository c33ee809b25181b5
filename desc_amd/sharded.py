"""Multi-GPU DESC_PGD: one process per GPU, edges-with-cycles sharded across ranks,
O(m) exchange per iteration with torch.distributed (backend "nccl" = RCCL over xGMI).

Per iteration (node form):
    1. every rank: column sums of its own segments' weights        -> T_send (partial sums for
       every edge-with-cycles, grouped by the rank that owns the edge)
    2. reduce-scatter(sum) of T_send -> T_recv                      (the mirror-weight sums
       T1/T2 of DESC_PGD.m:185-191 couple edges of different ranks; a rank only needs the
       totals of its own edges)
    3. every rank: sweep of its own chunks (gradient, projection, new S of its edges);
       packs S of its edges + its two scalar partials into its slice of `sall`
    4. all-gather of sall                                           (S_vec is read by every
       rank's gathers, DESC_PGD.m:193)
    5. every rank: scatter S into its replica, add the scalar partials in rank order,
       traces + early-stop rule -> identical decisions on all ranks
The reference itself is single-process MATLAB; nothing here has a counterpart in it.

Two drivers:

* ``NativeShard`` (default on GPUs): the *fused* protocol of the C ABI -- ``desc_pgd_shard_start / _iterate / _run``
  enqueue whole iterations from C on two streams (the all-gather of S and its unpacking overlap the next column-sum
  pass); the collectives are function pointers.  ``RcclComm`` hands over the ``ncclReduceScatter`` / ``ncclAllGather``
  entry points of the ``librccl.so`` PyTorch ships (the copy already mapped into the process) and a communicator created
  with ``ncclCommInitRank`` from an id broadcast over ``torch.distributed``: no Python between iterations.
  ``TrampolineComm`` passes Python callbacks instead (staged through host memory over gloo): functional tests only.
* ``ShardedDriver`` + ``HipShard``: the *piecewise* protocol, one ctypes call per step with ``torch.distributed``
  collectives in between; backend-agnostic (the CPU tests plug in a NumPy shard), and the fallback of the bench when the
  native communicator cannot be created.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

from . import _lib


class TorchComm:
    """reduce-scatter / all-gather over a torch.distributed process group.  With the gloo
    backend device tensors are staged through host memory (functional tests only)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.staged = dist.get_backend(group) == "gloo"

    def reduce_scatter_sum(self, recv, send):
        """recv (t_part) = sum over ranks of their send[rank*t_part:(rank+1)*t_part]; send has one spare
        element at the end.  gloo has no reduce-scatter: all-reduce on the host, keep this rank's part."""
        L = recv.numel()
        body = send[:self.world * L]
        if self.staged:
            h = body.cpu() if body.is_cuda else body.clone()
            self.dist.all_reduce(h, group=self.group)
            recv.copy_(h.view(self.world, L)[self.rank])
        else:
            self.dist.reduce_scatter_tensor(recv, body, group=self.group)

    def all_gather_slices(self, full, slice_len):
        """full = world slices of slice_len; every rank has filled its own slice."""
        mine = full.view(self.world, slice_len)[self.rank]
        if self.staged:
            h = full.cpu() if full.is_cuda else full
            parts = [torch_empty_like(mine, h) for _ in range(self.world)]
            self.dist.all_gather(parts, h.view(self.world, slice_len)[self.rank].clone(), group=self.group)
            for r, p in enumerate(parts):
                h.view(self.world, slice_len)[r].copy_(p)
            if full.is_cuda:
                full.copy_(h)
        else:
            self.dist.all_gather_into_tensor(full, mine, group=self.group)

    def barrier(self):
        self.dist.barrier(group=self.group)

    def max_float(self, x):
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        if not self.staged:
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


def torch_empty_like(mine, host_full):
    import torch
    return torch.empty(mine.shape, dtype=host_full.dtype)


class SingleComm:
    """world_size 1 without torch."""
    rank, world = 0, 1

    def reduce_scatter_sum(self, recv, send): recv.copy_(send[:recv.numel()])
    def all_gather_slices(self, full, slice_len): pass
    def barrier(self): pass
    def max_float(self, x): return x


class HipShard:
    """One rank's share of the problem on its GPU, through the C ABI."""

    def __init__(self, prob, structure, device, rank, world, stream=None):
        import torch
        self.torch = torch
        self.solver = _lib.Solver(prob, structure, device, rank, world)
        self.info = self.solver.shard_info()
        dev = torch.device("cuda", device)
        torch.cuda.set_device(dev)
        # One side stream carries the library's kernels AND (as torch's current stream inside
        # stream_ctx) the collectives, so they are ordered without host synchronisation.
        self.stream = stream if stream is not None else torch.cuda.Stream(dev)
        with torch.cuda.stream(self.stream):
            # the mirror sums are exchanged as 64-bit fixed-point integers (k_colsum_node): integer sums do not depend on the order the
            # collective adds the ranks' parts in, so S_vec comes out bitwise the same for every number of ranks
            self.T = torch.zeros(self.info.t_len, dtype=torch.int64, device=dev)        # send: zero padding stays zero
            self.T_recv = torch.zeros(self.info.xparts * self.info.t_part, dtype=torch.int64, device=dev)
            self.sall = torch.zeros(self.info.world * self.info.slice_len, dtype=torch.float64, device=dev)
        self.stream.synchronize()
        self.solver.shard_bind(self.T.data_ptr(), self.T_recv.data_ptr(), self.sall.data_ptr(), self.stream.cuda_stream)
        self.slice_len = self.info.slice_len
        self.xparts, self.t_part = int(self.info.xparts), int(self.info.t_part)

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def reset(self, params): self.solver.reset(params)
    def colsum(self): self.solver.shard_colsum()
    def sweep(self): self.solver.shard_sweep()
    def finish(self, initial=0): self.solver.shard_finish(initial)
    def objective(self, phase): self.solver.shard_objective(phase)
    def stopped(self): return self.solver.stopped()
    def sync(self): self.solver.sync()
    def download(self): return self.solver.download()
    def destroy(self): self.solver.destroy()


class ShardedDriver:
    """Runs the iteration protocol; `shard` provides the compute steps, `comm` the collectives."""

    def __init__(self, shard, comm):
        self.shard, self.comm = shard, comm

    def _ctx(self):
        import contextlib
        return self.shard.stream_ctx() if hasattr(self.shard, "stream_ctx") else contextlib.nullcontext()

    def start(self, params):
        s, c = self.shard, self.comm
        with self._ctx():
            s.reset(params)
            s.finish(1)                                  # pack the initial S of the owned edges
            c.all_gather_slices(s.sall, s.slice_len)
            s.finish(2)                                  # every rank now holds the full initial S_vec

    def iterate(self, n):
        s, c = self.shard, self.comm
        with self._ctx():
            for _ in range(n):
                s.colsum()
                reduce_scatter_parts(c, s)
                s.sweep()
                c.all_gather_slices(s.sall, s.slice_len)
                s.finish(0)

    def finish(self):
        s, c = self.shard, self.comm
        with self._ctx():
            s.objective(0)
            c.all_gather_slices(s.sall, s.slice_len)
            s.objective(1)
            return s.download()

    def run(self, params, check_every=16):
        self.start(params)
        left = params.iters
        while left > 0:
            n = min(left, check_every)
            self.iterate(n)
            left -= n
            if left > 0 and self.shard.stopped():    # identical on every rank
                break
        return self.finish()


def reduce_scatter_parts(comm, shard):
    """The reduce-scatter of the mirror sums, one exchange part at a time (desc_shard_info.xparts; shards without the attribute have one part):
    part c = blocks [c * world, (c + 1) * world) of T -> block c of T_recv."""
    X = int(getattr(shard, "xparts", 1))
    if X == 1:
        comm.reduce_scatter_sum(shard.T_recv, shard.T)
        return
    L, W = int(shard.t_part), comm.world
    for c in range(X):
        comm.reduce_scatter_sum(shard.T_recv[c * L:(c + 1) * L], shard.T[c * W * L:(c + 1) * W * L + 1])


class RcclComm:
    """RCCL communicator created next to torch's own, for the fused C protocol.  Uses the librccl.so that ships
    with PyTorch -- the copy torch.distributed's "nccl" backend has already mapped (one ROCm runtime per process).

    Creation is agreed on by all ranks step by step (library load, unique id, ncclCommInitRank), each step followed by a
    MIN-reduction of a success flag over torch.distributed, so that no rank walks into the collective ncclCommInitRank
    while another has already given up; `ok` tells the caller whether every rank holds a communicator."""

    def __init__(self, rank, world, device):
        import torch
        import torch.distributed as dist
        self.comm = C.c_void_p()
        self.ok = False
        self.count = 0

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        def all_ok(flag):
            if world == 1:
                return bool(flag)
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=torch.device("cuda", device))
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        # 1. the library and its entry points
        try:
            self.lib = L = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
            L.ncclGetUniqueId.restype = C.c_int; L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            L.ncclCommInitRank.restype = C.c_int; L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            L.ncclCommCount.restype = C.c_int; L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            L.ncclCommDestroy.restype = C.c_int; L.ncclCommDestroy.argtypes = [C.c_void_p]
            self.reduce_scatter, self.all_gather = L.ncclReduceScatter, L.ncclAllGather
            loaded = True
        except (OSError, AttributeError) as e:
            self.error = repr(e); loaded = False
        if not all_ok(loaded):
            self.error = getattr(self, "error", "another rank could not load librccl.so")
            return
        # 2. rank 0's unique id reaches everybody (None if rank 0 could not make one)
        uid = UniqueId()
        box = [None]
        if rank == 0:
            rc = L.ncclGetUniqueId(C.byref(uid))
            box = [bytes(C.string_at(C.byref(uid), 128)) if rc == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            self.error = "ncclGetUniqueId failed on rank 0"
            return
        C.memmove(C.byref(uid), box[0], 128)
        # 3. the collective initialisation, then agreement on its outcome
        torch.cuda.set_device(device)
        rc = L.ncclCommInitRank(C.byref(self.comm), world, uid, rank)
        good = rc == 0 and bool(self.comm)
        cnt = C.c_int(0)
        if good and L.ncclCommCount(self.comm, C.byref(cnt)) == 0:
            self.count = int(cnt.value)
        good = good and self.count == world
        if not all_ok(good):
            self.error = f"ncclCommInitRank rc={rc}, ncclCommCount={self.count} (world {world}) on this or another rank"
            self.destroy()
            return
        self.ok = True

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


class TrampolineComm:
    """Python callbacks with the RCCL signatures, for the fused C protocol on backends other than RCCL (tests:
    two processes sharing one GPU over gloo).  Each call drains the device, moves the buffer through host memory
    and runs the torch.distributed collective: slow by construction."""

    def __init__(self, group=None, device=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.comm = None
        L = _lib.load()               # copies through the library's own HIP runtime (never a second copy of it)
        self.device = device

        def rs(send, recv, recvcount, dtype, op, comm, stream):
            try:
                L.desc_device_synchronize(self.device)
                h = torch.empty(self.world * recvcount, dtype=torch.int64 if dtype == 4 else torch.float64)   # 4 = ncclInt64 (fixed-point mirror sums), 8 = ncclDouble
                L.desc_memcpy_d2h(h.data_ptr(), send, 8 * h.numel())
                dist.all_reduce(h, group=self.group)
                mine = h.view(self.world, recvcount)[self.rank].contiguous()
                L.desc_memcpy_h2d(recv, mine.data_ptr(), 8 * recvcount)
                L.desc_device_synchronize(self.device)
                return 0
            except Exception:            # never raise through the C frame
                return 1

        def ag(send, recv, sendcount, dtype, comm, stream):
            try:
                L.desc_device_synchronize(self.device)
                mine = torch.empty(sendcount, dtype=torch.float64)
                L.desc_memcpy_d2h(mine.data_ptr(), send, 8 * sendcount)
                parts = [torch.empty(sendcount, dtype=torch.float64) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                full = torch.cat(parts)
                L.desc_memcpy_h2d(recv, full.data_ptr(), 8 * full.numel())
                L.desc_device_synchronize(self.device)
                return 0
            except Exception:
                return 1

        self.reduce_scatter = _lib.RS_FN(rs)
        self.all_gather = _lib.AG_FN(ag)

    def destroy(self):
        pass


class NativeShard:
    """One rank's share of the problem driven through the fused C protocol."""

    def __init__(self, prob, structure, device, rank, world, comm=None):
        self.solver = _lib.Solver(prob, structure, device, rank, world)
        self.info = self.solver.shard_info()
        self.comm = comm
        if comm is None:
            self.solver.shard_set_collectives()
        else:
            self.solver.shard_set_collectives(comm.comm, comm.reduce_scatter, comm.all_gather)

    def start(self, params): self.solver.shard_start(params)
    def iterate(self, n): self.solver.shard_iterate(n)
    def run(self, params): return self.solver.shard_run(params)
    def sync(self): self.solver.sync()
    def destroy(self): self.solver.destroy()


def init_distributed():
    """torchrun-style bootstrap: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    backend = os.environ.get("DESC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    device = local % max(ndev, 1)
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def shared_problem(name, rank, world, comm, generate):
    """The synthetic workload, generated ONCE: rank 0 runs the generator and parks the arrays in /dev/shm, the other ranks of the node map
    them (round 3 had every rank run the generator: 4.5 s of NumPy per rank at C4, all at the same time on the host's cores).
    Returns (nn, ii, jj, rij, err_vec)."""
    if world == 1:
        mo, nn, ii, jj, rij = generate(name)
        return nn, ii, jj, rij, mo.ErrVec
    base = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", f"desc_amd_bench_{os.environ.get('MASTER_PORT', '0')}_{name}")
    keys = ("ii", "jj", "rij", "err")
    if rank == 0:
        mo, nn, ii, jj, rij = generate(name)
        arrs = dict(ii=ii, jj=jj, rij=np.ascontiguousarray(rij).reshape(-1), err=np.asarray(mo.ErrVec, dtype=np.float64))
        for k in keys:
            np.save(f"{base}_{k}.npy", arrs[k])
        with open(f"{base}_n.txt", "w") as f:
            f.write(str(int(nn)))
    comm.barrier()
    if rank != 0:
        with open(f"{base}_n.txt") as f:
            nn = int(f.read())
        arrs = {k: np.load(f"{base}_{k}.npy", mmap_mode="r") for k in keys}
    out = (nn, np.ascontiguousarray(arrs["ii"]), np.ascontiguousarray(arrs["jj"]), np.ascontiguousarray(arrs["rij"]), np.ascontiguousarray(arrs["err"]))
    comm.barrier()                                      # every rank holds its copy: the files can go
    if rank == 0:
        for k in keys:
            try:
                os.remove(f"{base}_{k}.npy")
            except OSError:
                pass
        try:
            os.remove(f"{base}_n.txt")
        except OSError:
            pass
    return out


def _bench_one(name, args, rank, world, device, comm, describe, generate, native_comm):
    """K timed sharded iterations of one workload; returns the measurements (identical on every rank).
    native_comm: RcclComm (fused C protocol) or None (piecewise protocol over torch.distributed)."""
    import torch
    K, W = args.steps, args.warmup
    nn, ii, jj, rij, err_vec = shared_problem(name, rank, world, comm, generate)
    prob = _lib.ProblemArrays(nn, ii, jj, rij)
    t0 = time.perf_counter()
    try:
        st = _lib.Structure.build(prob, 30, args.seed, _lib.BUILD_DEVICE, device)
    except _lib.DescError as e:
        if e.code != _lib.ERR_TOO_LARGE:
            raise
        st = _lib.Structure.build(prob, 30, args.seed, _lib.BUILD_HOST, device)
    t_struct = time.perf_counter() - t0
    n_sample = st.sizes()["n_sample"]
    p = _lib.default_params()
    p.iters = W + K + 4
    p.lr = 0.01
    p.patience = (1 << 31) - 1          # the bench times exactly K sweeps: never stop early
    p.seed = args.seed
    t0 = time.perf_counter()
    if native_comm is not None or world == 1:
        shard = NativeShard(prob, st, device, rank, world, native_comm)
        t_create = time.perf_counter() - t0
        st.free()
        shard.start(p)
        shard.iterate(W)
        shard.sync(); torch.cuda.synchronize(); comm.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        shard.iterate(K)
        shard.sync(); torch.cuda.synchronize(); comm.barrier(); torch.cuda.synchronize()
        dt = comm.max_float(time.perf_counter() - t0)
        # objective of the last iterate + download through the piecewise calls of the same handle
        out = shard.solver.download()
        driver = "fused C protocol (two streams; RCCL entry points called from the library)"
    else:
        shard = HipShard(prob, st, device, rank, world)
        t_create = time.perf_counter() - t0
        st.free()
        drv = ShardedDriver(shard, comm)
        drv.start(p)
        drv.iterate(W)
        with shard.stream_ctx():
            torch.cuda.synchronize(); comm.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.iterate(K)
        with shard.stream_ctx():
            torch.cuda.synchronize(); comm.barrier(); torch.cuda.synchronize()
        dt = comm.max_float(time.perf_counter() - t0)
        out = drv.finish()
        driver = "piecewise protocol (torch.distributed collectives between ctypes calls)"
    info = shard.info
    lay = shard.solver.layout_stats()
    # this rank's share and what it exchanges per iteration: gathered to rank 0 for the line
    mine = dict(rank=rank, cycles=int(info.cyc_hi - info.cyc_lo), segments=int(info.seg_hi - info.seg_lo), pieces=lay.get("pieces", 0),
                sweep_kernel=shard.solver.last_sweep(),
                reduce_scatter_bytes_out=int(8 * info.t_part * (world - 1)), all_gather_bytes_out=int(8 * info.slice_len * (world - 1)))
    per_rank = [None] * world
    if world > 1:
        import torch.distributed as dist
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]
    res = dict(name=name, nn=nn, m=shard.solver.m, m_pos=info.m_pos, m_cycle=info.m_cycle, n_sample=int(n_sample), dt=dt,
               t_struct=t_struct, t_create=t_create, err=float(np.mean(np.abs(out["S_vec"] - err_vec))),
               workload=describe(name), driver=driver, per_rank=per_rank, S_vec=out["S_vec"], ii=ii, jj=jj, rij=rij, device=device)
    shard.destroy()
    return res


def _cpu_leg(r, args, cpu_baseline):
    """cpu_baseline + parity_vs_cpu for the N > 1 line (rank 0, after the timed region).  The oracle needs the cycle structure on the
    host: rank 0 builds it once more on its GPU and exports it (bench.py does the same at N = 1); the sharded S_vec it is compared with
    comes from a fresh one-GPU solve of the same iteration count -- what the ranks computed together is checked against the one-GPU
    run by tests/test_gpu_sharded.py, and against the truth by `mean_abs_err_vs_truth` above."""
    prob = _lib.ProblemArrays(r["nn"], r["ii"], r["jj"], r["rij"])
    st = _lib.Structure.build(prob, 30, args.seed, _lib.BUILD_DEVICE, r["device"])
    arrays = st.arrays()
    st.free()
    cb, ref, it = cpu_baseline(r["nn"], r["ii"], r["jj"], r["rij"], arrays, budget_s=12.0, max_iters=10 if r["name"] in ("C4", "C5") else 50)
    p = _lib.default_params()
    p.iters = it; p.lr = 0.01; p.seed = args.seed; p.patience = (1 << 31) - 1; p.device = r["device"]
    out = _lib.solve(prob, p)
    d = np.abs(out["S_vec"] - ref["S_vec"])
    return cb, {"iters": int(it), "mean_abs": float(d.mean()), "max_abs": float(d.max()),
                "what": "S_vec of a one-GPU HIP run vs oracle/desc_oracle.c (OpenMP) after the same iterations, same structure"}


def bench_sharded(args, WORKLOADS, describe, generate, cpu_baseline):
    """bench.py body for N > 1 ranks (started by bench.py itself or by torch.distributed.run, one rank per GPU).

    Same workload as at N = 1 (C4, BASELINE.json configs[3]; strong scaling: total work fixed, edges sharded over the
    ranks, reduce-scatter + all-gather per iteration); the default line also carries `secondary_config` = C2, which is
    small enough that one GPU is about as fast as any sharding."""
    import torch                                         # before libdesc_amd.so: see _lib.load()
    rank, world, device = init_distributed()
    comm = TorchComm()
    K, W = args.steps, args.warmup
    name = args.workload or "C4"
    native = None
    rccl_ranks, rccl_source = None, None
    if os.environ.get("DESC_SHARD_DRIVER", "native") == "native" and not comm.staged:
        native = RcclComm(rank, world, device)
        if not native.ok:
            if rank == 0:
                print(f"[desc_amd] native RCCL communicator unavailable ({native.error}); using the piecewise torch.distributed driver", file=sys.stderr, flush=True)
            native = None
        else:
            rccl_ranks, rccl_source = native.count, "ncclCommCount of the communicator the library's collectives run on"
    if native is None and not comm.staged:
        rccl_ranks, rccl_source = comm.world, "torch.distributed world size (backend nccl = RCCL)"
    r = _bench_one(name, args, rank, world, device, comm, describe, generate, native)
    extra = None
    if args.workload is None and not getattr(args, "no_secondary", False):
        x = _bench_one("C2", args, rank, world, device, comm, describe, generate, native)
        xb = 72.0 * x["m_cycle"] + 12.0 * x["m_pos"]
        extra = {"workload": x["workload"], "value": K / x["dt"], "unit": "iters/s", "ms_per_step": x["dt"] / K * 1e3,
                 "m_cycle": x["m_cycle"], "roofline_frac_of_aggregate_hbm": xb / (x["dt"] / K) / 1e9 / (8000.0 * world),
                 "setup_ms": {"structure": x["t_struct"] * 1e3, "create_shard": x["t_create"] * 1e3},
                 "mean_abs_err_vs_truth": x["err"]}
    dt = r["dt"]
    bytes_iter = 72.0 * r["m_cycle"] + 12.0 * r["m_pos"]
    line = {
        "metric": "DESC_PGD iters/sec", "value": K / dt, "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "rccl_ranks": rccl_ranks, "rccl_ranks_source": rccl_source,
        "config": {"workload": r["workload"], "n": r["nn"], "m": r["m"], "m_pos": r["m_pos"], "m_cycle": r["m_cycle"],
                   "n_sample": r["n_sample"], "sampling_seed": args.seed,
                   "parallelism": f"edges sharded over {world} GPUs by node ranges; per iteration: reduce-scatter of the mirror sums (2 m_pos int64 fixed-point words, "
                                  f"in exchange parts under the sweep) + all-gather of S; " + r["driver"]},
        "roofline": {"bound": "hbm", "achieved": bytes_iter / (dt / K) / 1e9, "peak": 8000.0 * world, "unit": "GB/s",
                     "frac": bytes_iter / (dt / K) / 1e9 / (8000.0 * world), "traffic": None,
                     "kernel": "whole iteration incl. collectives (aggregate over ranks)", "bytes_per_launch": bytes_iter},
        "cycle_updates_per_s": r["m_cycle"] * K / dt,
        "cpu_baseline": None,
        "per_rank": r["per_rank"],
        "setup_ms": {"structure": r["t_struct"] * 1e3, "create_shard": r["t_create"] * 1e3},
        "mean_abs_err_vs_truth": r["err"],
        "secondary_config": extra,
    }
    if rank == 0 and not getattr(args, "no_cpu_baseline", False):
        # The CPU data point of the SAME workload, AFTER the timed region, on rank 0's host cores (the other ranks wait at the barrier
        # below): the oracle's OpenMP restatement for a bounded number of iterations, and the GPU result of the same count next to it.
        try:
            cb, parity = _cpu_leg(r, args, cpu_baseline)
            line["cpu_baseline"] = cb
            line["gpu_over_cpu"] = line["value"] / cb["value"]
            line["parity_vs_cpu"] = parity
        except Exception as e:                           # test infrastructure: never lose the line over it
            line["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    import torch.distributed as dist
    dist.barrier()
    if native is not None:
        native.destroy()
    dist.destroy_process_group()
    return 0
