"""Test infrastructure: drives the host side of libdesc_amd (built against hipmock.cpp) through the call sequences of the GPU parity
tests -- host structure builder, solver set-up with every layout forced (DESC_DEBUG_VARIANT 3/2/1), run + download into fenced
caller buffers, device-resident problem, one-shot solve -- so that AddressSanitizer / UBSan / ThreadSanitizer see every host-side
write.  Kernels do not run under the mock: values are not checked here, memory behaviour is.

Run by tests/test_host_sanitizers.py in a subprocess with the sanitizer runtime preloaded."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DESC_DEBUG_GUARD"] = "1"
os.environ["DESC_CACHE_MB"] = "0"        # every "device" block fresh from calloc: what a kernel would have produced reads back as zeros
os.environ["DESC_DEBUG_OVERLAP_UPLOAD"] = "2"      # desc_pgd_solve: the helper-thread upload of the rotations also on these small graphs

import numpy as np  # noqa: E402

from desc_amd import _lib as lib  # noqa: E402
from tests.helpers import c_params, make_problem  # noqa: E402

VAR = {"band": "3", "node": "2", "gather": "1"}
QUICK = os.environ.get("HOSTSAN_QUICK") == "1"


def solve_once(nn, ii, jj, rij, p, variant, where, adam=False):
    os.environ["DESC_DEBUG_VARIANT"] = VAR[variant]
    try:
        prob = lib.ProblemArrays(nn, ii, jj, rij)
        st = lib.Structure.build(prob, 30, p.seed, where, 0)
        arrays = st.arrays()
        solver = lib.Solver(prob, st, 0)
        solver.s0()
        ad = (np.zeros(solver.m_cycle), np.zeros(solver.m_cycle)) if adam else None
        solver.run(p, want_w=True, adam=ad)
        solver.destroy(); st.free()
    finally:
        os.environ.pop("DESC_DEBUG_VARIANT", None)
    return arrays


def main():
    L = lib.load()
    assert hasattr(L, "hipmock_launch_count"), "this driver must run against the mock build"
    cases = [(12, 0.6, 9), (60, 0.3, 2), (260, 0.5, 5)] if QUICK else [(12, 0.6, 9), (30, 0.5, 1), (60, 0.3, 2), (120, 0.6, 3), (200, 0.5, 4), (260, 0.5, 5)]
    for n, p, seed in cases:                                     # tests/test_gpu_parity.py::test_uniform_constant_step
        mo, nn, ii, jj, rij = make_problem("uniform", n=n, p=p, q=0.2, sigma=0.1, seed=seed)
        ref = None
        for variant in ("band", "node", "gather"):
            a = solve_once(nn, ii, jj, rij, c_params(100, lr=0.01, seed=11), variant, lib.BUILD_HOST)
            if ref is None:
                ref = a
            for k in ("pos_edge", "cum_ind", "k", "e_jk", "e_ki", "ikj", "jki"):
                assert np.array_equal(a[k], ref[k]), k
        # the generator must keep returning the same sorted edge list (gpurun_out/r2_tests3.log: it once did not)
        mo2, nn2, ii2, jj2, rij2 = make_problem("uniform", n=n, p=p, q=0.2, sigma=0.1, seed=seed)
        assert np.array_equal(ii, ii2) and np.array_equal(jj, jj2) and np.array_equal(rij, rij2)
        print("ok uniform", n, flush=True)
    mo, nn, ii, jj, rij = make_problem("nonuniform", n=150, p=0.4, seed=6, crpt_type="self-consistent")
    for variant in ("band", "node", "gather"):
        solve_once(nn, ii, jj, rij, c_params(60, lr=0.01, seed=2), variant, lib.BUILD_HOST)
        for kind, hs in ((1, 0), (2, 0), (2, 1)):
            pp = c_params(20, step_kind=kind, lr=0.01, seed=3, decay_interval=10, hybrid_strategy=hs)
            solve_once(nn, ii, jj, rij, pp, variant, lib.BUILD_HOST, adam=(kind == 2 and hs == 0))
    print("ok nonuniform + plugins", flush=True)
    # long segments (> 64 cycles) and the multithreaded CSR / compaction passes (m >= 2^18)
    mo, nn, ii, jj, rij = make_problem("uniform", n=310, p=0.95, q=0.2, sigma=0.1, seed=7)
    for variant in ("band", "gather"):
        solve_once(nn, ii, jj, rij, c_params(5, lr=0.01, seed=3), variant, lib.BUILD_HOST)
    if not QUICK:
        mo, nn, ii, jj, rij = make_problem("uniform", n=1200, p=0.45, q=0.2, sigma=0.1, seed=3)
        # (BUILD_DEVICE makes the CSR index and the cycles with kernels since round 3: nothing for the mock to run -- the host builder only)
        for variant in ("band", "node"):
            solve_once(nn, ii, jj, rij, c_params(3, lr=0.01, seed=5), variant, lib.BUILD_HOST)
        dp = lib.DeviceProblem(lib.ProblemArrays(nn, ii, jj, rij)); dp.free()
    print("ok long segments / threads", flush=True)
    # the marshalling natives on strided views, multithreaded sizes, with and without a permutation; error rows inside a later thread's range
    from desc_amd.algorithms import marshal_edges
    big = 700000
    e = np.arange(big)
    Ind = np.stack([e // 3 + 1, e // 3 + 2 + e % 3], axis=1)
    wide = np.zeros((big, 5)); wide[:, 1] = Ind[:, 0]; wide[:, 3] = Ind[:, 1]
    R = np.random.default_rng(0).standard_normal((3, 3, big))
    n1, i1, j1, r1, p1 = marshal_edges(wide[:, 1::2], R)
    n2, i2, j2, r2, p2 = marshal_edges(Ind[::-1].astype(np.int32), R[:, :, ::-1])
    assert p1 is None and np.array_equal(i1, i2) and np.array_equal(j1, j2) and np.array_equal(r1, r2) and r1[9 * 5 + 1 + 3 * 2] == R[1, 2, 5]
    bad = wide.copy(); bad[big - 7, 1] = np.nan
    try:
        marshal_edges(bad[:, 1::2], None); raise AssertionError("NaN accepted")
    except ValueError as e:
        assert "row %d" % (big - 7) in str(e)
    print("ok marshalling", flush=True)
    # device-resident problem + one-shot solve + the next rows' host sides
    mo, nn, ii, jj, rij = make_problem("uniform", n=90, p=0.5, q=0.2, sigma=0.1, seed=8)
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    dp = lib.DeviceProblem(prob)
    st = lib.Structure.build(prob, 30, 1, lib.BUILD_HOST, 0)
    s = lib.Solver(dp, st, 0); s.run(c_params(7, seed=1)); s.destroy(); st.free()
    ph = c_params(9, seed=1); ph.build_where = lib.BUILD_HOST
    lib.solve(prob, ph)
    lib.cemp_run(dp, [1.0, 2.0], 2, 20, seed=3)
    dp.free()
    # regression: a handle re-armed with a smaller budget must not be iterated past it (the device trace buffers keep the
    # larger size, the caller's trace buffers have the smaller one: desc_pgd_download would write past them)
    st = lib.Structure.build(prob, 30, 1, lib.BUILD_HOST, 0)
    s = lib.Solver(prob, st, 0)
    s.reset(c_params(100, seed=1)); s.iterate(100); s.download()
    s.reset(c_params(5, seed=1)); s.iterate(5)
    try:
        s.iterate(1)
    except lib.DescError as e:
        assert e.code == lib.ERR_INVALID
    else:
        raise AssertionError("iterating past params.iters of the last reset must fail")
    out = s.download(want_w=True)
    assert out["iters_run"] == 5 and out["obj"].shape == (5,)
    s.destroy(); st.free()
    # shard planning of the multi-GPU path (host side: ranges, exchange layout), world 2 and 3
    for where in (lib.BUILD_HOST,):          # (a device-built structure has no cycles under the mock: nothing to shard)
        st = lib.Structure.build(prob, 30, 1, where, 0)
        for world in (2, 3):
            for rank in range(world):
                s = lib.Solver(prob, st, 0, rank, world)
                info = s.shard_info()
                assert info.world == world and info.rank == rank
                s.shard_bind(None, None, None)
                s.reset(c_params(3, seed=1)); s.shard_finish(1); s.shard_colsum(); s.shard_sweep(); s.shard_finish(0)
                s.destroy()
        st.free()
    # round 4: the band sweep forced on a graph cut into many bands -- exchange parts (virtual owners, per-part piece plans), the ranks' column-sum
    # node lists, and the fused two-stream protocol with the reduce-scatter in parts (collectives: callbacks that do nothing)
    rs_calls, ag_calls = [], []
    rs_cb = lib.RS_FN(lambda send, recv, count, dtype, op, comm, stream: rs_calls.append((count, dtype)) or 0)
    ag_cb = lib.AG_FN(lambda send, recv, count, dtype, comm, stream: ag_calls.append((count, dtype)) or 0)
    os.environ["DESC_DEBUG_VARIANT"] = "3"; os.environ["DESC_DEBUG_ROW_CAP"] = "200"
    try:
        mo, nn, ii, jj, rij = make_problem("uniform", n=120, p=0.5, q=0.2, sigma=0.1, seed=4)
        prob2 = lib.ProblemArrays(nn, ii, jj, rij)
        st = lib.Structure.build(prob2, 30, 1, lib.BUILD_HOST, 0)
        for world, parts in ((3, "2"), (8, "2"), (3, "3"), (2, "1")):
            os.environ["DESC_SHARD_PARTS"] = parts
            for rank in range(world):
                s = lib.Solver(prob2, st, 0, rank, world)
                info = s.shard_info()
                assert info.xparts == int(parts) and info.t_len == world * info.xparts * info.t_part + 1
                s.shard_layout()
                s.shard_set_collectives(None, rs_cb, ag_cb)
                n_rs = len(rs_calls)
                s.shard_start(c_params(4, seed=1)); s.shard_iterate(3); s.sync(); s.stopped()
                assert len(rs_calls) - n_rs == 3 * int(parts) and all(c == (info.t_part, 4) for c in rs_calls[n_rs:])      # one int64 reduce-scatter per part and iteration
                s.download()
                s.destroy()
        st.free()
    finally:
        for k in ("DESC_DEBUG_VARIANT", "DESC_DEBUG_ROW_CAP", "DESC_SHARD_PARTS"):
            os.environ.pop(k, None)
    print("ok exchange parts / fused protocol", flush=True)
    # several host threads at once, each with its own problem (tests/test_gpu_parity.py::test_concurrent_solves_from_several_host_threads):
    # what ThreadSanitizer is here for -- the block / stream pools, the upload-order counter, the per-thread error text
    import threading
    probs = []
    for n, p, seed in ((40, 0.5, 1), (70, 0.4, 2), (90, 0.5, 3)):
        mo, nn, ii, jj, rij = make_problem("uniform", n=n, p=p, q=0.2, sigma=0.1, seed=seed)
        probs.append(lib.ProblemArrays(nn, ii, jj, rij))
    failed = []

    def worker(t):
        try:
            for rep in range(3):
                pr = probs[(t + rep) % len(probs)]
                pp = c_params(6, seed=1); pp.build_where = lib.BUILD_HOST
                if (t + rep) % 2:
                    lib.solve(pr, pp)
                else:
                    st_ = lib.Structure.build(pr, 30, 1, lib.BUILD_HOST, 0)
                    s_ = lib.Solver(pr, st_, 0); s_.run(pp); s_.destroy(); st_.free()
                try:
                    lib.Structure.build(lib.ProblemArrays(3, np.array([0, 0], dtype=np.int32), np.array([1, 1], dtype=np.int32)))
                except lib.DescError as e:
                    assert "sorted" in str(e) or "edge" in str(e), str(e)          # this thread's own error text
        except Exception as e:          # noqa: BLE001
            failed.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not failed, failed
    print("ok concurrent host threads", flush=True)
    lib.verify_guards()
    lib.trim_memory()
    assert L.hipmock_live_blocks() == 0, "device blocks leaked"
    print("HOSTSAN OK launches", L.hipmock_launch_count(), flush=True)


if __name__ == "__main__":
    main()
