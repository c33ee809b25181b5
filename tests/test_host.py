"""CPU tests (no GPU): host logic of the product -- structure builder against the oracle
and the golden fixtures, C-ABI surface, argument marshalling, error behaviour, models."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from desc_amd import ConstantStepSize, HybridGradient, PiecewiseStepSize
from desc_amd.algorithms import DESC_PGD, make_c_params, marshal_edges
from desc_amd.models import Nonuniform_Topology, Uniform_Topology
from tests.helpers import STRUCT_KEYS, assert_structure_equal, make_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def test_abi_exports_every_declared_symbol(lib):
    """libdesc_amd.so loads without a GPU and exports every function include/desc_amd.h declares."""
    hdr = open(os.path.join(ROOT, "include", "desc_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(desc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = lib.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in desc_amd.h but not exported"
    assert declared == set(lib.EXPORTS)
    assert L.desc_version().startswith(b"desc_amd")


def test_struct_layouts_match_header(lib):
    """ctypes mirrors of the ABI structs have the C sizes (x86-64 SysV)."""
    assert C.sizeof(lib.Problem) == 40
    assert C.sizeof(lib.Params) == 112
    assert C.sizeof(lib.Result) == 96
    assert C.sizeof(lib.StructureView) == 104
    assert C.sizeof(lib.StructureInfo) == 56
    assert C.sizeof(lib.RefineInfo) == 40
    p = lib.default_params()
    assert (p.iters, p.step_kind, p.patience, p.n_sample_min, p.build_where) == (100, 0, 30, 30, lib.BUILD_DEVICE)
    assert p.lr == 0.01 and p.stop_tol == 1e-5


def test_no_gpu_fails_loudly(lib):
    """Without a HIP device the solver must refuse -- there is no CPU fallback."""
    if lib.load().desc_device_count() > 0:
        pytest.skip("a GPU is visible")
    mo, nn, ii, jj, rij = make_problem("uniform", n=20, p=0.5, seed=1)
    prob = lib.ProblemArrays(nn, ii, jj, rij)
    st = lib.Structure.build(prob)
    with pytest.raises(lib.DescError):
        lib.Solver(prob, st, 0)
    with pytest.raises(lib.DescError):
        DESC_PGD(mo.Ind, mo.RijMat, dict(iters=3, Gradient=ConstantStepSize(0.01), verbose=False))
    with pytest.raises(lib.DescError):
        lib.DeviceProblem(prob, 0)                        # the device-resident problem has no host stand-in either
    from desc_amd import CEMP, DESC, Spectral
    for call in (lambda: DESC(mo.Ind, mo.RijMat, dict(iters=3, Gradient=ConstantStepSize(0.01), verbose=False)),
                 lambda: Spectral(mo.Ind, mo.RijMat), lambda: CEMP(mo.Ind, mo.RijMat, dict(max_iter=2, reweighting=[1.0], nsample=10))):
        with pytest.raises(lib.DescError):
            call()


@pytest.mark.parametrize("kind,n,p", [("uniform", 35, 0.5), ("uniform", 130, 0.55), ("uniform", 400, 0.1), ("nonuniform", 90, 0.4)])
def test_host_structure_equals_oracle(lib, oracle, kind, n, p):
    """a-1..a-3 (DESC_PGD.m:19-127): product host builder vs the oracle's sparse C builder --
    integer structure bit-exact (bitmap path of the product vs merge path of the oracle)."""
    mo, nn, ii, jj, rij = make_problem(kind, n=n, p=p, seed=3)
    for seed in (0, 12345):
        a = lib.Structure.build(lib.ProblemArrays(nn, ii, jj), 30, seed).arrays()
        b = oracle.build_structure(nn, ii, jj, seed=seed)
        assert_structure_equal(a, b)
        assert np.array_equal(a["codeg"], b["codeg"])
        assert a["max_cnt"] == int(np.diff(b["cum_ind"]).max())
    # invariants of the structure itself
    seg = np.repeat(np.arange(a["m_pos"]), np.diff(a["cum_ind"]))
    e = a["pos_edge"][seg]
    i, j, k = ii[e], jj[e], a["k"]
    assert ((k != i) & (k != j)).all()
    lo, hi = np.minimum(j, k), np.maximum(j, k)
    assert np.array_equal(ii[a["e_jk"]], lo) and np.array_equal(jj[a["e_jk"]], hi)
    lo, hi = np.minimum(i, k), np.maximum(i, k)
    assert np.array_equal(ii[a["e_ki"]], lo) and np.array_equal(jj[a["e_ki"]], hi)
    ok = a["ikj"] >= 0                       # IKJ(c) is the cycle of edge {i,k} through j
    assert np.array_equal(a["pos_edge"][seg[a["ikj"][ok]]], a["e_ki"][ok]) and np.array_equal(a["k"][a["ikj"][ok]], j[ok])
    ok = a["jki"] >= 0
    assert np.array_equal(a["pos_edge"][seg[a["jki"][ok]]], a["e_jk"][ok]) and np.array_equal(a["k"][a["jki"][ok]], i[ok])
    assert (np.diff(a["cum_ind"]) == np.minimum(a["codeg"][a["pos_edge"]], a["n_sample"])).all()


def test_sample_key_is_shared_definition(lib, oracle):
    L = lib.load()
    for s, e, k in [(0, 0, 0), (7, 123456, 99), (2**63 + 5, 2**31, 17)]:
        assert L.desc_sample_key(s, e, k) == oracle.sample_key(s, e, k) == oracle.lib().oracle_sample_key(s, e, k)


def test_structure_rejects_bad_input(lib):
    ii = np.array([0, 0, 1], dtype=np.int32); jj = np.array([1, 2, 2], dtype=np.int32)
    lib.Structure.build(lib.ProblemArrays(3, ii, jj)).free()
    with pytest.raises(lib.DescError, match="sorted"):
        lib.Structure.build(lib.ProblemArrays(3, ii[::-1].copy(), jj[::-1].copy()))
    with pytest.raises(lib.DescError, match="i < j"):
        lib.Structure.build(lib.ProblemArrays(3, jj, ii))
    with pytest.raises(lib.DescError):
        lib.Structure.build(lib.ProblemArrays(2, ii, jj))          # node id >= n
    with pytest.raises(lib.DescError):
        lib.Structure.build(lib.ProblemArrays(3, np.array([0, 0], dtype=np.int32), np.array([1, 1], dtype=np.int32)))  # duplicate


def test_large_edge_list_reports_the_first_offending_row(lib):
    """Edge lists of 2^20 rows and more are validated in chunks by several threads: the row reported is the first offending one, as in a
    sequential scan (two defects planted, in different chunks)."""
    n = 1600
    iu, ju = np.triu_indices(n, 1)
    ii = iu.astype(np.int32)[:1_200_000].copy(); jj = ju.astype(np.int32)[:1_200_000].copy()
    assert len(ii) >= 1 << 20
    ii[[900_000, 900_001]] = ii[[900_001, 900_000]]; jj[[900_000, 900_001]] = jj[[900_001, 900_000]]      # unsorted at row 900001 ...
    jj[300_000] = ii[300_000]                                                                             # ... and i == j at row 300000: reported
    with pytest.raises(lib.DescError, match=r"edge 300000 = .*i < j"):
        lib.Structure.build(lib.ProblemArrays(n, ii, jj))
    jj[300_000] = ju[300_000]
    with pytest.raises(lib.DescError, match="sorted by \\(i,j\\) at row 900001"):
        lib.Structure.build(lib.ProblemArrays(n, ii, jj))


def test_structure_edge_cases(lib):
    # tree: no triangles; median([]) = NaN -> n_sample = 30 (DESC_PGD.m:43)
    a = lib.Structure.build(lib.ProblemArrays(5, np.array([0, 1, 2, 2], dtype=np.int32), np.array([1, 2, 3, 4], dtype=np.int32))).arrays()
    assert a["m_pos"] == 0 and a["m_cycle"] == 0 and a["n_sample"] == 30 and (a["codeg"] == 0).all()
    # n_sample_min respected, even-length median, sampling when codeg == n_sample (>=)
    mo, nn, ii, jj, _ = make_problem("uniform", n=40, p=0.9, seed=2)
    a = lib.Structure.build(lib.ProblemArrays(nn, ii, jj), 5, 1).arrays()
    pos = a["codeg"][a["codeg"] > 0]
    assert a["n_sample"] == max(5, int(np.ceil(np.median(pos) / 4)))
    assert (np.diff(a["cum_ind"]) <= a["n_sample"]).all()


def test_structure_sizes_is_cheap_and_matches_view(lib):
    """desc_structure_sizes: the O(1) query bindings use to size per-cycle vectors (MEX shim, DESC_PGD())."""
    mo, nn, ii, jj, _ = make_problem("uniform", n=60, p=0.6, seed=3)
    st = lib.Structure.build(lib.ProblemArrays(nn, ii, jj), 30, 7)
    sz, a = st.sizes(), st.arrays()
    for key in ("n", "m", "m_pos", "m_cycle", "n_sample", "max_cnt"):
        assert sz[key] == a[key], key
    assert sz["built_where"] == lib.BUILD_HOST and sz["host_resident"] and sz["ms_build"] > 0
    assert lib.host_exports() == 0                  # nothing device-built in a CPU-only process
    with pytest.raises(lib.DescError) as ei:
        lib.check(lib.load().desc_structure_sizes(None, None))
    assert ei.value.code == lib.ERR_INVALID


def test_marshal_edges_sorts_and_unpermutes():
    mo = Uniform_Topology(25, 0.5, 0.2, 0.1, seed=4)
    perm = np.random.default_rng(0).permutation(mo.Ind.shape[0])
    n, ii, jj, rij, p = marshal_edges(mo.Ind[perm], mo.RijMat[:, :, perm])
    n0, ii0, jj0, rij0, p0 = marshal_edges(mo.Ind, mo.RijMat)
    assert p0 is None and p is not None
    assert np.array_equal(ii, ii0) and np.array_equal(jj, jj0) and np.array_equal(rij, rij0)
    assert np.array_equal(perm[p], np.arange(len(perm)))
    # MATLAB memory order of a 3x3xm array: element (r,c,l) at r + 3c + 9l
    l = 7
    assert rij0[9 * l + 1 + 3 * 2] == mo.RijMat[1, 2, l]
    with pytest.raises(ValueError):
        marshal_edges(mo.Ind[:, ::-1], mo.RijMat)               # i > j
    with pytest.raises(ValueError):
        marshal_edges(np.vstack([mo.Ind, mo.Ind[:1]]), None)    # duplicate edge
    with pytest.raises(ValueError):
        marshal_edges(mo.Ind, mo.RijMat[:, :, :-1])


def _marshal_reference(Ind, R):
    """What marshal_edges has to return for sorted input, stated with plain NumPy."""
    Ind = np.asarray(Ind)
    ii, jj = (Ind[:, 0] - 1).astype(np.int32), (Ind[:, 1] - 1).astype(np.int32)
    rij = np.stack([np.asarray(R)[r, c, :] for c in range(3) for r in range(3)], axis=1).reshape(-1)      # r + 3c + 9l
    return int(Ind.max()), ii, jj, rij


@pytest.mark.parametrize("dtype", [np.float64, np.int64, np.int32, np.float32, np.uint16])
def test_marshal_natives_formats_and_strides(lib, dtype):
    """desc_marshal_edges / desc_marshal_rij (the caller's side of DESC_PGD.m:14): every element type and memory order a NumPy or MATLAB
    caller may hold gives the same ABI arrays."""
    mo = Uniform_Topology(30, 0.5, 0.2, 0.1, seed=9)
    m = mo.Ind.shape[0]
    n0, ii0, jj0, rij0 = _marshal_reference(mo.Ind, mo.RijMat)
    wide = np.zeros((m, 6), dtype=dtype); wide[:, 1] = mo.Ind[:, 0]; wide[:, 4] = mo.Ind[:, 1]
    big = np.zeros((3, 3, 2 * m)); big[:, :, ::2] = mo.RijMat
    for Ind in (mo.Ind.astype(dtype), np.asfortranarray(mo.Ind.astype(dtype)), wide[:, 1::3]):
        for R in (np.ascontiguousarray(mo.RijMat), np.asfortranarray(mo.RijMat), big[:, :, ::2], np.ascontiguousarray(mo.RijMat.transpose(2, 0, 1)).transpose(1, 2, 0)):
            n, ii, jj, rij, perm = marshal_edges(Ind, R)
            assert perm is None and n == n0 and ii.dtype == np.int32 and jj.dtype == np.int32
            assert np.array_equal(ii, ii0) and np.array_equal(jj, jj0) and np.array_equal(rij, rij0)
    # reversed rows: unsorted, the permutation brings the caller's order back
    n, ii, jj, rij, perm = marshal_edges(mo.Ind[::-1].astype(dtype), mo.RijMat[:, :, ::-1])
    assert np.array_equal(ii, ii0) and np.array_equal(jj, jj0) and np.array_equal(rij, rij0) and np.array_equal(perm, np.arange(m)[::-1])
    # the natives themselves, as the header documents them
    nn, i2, j2, srt = lib.marshal_edges_native(mo.Ind.astype(np.float64))
    assert (nn, srt) == (n0, True) and np.array_equal(i2, ii0) and np.array_equal(j2, jj0)
    assert lib.marshal_edges_native(mo.Ind[::-1].astype(np.int64))[3] is False
    assert lib.marshal_edges_native(np.zeros((0, 2)))[0] == 0
    pp = np.random.default_rng(1).permutation(m)
    assert np.array_equal(lib.marshal_rij_native(np.ascontiguousarray(mo.RijMat), pp).reshape(m, 9), rij0.reshape(m, 9)[pp])


def test_marshal_edges_rejects_what_the_reference_cannot_index():
    mo = Uniform_Topology(25, 0.5, 0.2, 0.1, seed=4)
    Ind = mo.Ind.astype(np.float64)
    bad_row = 11
    for spoil, text in ((0.5, "integer"), (np.nan, "integer"), (np.inf, "integer"), (-3.0, "1-based"), (0.0, "1-based")):
        bad = Ind.copy(); bad[bad_row, 0] = bad[bad_row, 0] + spoil if spoil == 0.5 else spoil
        with pytest.raises(ValueError, match=text) as ei:
            marshal_edges(bad, None)
        assert "row %d" % bad_row in str(ei.value)
    loop = Ind.copy(); loop[3, 1] = loop[3, 0]
    with pytest.raises(ValueError, match="1-based"):
        marshal_edges(loop, None)                                   # i == j
    with pytest.raises(ValueError, match="integer"):
        marshal_edges(np.array([["a", "b"]]), None)
    with pytest.raises(ValueError):
        marshal_edges(np.vstack([Ind[::-1], Ind[:1]]), None)        # unsorted AND a duplicate


def test_params_translation():
    p, G = make_c_params(dict(iters=7, Gradient=ConstantStepSize(0.5), learning_rate=123.0))
    assert (p.iters, p.step_kind, p.lr) == (7, 0, 0.5)          # learning_rate is never read (DESC_PGD.m:169)
    g = PiecewiseStepSize(0.1, 25); g.t = 4
    p, _ = make_c_params(dict(iters=3, Gradient=g))
    assert (p.step_kind, p.lr, p.decay_interval, p.t0) == (1, 0.1, 25.0, 4)
    g = HybridGradient(0.01, 0.9, 0.99, 10); g.stopAdam()
    p, _ = make_c_params(dict(iters=3, Gradient=g))
    assert (p.step_kind, p.hybrid_strategy, p.beta2) == (2, 1, 0.99)
    with pytest.raises(TypeError):
        make_c_params(dict(iters=3, Gradient=object()))
    with pytest.raises(ValueError):
        make_c_params(dict(Gradient=ConstantStepSize(1)))


def test_step_plugins_match_reference_formulas():
    g = np.array([1.0, -2.0, 0.5])
    assert np.allclose(ConstantStepSize(0.1).GetStep(g), -0.1 * g)
    P = PiecewiseStepSize(1.0, 2)
    steps = [P.GetStep(g)[0] for _ in range(5)]                  # t=1..5: 1/(fix(t/2)+1)
    assert np.allclose(steps, [-1, -1 / 2, -1 / 2, -1 / 3, -1 / 3])
    H = HybridGradient(0.1, 0.9, 0.999, 10)
    s1 = H.GetStep(g)                                            # first Adam step = -lr*g/(|g|+1e-8)
    assert np.allclose(s1, -0.1 * g / (np.abs(g) + 1e-8))
    H.stopAdam(); s2 = H.GetStep(g)                              # t=2: 100*lr/(fix(2/10)+1)
    assert np.allclose(s2, -10.0 * g) and H.t == 2


@pytest.mark.parametrize("gen", [lambda: Uniform_Topology(60, 0.5, 0.3, 0.1, "uniform", seed=1),
                                 lambda: Uniform_Topology(60, 0.5, 0.3, 0.1, "self-consistent", seed=2),
                                 lambda: Nonuniform_Topology(60, 0.5, 0.5, 0.5, 0.1, 0.1, "self-consistent", seed=3),
                                 lambda: Nonuniform_Topology(60, 0.5, 0.5, 0.5, 0.1, 0.1, "adv", seed=4)])
def test_models_produce_valid_problems(gen):
    mo = gen()
    Ind, R = mo.Ind, mo.RijMat
    m = Ind.shape[0]
    assert (Ind[:, 0] < Ind[:, 1]).all() and Ind.min() >= 1
    key = Ind[:, 0] * 1000 + Ind[:, 1]
    assert (np.diff(key) > 0).all()                              # sorted (1,2),(1,3),...,(2,3),...
    Rm = np.transpose(R, (2, 0, 1))
    assert np.abs(Rm @ np.transpose(Rm, (0, 2, 1)) - np.eye(3)).max() < 1e-12
    assert np.abs(np.linalg.det(Rm) - 1).max() < 1e-12
    assert mo.ErrVec.shape == (m,) and (mo.ErrVec >= 0).all() and (mo.ErrVec <= 1).all()
    good = ~mo.corrupted
    assert mo.ErrVec[good].mean() < mo.ErrVec[mo.corrupted].mean()
    A = mo.AdjMat
    assert A.shape == (60, 60) and (A == A.T).all() and A.sum() == 2 * m
    # uniform corruption fraction is about q
    if hasattr(mo, "corrupted") and "Uniform" in gen.__code__.co_consts.__repr__():
        pass


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_structure_and_c_oracle(lib, oracle, path):
    """Golden fixtures (literal NumPy restatement) vs the product's host structure builder
    and vs the sparse C oracle."""
    g = np.load(path)
    n, ii, jj, rij, perm = marshal_edges(g["Ind"], g["RijMat"])
    assert perm is None
    seed = int(g["sampling_seed"])
    a = lib.Structure.build(lib.ProblemArrays(n, ii, jj), 30, seed).arrays()
    assert a["n_sample"] == int(g["n_sample"])
    for key in STRUCT_KEYS:
        assert np.array_equal(a[key], g[key]), key
    st = oracle.build_structure(n, ii, jj, seed=seed)
    S0 = oracle.cycle_d(ii, jj, rij.reshape(-1, 9), st)
    assert np.abs(S0 - g["S0_long"]).max() < 1e-15
    step = g["step"]; kind = int(g["step_kind"])
    kw = dict(lr=step[0])
    if kind == 1:
        kw.update(step_kind=1, decay_interval=step[1])
    if kind == 2:
        kw.update(step_kind=2, beta1=step[1], beta2=step[2], decay_interval=step[3])
    res = oracle.pgd_run(st, S0, int(g["iters"]), **kw)
    assert res["iters_run"] == int(g["iters_run"])
    assert np.abs(res["S_vec"] - g["S_vec"]).max() < 1e-12
    assert np.abs(res["w"] - g["wijk"]).max() < 1e-12
    assert np.allclose(res["obj"], g["obj_vals"], rtol=1e-13)
    assert np.allclose(res["avg"], g["avg_changes"], rtol=1e-10, atol=1e-16)


def _build_c_example(tmp_path):
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "desc_example")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(root, "include", "desc_amd.h")])
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "desc_pgd_example.c"),
                           "-L", os.path.join(root, "desc_amd"), "-ldesc_amd", "-lm", "-o", exe])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "desc_amd") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    return subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)


def test_header_is_c99_and_c_client_fails_loudly_without_gpu(lib, tmp_path):
    """include/desc_amd.h is plain C; a gcc-built client links against the library and, with no GPU,
    gets an error code and a message -- never a silent CPU result."""
    try:
        ndev = lib.device_count()
    except lib.DescError:
        ndev = 0
    if ndev > 0:
        pytest.skip("a GPU is visible: see tests/test_gpu_parity.py::test_c_client")
    r = _build_c_example(tmp_path)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.parametrize("shim", ["desc_pgd_mex.c", "desc_amd_mex.c"])
def test_mex_shims_are_syntactically_valid_c(shim):
    """The MEX shims cannot be built without MATLAB; they are at least syntax- and type-checked against the
    documented prototypes of the MEX / C Matrix API functions they call (tests/mock_mex/mex.h, a test mock)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(root, "tests", "mock_mex"), "-I", os.path.join(root, "include"), os.path.join(root, "matlab", shim)])


@pytest.mark.parametrize("kind,n,p,world,grid,jm", [("uniform", 300, 0.5, 1, 256, "0"), ("uniform", 300, 0.5, 1, 256, "1"), ("uniform", 900, 0.12, 3, 64, "1"),
                                                    ("nonuniform", 400, 0.3, 2, 256, "0"), ("uniform", 1500, 0.05, 1, 17, "1"), ("uniform", 60, 0.9, 1, 8, "1")])
def test_band_sweep_work_plan_invariants(lib, kind, n, p, world, grid, jm, monkeypatch):
    """Host-side plan of the band sweep (desc_debug_band_plan: bands sized for the LDS, ranks' ranges, pieces in contiguous
    or j-block-major mode): every owned segment lies in exactly one piece, every piece inside one band with that band's rows,
    and the per-workgroup cycle counts are balanced.  Runs without a GPU."""
    monkeypatch.setenv("DESC_DEBUG_JMAJOR", jm)
    monkeypatch.setenv("DESC_DEBUG_JBLOCK", "40")
    mo, nn, ii, jj, _ = make_problem(kind, n=n, p=p, seed=n)
    prob = lib.ProblemArrays(nn, ii, jj)
    st = lib.Structure.build(prob, 30, 3)
    covered = 0
    for rank in range(world):
        stats = np.zeros(8, dtype=np.int64)
        lib.check(lib.load().desc_debug_band_plan(C.byref(prob.c), st.handle, world, rank, grid, lib.ptr(stats, lib.I64P)))
        bands, pieces, rows, wmax, wmin, jmajor, seg_lo, seg_hi = stats
        assert bands >= 1 and pieces >= 1 and 0 < rows <= 19200 and jmajor == int(jm)
        assert seg_lo == covered
        covered = seg_hi
        if pieces >= 4 * grid or not jmajor:                      # enough units to balance (forced j-block mode on a tiny graph has few)
            assert wmax <= 2.0 * max(wmin, 1) + 16384, (wmax, wmin)  # list scheduling / equal-cycle cuts: no starved or overloaded workgroup
    assert covered == st.sizes()["m_pos"]
