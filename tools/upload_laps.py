"""Where the host -> HBM time of the rotation array goes (C4: 180 MB): the raw runtime copy of the same NumPy buffer, next to the library's paths."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from desc_amd import _lib    # noqa: E402

L = _lib.load()
hip = C.CDLL("libamdhip64.so")
m = 2_500_000
rng = np.random.default_rng(0)
rij = rng.standard_normal(9 * m)
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(rij.nbytes)) == 0
for lap in range(4):
    t0 = time.perf_counter(); rc = hip.hipMemcpy(d, C.c_void_p(rij.ctypes.data), C.c_size_t(rij.nbytes), 1); t1 = time.perf_counter()
    assert rc == 0
    print("raw hipMemcpy of the NumPy buffer (pageable, %d MB): %.2f ms" % (rij.nbytes >> 20, (t1 - t0) * 1e3), flush=True)
for lap in range(3):
    t0 = time.perf_counter(); L.desc_memcpy_h2d(d, C.c_void_p(rij.ctypes.data), C.c_size_t(rij.nbytes)); t1 = time.perf_counter()
    print("desc_memcpy_h2d: %.2f ms" % ((t1 - t0) * 1e3), flush=True)
fresh = rng.standard_normal(9 * m)
t0 = time.perf_counter(); hip.hipMemcpy(d, C.c_void_p(fresh.ctypes.data), C.c_size_t(fresh.nbytes), 1); t1 = time.perf_counter()
print("raw hipMemcpy of a buffer the runtime has not seen before: %.2f ms" % ((t1 - t0) * 1e3), flush=True)
for pin in ("1", "0"):
    os.environ["DESC_UPLOAD_PIN"] = pin
    ii = np.arange(m, dtype=np.int32) // 500; jj = ii + 1 + (np.arange(m, dtype=np.int32) % 500)
    prob = _lib.ProblemArrays(int(jj.max()) + 1, ii, jj, rij)
    for lap in range(3):
        t0 = time.perf_counter(); dp = _lib.DeviceProblem(prob, 0); t1 = time.perf_counter()
        dp.free()
        print("DESC_UPLOAD_PIN=%s  desc_problem_upload (rotations + edge list + CSR index): %.2f ms" % (pin, (t1 - t0) * 1e3), flush=True)
