"""Host side of libdesc_amd under AddressSanitizer + UBSan and under ThreadSanitizer, without a GPU.

The library is compiled --offload-host-only and linked against tests/hipmock/hipmock.cpp (device memory = host heap, copies =
memcpy, kernels = no-ops), then tests/hipmock/drive.py replays the native call sequences of the GPU parity tests (host builder,
solver set-up with every layout forced, run / download into fenced caller buffers, shard planning, device-resident problem) in a
subprocess with the sanitizer runtime preloaded.  Background: gpurun_out/r2_tests3.log of round 2 (a NumPy array of the test
process was found modified after 32 passing GPU tests), DESIGN.md section 9."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipmock"))


@pytest.mark.parametrize("san", ["address", "thread"])
def test_host_side_is_clean_under_sanitizers(san):
    import build_host
    rt = build_host.runtime_lib(san)
    if rt is None:
        pytest.skip(f"no {san} sanitizer runtime in this toolchain")
    so = build_host.build(ROOT, san)
    env = dict(os.environ, LD_PRELOAD=rt, DESC_AMD_LIB=so, OPENBLAS_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0")
    if san == "thread":
        env["HOSTSAN_QUICK"] = "1"              # TSan slows the interpreter ~10x: the reduced case list still runs every threaded pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipmock", "drive.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    report = r.stdout[-3000:] + r.stderr[-6000:]
    assert r.returncode == 0, report
    assert "HOSTSAN OK" in r.stdout, report
    for marker in ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:"):
        assert marker not in r.stderr, report
