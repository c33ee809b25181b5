"""Build recipe for libdesc_amd.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdesc_amd.so")

FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
         # the reference is plain IEEE double arithmetic; keep a*b+c unfused so the
         # device follows the same rounding sequence as the CPU restatement
         "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-pthread"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "desc_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP/C++ source of the package into desc_amd/libdesc_amd.so."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
