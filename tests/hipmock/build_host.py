"""Test infrastructure: build the HOST side of libdesc_amd against the mock HIP runtime (hipmock.cpp) with a sanitizer.

    python tests/hipmock/build_host.py [--tree DIR] [--san address|thread|none] [--out PATH]

`--offload-host-only` compiles the host code of every .hip / .cpp source (kernels become launch stubs); the mock runtime
supplies the HIP entry points.  No GPU is needed to run the result."""
from __future__ import annotations

import argparse
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SAN = {"address": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
       "thread": ["-fsanitize=thread"], "none": []}


def build(tree=ROOT, san="address", out=None, verbose=False):
    out = out or os.path.join(HERE, f"libdesc_amd_host_{san}.so")
    csrc = os.path.join(tree, "desc_amd", "csrc")
    srcs = sorted(glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(HERE, "hipmock.cpp")]
    deps = srcs + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(tree, "include", "desc_amd.h"), __file__]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-Wno-unused-value", "-Wno-option-ignored",
             "-I", os.path.join(tree, "include"), "-I", csrc] + SAN[san]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, srcs))
    # every host-only object refers to its (absent) device code object through an external __hip_fatbin_<hash>
    undef = subprocess.check_output(["nm", "-u"] + objs, text=True)
    fat = sorted({ln.split()[-1] for ln in undef.splitlines() if "__hip_fatbin" in ln})
    stub = os.path.join(objdir, "fatbin_stub.c")
    with open(stub, "w") as f:
        for s in fat:
            f.write(f"const char {s}[8] = {{0}};\n")
    # hipcc would add -lamdhip64 at link time: link with the clang driver underneath instead
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
    link = [clang, "-shared", "-fPIC", "-pthread", "-o", out] + SAN[san] + (["-shared-libsan"] if san != "none" else []) + objs + ["-x", "c", stub]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


def runtime_lib(san):
    """Path of the sanitizer runtime to LD_PRELOAD into a Python process that loads the library."""
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    name = {"address": "libclang_rt.asan-x86_64.so", "thread": "libclang_rt.tsan-x86_64.so"}[san]
    p = subprocess.check_output([clang, f"-print-file-name={name}"], text=True).strip()
    return p if os.path.exists(p) else None


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", default=ROOT)
    ap.add_argument("--san", default="address")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    print(build(a.tree, a.san, a.out, verbose=True))
