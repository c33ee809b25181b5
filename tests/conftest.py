import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# every output buffer the ctypes layer hands to the library is fenced by guard words, verified after each native call
# (desc_amd/_lib.py); must be set before desc_amd._lib is imported
os.environ.setdefault("DESC_DEBUG_GUARD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch bundles its own ROCm runtime under the same sonames as /opt/rocm
    # (libamdhip64.so.7, libhsa-runtime64.so.1).  One process can hold only one of them, and
    # torch does not work on top of the system copy -- so when torch will be used next to
    # libdesc_amd.so (the multi-GPU tests) it has to be imported first.
    expr = config.getoption("-m", default="") or ""
    if "gpu" in expr and "not gpu" not in expr:
        import torch  # noqa: F401


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def lib():
    """libdesc_amd.so.  Never rebuilt on the GPU box: the prebuilt .so travels with the tree."""
    from desc_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib
