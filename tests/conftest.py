"""pytest configuration: `gpu` marker, locations of the oracle and of the reference binary."""
import os
import subprocess
import sys

import pytest

try:        # Load torch's bundled HIP runtime BEFORE librcgpu.so pulls in /opt/rocm's: with the opposite order torch's lazy
    import torch  # noqa: F401   # device init later reports "No HIP GPUs are available" (two runtimes, one SONAME).
except Exception:  # pragma: no cover - CPU-only environments without torch still run the host tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure librcgpu.so and liboracle.so exist (builds them when the toolchain is here)."""
    lib = os.path.join(ROOT, "rawcooked_amd", "librcgpu.so")
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    shim = os.path.join(ROOT, "rawcooked_amd", "rcgpu-ffmpeg")
    if not (os.path.exists(lib) and os.path.exists(ora) and os.path.exists(shim)):
        import __graft_entry__
        __graft_entry__.build()
    return lib, ora


@pytest.fixture(scope="session")
def refbin():
    """oracle/_ref/rawcooked: the real reference, built by oracle/Makefile.ref where /root/reference exists."""
    p = os.path.join(ROOT, "oracle", "_ref", "rawcooked")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/rawcooked not built (needs /root/reference)")
    return p


@pytest.fixture(scope="session")
def linkedbin():
    """oracle/_ref/rawcooked_linked: the reference with librcgpu.so linked in through the two patches of INTEGRATION.md routes B and C
    (oracle/Makefile.ref, target `linked`)."""
    p = os.path.join(ROOT, "oracle", "_ref", "rawcooked_linked")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/rawcooked_linked not built (needs /root/reference)")
    return p
