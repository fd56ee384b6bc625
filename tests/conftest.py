import os
import subprocess
import sys

import pytest

try:   # torch brings its own copy of the HIP runtime: it has to be in the process BEFORE libffs_hip.so pulls in /opt/rocm's (two
    import torch  # noqa: F401  runtimes in one process cannot both have the GPU; bench.py imports torch first for the same reason)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fast-feedback-service_amd")
sys.path.insert(0, os.path.join(PKG, "python"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # CPU-side libraries (oracle, synthetic generator) are cheap to build; the HIP library
    # is built by __graft_entry__.build() / `make hip` and only *loaded* here.
    need = [os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(PKG, "libffs_synth.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-C", ROOT, "oracle", "synth"], check=True,
                       stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def ffs():
    import ffs_amd
    return ffs_amd
