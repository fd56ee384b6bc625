"""Handle lifetime through the C ABI (include/ffs_hip.h "Lifetime rules"): a process may let go of contexts,
streams and stacks in any order, destroy them twice, or exit with them alive and with batches in flight -- exit code 0 and nothing on
stderr.  Each case is ONE fresh child process (never a re-exec of a process that holds the GPU), run once.
Reference: the worker's RAII members and CUDA_CHECK's exceptions, spotfinder/spotfinder.cc:729-742, include/cuda_common.hpp:28-45."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_child(*argv, timeout=300):
    env = dict(os.environ)
    env.pop("FFS_HIP_LIB", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "lifecycle_child.py"), *argv], capture_output=True, text=True,
                       timeout=timeout, env=env)
    # (torch's bundled libdrm looks for a device-name table this image does not ship and says so on stderr: not ours)
    err = "".join(l for l in p.stderr.splitlines(True) if "libdrm/amdgpu.ids" not in l)
    return p.returncode, p.stdout, err


@pytest.mark.parametrize("mode", ["leak", "leak_hard", "inflight", "ctx_first", "reverse_gc"])
def test_child_process_leaves_cleanly(mode):
    rc, out, err = run_child(mode)
    assert rc == 0 and err == "", (rc, out[-400:], err[-2000:])
    assert out.startswith("ok ")


@pytest.mark.parametrize("mode", ["leak", "ctx_first"])
def test_child_process_leaves_cleanly_with_torch_first(mode):
    pytest.importorskip("torch")
    rc, out, err = run_child(mode, "torch")
    assert rc == 0 and err == "", (rc, out[-400:], err[-2000:])
