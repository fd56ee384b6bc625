"""CPU: host-side pieces of the spotfinder driver that need no GPU -- codecs, argument handling
and exit codes (spotfinder/spotfinder.cc:291-398, src/ffs/arg_parser.cc:72-77)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fast-feedback-service_amd", "bin")
SPOTFINDER = os.path.join(BIN, "spotfinder")
TOOL = os.path.join(BIN, "ffs_hosttool")

pytestmark = pytest.mark.skipif(not os.path.exists(SPOTFINDER), reason="run `make cli` first")


def run(*args, **kw):
    return subprocess.run([SPOTFINDER, *args], capture_output=True, text=True, timeout=60, **kw)


def test_codecs_selftest():
    p = subprocess.run([TOOL, "selftest"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "selftest ok" in p.stdout, p.stdout + p.stderr


def test_version_and_help_exit_zero():
    p = run("--version")
    assert p.returncode == 0 and not p.stderr
    p = run("--help")
    assert p.returncode == 0 and "Usage: spotfinder" in p.stdout


def test_list_devices_exits_zero():
    # the Zocalo service probes this at start-up (src/ffs/service.py:220-238)
    p = run("--list-devices")
    assert p.returncode == 0 and not p.stderr


@pytest.mark.parametrize("argv", [[], ["--images"], ["--threads", "x", "synth:tiny"], ["--bogus", "synth:tiny"],
                                  ["--sample", "synth:tiny"], ["a", "b"]])
def test_bad_arguments_print_usage_and_exit_one(argv):
    p = run(*argv)
    assert p.returncode == 1 and "Usage: spotfinder" in p.stdout and not p.stderr


def test_bad_algorithm_and_thread_count():
    assert run("synth:tiny", "-a", "nonsense").returncode == 1
    p = run("synth:tiny", "--threads", "0")
    assert p.returncode == 1 and "Thread count must be >= 1" in p.stdout


def test_spotfinder32_alias_exists():
    assert os.path.exists(os.path.join(BIN, "spotfinder32"))


def _fnv(data: bytes) -> int:
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("encoder", ["lz4", "literals"])
@pytest.mark.parametrize("shape,dtype", [((61, 97), "uint16"), ((31, 33), "uint32"), ((1, 7), "uint16")])
def test_numpy_chunk_writer_against_host_codec(tmp_path, encoder, shape, dtype):
    """The numpy/liblz4 chunk writer used for GPU decode tests produces what the host's bitshuffle-LZ4
    decoder (the restatement of the reference's read path) reads back bit for bit."""
    import numpy as np
    from ffs_amd import bslz4
    if encoder == "lz4" and bslz4.liblz4() is None:
        pytest.skip("no liblz4")
    rng = np.random.default_rng(5)
    img = rng.poisson(3.0, shape).astype(dtype)
    img[0, 0] = np.iinfo(dtype).max
    chunk = bslz4.compress(img, encoder)
    (tmp_path / "c.bin").write_bytes(chunk)
    p = subprocess.run([TOOL, "chunkfnv", str(tmp_path / "c.bin"), str(img.size), str(img.dtype.itemsize)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout
    assert p.stdout.split()[1] == "%016x" % _fnv(img.tobytes())
