"""CPU: the NXmx/HDF5 frame source (host/h5_reader.cc) against fixtures written by
`ffs_hosttool mkh5` in the three layouts the reference's reader resolves
(h5read/src/h5read.c:905-990), metadata paths (:795-900), and frame availability for a
collection still being written (:379-420).  The LZ4 block codec is cross-checked against the
system liblz4 when one is present."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fast-feedback-service_amd", "bin")
TOOL = os.path.join(BIN, "ffs_hosttool")


def tool(*args):
    p = subprocess.run([TOOL, *args], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def h5_enabled():
    return os.path.exists(TOOL) and subprocess.run([TOOL, "h5support"], capture_output=True, text=True).stdout.strip() == "1"


pytestmark = pytest.mark.skipif(not h5_enabled(), reason="needs `make cli` with HDF5 headers available")


def frame_lines(text):
    return [l for l in text.splitlines() if l.startswith("frame")]


@pytest.mark.parametrize("layout", ["vds-links", "vds-files", "plain"])
def test_layouts_round_trip_pixels(tmp_path, layout):
    spec = "synth:tiny:6"
    tool("mkh5", spec, str(tmp_path / "x_master.h5"), layout, "4")
    info = tool("h5info", str(tmp_path / "x_master.h5"))
    want = [l.split()[-1] for l in frame_lines(tool("synthinfo", spec))]
    got = [l.split()[-1] for l in frame_lines(info)]
    assert got == want and len(got) == 6
    assert info.splitlines()[0] == "images 6 shape 200 300 bytes 2 trusted 0 65535"
    assert "mask valid 60000" in info
    if layout != "plain":
        assert len(glob.glob(str(tmp_path / "x_0000*.h5"))) == 2


def test_metadata_and_oscillation(tmp_path):
    tool("mkh5", "synth:tinysweep:5", str(tmp_path / "s_master.h5"))
    info = tool("h5info", str(tmp_path / "s_master.h5")).splitlines()
    assert info[1] == "wavelength 0.976 distance 0.3 pixel 7.5e-05 7.5e-05 beam 100 150 osc 0 0.1"


@pytest.mark.parametrize("per_file,n_written,avail", [(4, 3, 3), (2, 2, 2), (3, 0, 0)])
def test_partially_written_collection(tmp_path, per_file, n_written, avail):
    """Frames without a chunk, and data files that do not exist yet, read as 'not available'."""
    tool("mkh5", "synth:tiny:6", str(tmp_path / "p_master.h5"), "vds-links", str(per_file), str(n_written))
    lines = frame_lines(tool("h5info", str(tmp_path / "p_master.h5")))
    assert [("unavailable" not in l) for l in lines] == [i < avail for i in range(6)]


def test_32bit_pixels(tmp_path):
    tool("mkh5", "synth:jungfrau9m:1", str(tmp_path / "j_master.h5"), "vds-files")
    info = tool("h5info", str(tmp_path / "j_master.h5"))
    assert info.splitlines()[0].startswith("images 1 shape 3072 3072 bytes 4")
    assert frame_lines(info)[0].split()[-1] == frame_lines(tool("synthinfo", "synth:jungfrau9m:1"))[0].split()[-1]


def test_lz4_against_system_liblz4(tmp_path):
    """Our LZ4 block decoder reads what liblz4 writes and vice versa (through an SHM fixture)."""
    cands = glob.glob("/opt/conda/lib/liblz4.so*") + glob.glob("/usr/lib/x86_64-linux-gnu/liblz4.so*")
    if not cands:
        pytest.skip("no liblz4 on this machine")
    lz4 = ctypes.CDLL(cands[0])
    lz4.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    tool("mkshm", "synth:tiny:1", str(tmp_path / "shm"))
    chunk = (tmp_path / "shm" / "image_000000_2").read_bytes()
    total = int.from_bytes(chunk[0:8], "big")
    bs = int.from_bytes(chunk[8:12], "big")
    assert total == 200 * 300 * 2 and bs % 8 == 0
    pos, done, planes = 12, 0, bytearray()
    while total - done >= bs or (total - done) >= 16:
        n = min(bs, (total - done) // 16 * 16)   # full blocks, then one block of a multiple of 8 elements
        if n == 0:
            break
        clen = int.from_bytes(chunk[pos:pos + 4], "big")
        out = ctypes.create_string_buffer(n)
        assert lz4.LZ4_decompress_safe(chunk[pos + 4:pos + 4 + clen], out, clen, n) == n
        planes += out.raw
        pos += 4 + clen
        done += n
    tail = chunk[pos:]
    assert len(tail) == total - done
    # un-shuffle the blocks with numpy and compare with the source frame
    from ffs_amd import synth
    p = synth.params(300, 200, np.uint16, seed=7, background=2.0, n_spots=40, sigma=(0.8, 1.6),
                     peak=(30.0, 5000.0), max_value=65535)
    want = synth.frames(p, [0], threads=1)[0].ravel()
    got = np.empty(total // 2, np.uint16)
    off = 0
    buf = np.frombuffer(bytes(planes), np.uint8)
    while off < done:
        n = min(bs, done - off)
        nelem = n // 2
        bits = np.unpackbits(buf[off:off + n].reshape(16, nelem // 8), axis=1, bitorder="little")  # [bit][elem]
        vals = (bits.astype(np.uint32) << np.arange(16, dtype=np.uint32)[:, None]).sum(0)
        got[off // 2:off // 2 + nelem] = vals
        off += n
    got[done // 2:] = np.frombuffer(tail, np.uint16)
    assert np.array_equal(got, want)
