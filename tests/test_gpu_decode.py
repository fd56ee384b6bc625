"""GPU: bitshuffle-LZ4 chunks decoded on the device (csrc/kernels_decode.hpp) -- bit-exact round trip
against the frames that were compressed, for chunks written by the real LZ4 encoder (liblz4), by the
host tool's own encoder (SHM fixtures) and by a literals-only encoder; block edge cases (partial last
block, raw tail, odd widths); then the whole hot path from compressed input; then corrupt chunks."""
import os
import subprocess

import numpy as np
import pytest

from ffs_amd import bslz4
from util import assert_frame_matches_oracle, make_frame

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "fast-feedback-service_amd", "bin", "ffs_hosttool")

SHAPES = [
    (64, 64, np.uint16),      # exactly one 4096-element block
    (97, 61, np.uint16),      # odd width: pixel pairs straddle rows; partial block + raw tail (5917 = 4096+1816+5)
    (300, 200, np.uint16),
    (130, 90, np.uint32),     # 2048-element blocks
    (33, 31, np.uint32),      # 1023 elements: one short block + 7 raw
    (7, 1, np.uint16),        # fewer than 8 elements: raw tail only
    (1030, 517, np.uint16),
]


@pytest.mark.parametrize("W,H,dtype", SHAPES, ids=lambda v: getattr(v, "__name__", str(v)))
@pytest.mark.parametrize("encoder", ["lz4", "literals"])
def test_round_trip(ffs, W, H, dtype, encoder):
    if encoder == "lz4" and bslz4.liblz4() is None:
        pytest.skip("no liblz4")
    rng = np.random.default_rng(W * 1000 + H)
    frames = []
    img, _ = make_frame(W=max(W, 16), H=max(H, 16), dtype=dtype, seed=W + H, n_spots=10)
    frames.append(np.ascontiguousarray(img[:H, :W]))
    frames.append(rng.integers(0, np.iinfo(dtype).max, (H, W), dtype=dtype, endpoint=True))   # incompressible
    frames.append(np.zeros((H, W), dtype))                                                    # one long match
    frames.append((np.arange(W * H).reshape(H, W) % 251).astype(dtype))                       # overlapping matches
    ctx = ffs.Context(W, H, dtype, max_batch=len(frames))
    st = ctx.stream()
    chunks = [bslz4.compress(f, encoder) for f in frames]
    ms, got = st.decode_only(chunks)
    for g, f in zip(got, frames):
        d = np.argwhere(g != f)
        assert d.size == 0, f"{len(d)} pixels differ, first at (y,x)={d[:4].tolist()}: got {g[tuple(d[0])]} want {f[tuple(d[0])]}"


def test_hosttool_chunks_and_hot_path(ffs, tmp_path):
    """Chunks written by the host tool's encoder (an Eiger-stream directory) go through
    ffs_submit_compressed and give the oracle's answers."""
    from ffs_amd import synth
    assert subprocess.run([TOOL, "mkshm", "synth:tiny:4", str(tmp_path / "shm")]).returncode == 0
    chunks = [(tmp_path / "shm" / ("image_%06d_2" % i)).read_bytes() for i in range(4)]
    p = synth.params(300, 200, np.uint16, seed=7, background=2.0, n_spots=40, sigma=(0.8, 1.6), peak=(30.0, 5000.0),
                     max_value=65535)
    frames = synth.frames(p, range(4), threads=2)
    ctx = ffs.Context(300, 200, np.uint16, max_batch=4)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    st = ctx.stream()
    mask = np.ones((200, 300), np.uint8)
    res = st.process_compressed(chunks, first_frame_id=11)
    assert [r.frame_id for r in res] == [11, 12, 13, 14]
    for fr, img in zip(res, frames):
        assert_frame_matches_oracle(fr, img, mask)
    # zero copy: chunks placed in the stream's pinned buffer by the caller
    hb = st.host_bytes()
    cur, views = 0, []
    for c in chunks:
        hb[cur:cur + len(c)] = np.frombuffer(c, np.uint8)
        views.append(hb[cur:cur + len(c)])
        cur += len(c) + 3      # deliberately unaligned
    res2 = st.process_compressed(views)
    for a, b in zip(res, res2):
        assert a.num_strong_pixels == b.num_strong_pixels and len(a.boxes) == len(b.boxes)
    # and the uncompressed path still works on the same stream
    assert_frame_matches_oracle(st.process(frames[0])[0], frames[0], mask)


def test_bad_chunks_are_refused(ffs):
    W, H = 200, 100
    img, _ = make_frame(W=W, H=H, seed=3, n_spots=10)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    st = ctx.stream()
    good = bslz4.compress(img)
    with pytest.raises(ffs.FfsError, match="header"):
        st.decode_only([good[:8]])
    wrong = bslz4.compress(img[:50])
    with pytest.raises(ffs.FfsError, match="header says"):
        st.decode_only([wrong])
    with pytest.raises(ffs.FfsError, match="run past"):
        st.decode_only([good[:len(good) // 2]])
    # corrupt LZ4 payload: the first sequence of the first block becomes "no literals, then a match"
    # whose offset points before the start of the block
    bad = bytearray(good)
    bad[16] = 0x0F
    bad[17] = 0xFF
    bad[18] = 0xFF
    with pytest.raises(ffs.FfsError, match="corrupt"):
        st.decode_only([bytes(bad)])
    with pytest.raises(ffs.FfsError, match="corrupt"):
        st.process_compressed([bytes(bad)])
    with pytest.raises(ffs.FfsError, match="run past"):
        st.process_compressed([good[:len(good) // 2]])
    with pytest.raises(ffs.FfsError, match="header says"):
        st.submit_compressed([wrong])
    # a literal run longer than the block (literals-only block with its length bytes raised)
    lits = bslz4.compress(img, "literals")
    b2 = bytearray(lits)
    assert b2[16] == 0xF0 and b2[17] == 0xFF
    b2[16 + 1 + (8192 - 15) // 255] = 0xFE      # last length byte: run now overshoots the payload
    with pytest.raises(ffs.FfsError, match="corrupt"):
        st.decode_only([bytes(b2)])
    # a block that decodes cleanly but to fewer bytes than a block holds
    short = bytearray(good[:16]) + bytes([0x10, 0x00]) + bytes(good[18:])
    short[12:16] = (2).to_bytes(4, "big")
    with pytest.raises(ffs.FfsError):
        st.decode_only([bytes(short)])
    # the stream is still usable
    _, got = st.decode_only([good])
    assert np.array_equal(got[0], img)


def test_eiger16m_chunk(ffs):
    """Full-size frame through the real encoder; also reports the decode rate."""
    if bslz4.liblz4() is None:
        pytest.skip("no liblz4")
    from ffs_amd import synth
    img = synth.frame(synth.eiger16m_params(seed=2200, n_spots=800), 0)
    chunk = bslz4.compress(img, "lz4")
    H, W = img.shape
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    st = ctx.stream()
    ms, got = st.decode_only([chunk, chunk], iters=3)
    assert np.array_equal(got[0], img) and np.array_equal(got[1], img)
    print(f"eiger16m: chunk {len(chunk) / 1e6:.2f} MB ({img.nbytes / len(chunk):.1f}x), decode {ms:.3f} ms / 2 frames")
