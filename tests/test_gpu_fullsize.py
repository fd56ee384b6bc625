"""GPU: BASELINE.json's full sizes.  Two Eiger-16M frames and one 3072^2 u32 frame are checked
pixel-for-pixel against the oracle (about a second of CPU each); the rest of the batch is covered by
size-independent properties: batch-position invariance, idempotence, masked pixels never strong,
list/mask/count consistency, and a checksum over all spot lists that must not depend on how the
frames were batched."""
import hashlib

import numpy as np
import pytest

from util import assert_frame_matches_oracle

pytestmark = pytest.mark.gpu


def _digest(results):
    h = hashlib.sha256()
    for r in sorted(results, key=lambda r: r.frame_id):
        h.update(np.int64(r.frame_id).tobytes())
        h.update(r.strong_k.tobytes())
        h.update(r.boxes.tobytes())
        h.update(r.reflections.tobytes())
    return h.hexdigest()


def test_eiger16m_batch(ffs):
    from ffs_amd import synth
    p = synth.eiger16m_params()
    mask = synth.mask_eiger16m()
    frames = synth.frames(p, range(6))
    ctx = ffs.Context(4148, 4362, np.uint16, max_batch=6)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=1, want_strong_mask=1)
    st = ctx.stream()
    res = st.process(frames, first_frame_id=0)
    # pixel-for-pixel against the oracle on two of them
    for i in (0, 5):
        assert_frame_matches_oracle(res[i], frames[i], mask)
    for r, img in zip(res, frames):
        assert r.num_strong_pixels == len(r.strong_k) == int(r.strong_mask.sum())
        assert not r.strong_mask[mask == 0].any()
        assert (np.diff(r.strong_k.astype(np.int64)) > 0).all()          # sorted, unique
        assert (img.reshape(-1)[r.strong_k] == r.strong_intensity).all() and (r.strong_intensity > 0).all()
        assert r.n_components >= len(r.boxes) >= len(r.reflections)
        assert int(r.boxes["num_pixels"].sum()) == r.num_strong_pixels_filtered
        assert (r.boxes["num_pixels"] >= 3).all()
        assert (r.reflections["peak_centroid_distance"] <= 2.0).all()
        assert 500 < len(r.boxes) < 3000
    d_all = _digest(res)
    # same frames, different batching and order of submission -> identical results
    ctx.set_params(want_strong_mask=0)
    again = []
    for i in (3, 0, 5, 1, 4, 2):
        again += st.process(frames[i], first_frame_id=i)
    assert _digest(again) == d_all
    pairs = st.process(frames[4:6], first_frame_id=4) + st.process(frames[0:4], first_frame_id=0)
    assert _digest(pairs) == d_all


def test_jungfrau9m_u32_frame(ffs):
    from ffs_amd import synth
    p = synth.jungfrau9m_params()
    mask = synth.mask_modules(3072, 3072, 1024, 512, 0, 0)
    synth.mask_dead_pixels(mask, 4, 500)
    frames = synth.frames(p, range(2))
    ctx = ffs.Context(3072, 3072, np.uint32, max_batch=2)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=1, want_strong_mask=1)
    res = ctx.stream().process(frames)
    assert_frame_matches_oracle(res[0], frames[0], mask)
    assert res[1].num_strong_pixels == int(res[1].strong_mask.sum()) > 1000


def test_sweep_3d_eiger_slab(ffs):
    """configs[4] shape: a fine-phi sweep through 3D connected components (reduced to 12 frames
    for test time); checked against the oracle's 3D labelling."""
    from ffs_amd import synth
    from oracle import oracle as O
    from util import assert_reflections_equal
    NZ = 12
    p = synth.sweep_params(seed=5000, n_frames=NZ, n_spots=400)
    mask = synth.mask_eiger16m()
    ctx = ffs.Context(4148, 4362, np.uint16, max_batch=6)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=1, min_spot_size=3, min_spot_size_3d=15)
    st = ctx.stream()
    stack = ffs.Stack3D(ctx)
    slices = []
    for z0 in range(0, NZ, 6):
        frames = synth.frames(p, range(z0, z0 + 6))
        res = st.process(frames, first_frame_id=z0)
        stack.add_batch(st)
        slices += [(r.strong_k, r.strong_intensity) for r in res]
    refl, n_calc, fs, fp = stack.finish()
    want = O.cc3d(slices, 4148, 4362, 15, 2.0)
    assert (n_calc, fs, fp) == (want.n_calculated, want.n_filtered_size, want.n_filtered_sep)
    assert_reflections_equal(refl, want.reflections)
    assert len(refl) > 20


def test_sweep_3d_eiger_100_frames(ffs):
    """configs[4] at its stated size: a 100-frame Eiger-16M fine-phi sweep, 800 reflections with a rocking
    curve, `--min-spot-size 3 --min-spot-size-3d 15` (the shape of the reference's
    tests/3d_connected_components.sh:27-37), through the batch path, the device-resident 3D stack and its
    finish.  EVERY frame's counts, boxes, reflections and strong-pixel list are held to the oracle's dispersion + 2D labelling
    (16 oracle threads), and the slices the oracle's 3D labelling gets are the ORACLE's own lists, not the HIP path's: a wrong
    list on any frame fails here, whatever the 3D stage makes of it."""
    from concurrent.futures import ThreadPoolExecutor
    from ffs_amd import synth
    from oracle import oracle as O
    from util import assert_reflections_equal, assert_frame_matches_oracle, oracle_frame
    NZ, B = 100, 25
    p = synth.sweep_params(seed=5000, n_frames=NZ, n_spots=800)
    mask = synth.mask_eiger16m()
    ctx = ffs.Context(4148, 4362, np.uint16, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=1, min_spot_size=3, min_spot_size_3d=15)
    st = ctx.stream()
    stack = ffs.Stack3D(ctx)
    slices = []
    for z0 in range(0, NZ, B):
        frames = synth.frames(p, range(z0, z0 + B), threads=16)
        res = st.process(frames, first_frame_id=z0)
        stack.add_batch(st)
        with ThreadPoolExecutor(16) as ex:                       # (the C oracle runs without the interpreter lock)
            want = list(ex.map(lambda img: oracle_frame(img, mask), frames))
        for j in range(B):
            assert_frame_matches_oracle(res[j], frames[j], mask, precomputed=want[j])
            slices.append((want[j][1].k.astype(np.uint32), want[j][1].intensity.astype(np.uint32)))
        del frames, want
    refl, n_calc, fs, fp = stack.finish()
    assert stack.last_finish_ms() > 0
    want = O.cc3d(slices, 4148, 4362, 15, 2.0)
    assert (n_calc, fs, fp) == (want.n_calculated, want.n_filtered_size, want.n_filtered_sep)
    assert_reflections_equal(refl, want.reflections)
    assert len(refl) > 300 and (refl["z_max"] - refl["z_min"]).max() >= 3
    sig = stack.signals()
    want_sig = O.cc3d_signals(slices, 4148, 4362, 15, 2.0)
    assert np.array_equal(sig["reflection"], want_sig)
    assert np.array_equal(sig["z"], np.concatenate([np.full(len(k), z) for z, (k, _) in enumerate(slices)]))
    # a second finish on the same stack gives the same answer (pooled buffers, nothing consumed)
    refl2, n_calc2, _, _ = stack.finish()
    assert n_calc2 == n_calc
    assert_reflections_equal(refl2, want.reflections)
