"""GPU parity: libffs_hip.so (through its C ABI) against the oracle on the same inputs.
Integer results (masks, lists, boxes, counts) bit-exact; float32 centroids bit-exact
(BASELINE.json asks for 1e-6)."""
import numpy as np
import pytest

from util import assert_frame_matches_oracle

pytestmark = pytest.mark.gpu


def _spotty(rng, H, W, lam, nspots, dtype=np.uint16, peak=400):
    img = rng.poisson(lam, (H, W)).astype(np.int64)
    for _ in range(nspots):
        cy, cx = rng.integers(0, H), rng.integers(0, W)
        s = rng.uniform(0.7, 1.8)
        pk = rng.uniform(10, peak)
        y0, y1 = max(cy - 6, 0), min(cy + 7, H)
        x0, x1 = max(cx - 6, 0), min(cx + 7, W)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1] += rng.poisson(pk * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)))
    return np.minimum(img, np.iinfo(dtype).max).astype(dtype)


def _mask(rng, H, W, dead=40):
    m = np.ones((H, W), np.uint8)
    if W > 40:
        m[:, W // 2 - 3:W // 2 + 2] = 0
    if H > 40:
        m[H // 3:H // 3 + 5, :] = 0
    ys, xs = rng.integers(0, H, dead), rng.integers(0, W, dead)
    m[ys, xs] = 0
    return m


@pytest.mark.parametrize("H,W", [(13, 20), (48, 64), (100, 497), (389, 517), (200, 1000), (64, 2100)])
def test_small_frames_match_oracle(ffs, H, W):
    rng = np.random.default_rng(H * 10007 + W)
    ctx = ffs.Context(W, H, np.uint16, max_batch=3)
    mask = _mask(rng, H, W)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1, want_reflections=1)
    st = ctx.stream()
    frames = np.stack([_spotty(rng, H, W, lam, max(3, H * W // 3000)) for lam in (0.3, 2.0, 30.0)])
    res = st.process(frames, first_frame_id=7)
    assert [r.frame_id for r in res] == [7, 8, 9]
    total = 0
    for fr, img in zip(res, frames):
        _, cc, _ = assert_frame_matches_oracle(fr, img, mask)
        total += cc.num_strong_pixels
    assert total > 0


def test_no_mask_and_all_masked(ffs):
    rng = np.random.default_rng(5)
    H, W = 70, 130
    ctx = ffs.Context(W, H, np.uint16, max_batch=1)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    st = ctx.stream()
    img = _spotty(rng, H, W, 1.0, 20)
    ones = np.ones((H, W), np.uint8)
    assert_frame_matches_oracle(st.process(img)[0], img, ones)      # NULL mask = all valid
    ctx.set_mask(np.zeros((H, W), np.uint8))
    fr = st.process(img)[0]
    assert fr.num_strong_pixels == 0 and len(fr.boxes) == 0 and fr.strong_mask.sum() == 0
    ctx.set_mask(ones)
    zero = np.zeros((H, W), np.uint16)
    fr = st.process(zero)[0]                                            # empty frame
    assert fr.num_strong_pixels == 0 and fr.n_components == 0


def test_saturated_and_edge_pixels(ffs):
    """Bright pixels at the corners/edges (clipped windows) and 65535 values."""
    H, W = 40, 72
    rng = np.random.default_rng(11)
    img = rng.poisson(1.0, (H, W)).astype(np.uint16)
    for (y, x) in [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (0, 30), (H - 1, 31), (17, 0), (18, W - 1)]:
        img[y, x] = 65535
    img[20:23, 40:43] = 60000
    mask = np.ones((H, W), np.uint8)
    mask[10, 10] = 0
    ctx = ffs.Context(W, H, np.uint16)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    fr = ctx.stream().process(img)[0]
    assert_frame_matches_oracle(fr, img, mask)
    assert fr.num_strong_pixels >= 8


def test_row_wrap_quirk(ffs):
    """(W-1, y) and (0, y+1) are joined because the reference links k to k+1 with no row-end
    check (connected_components.cc:62-70)."""
    H, W = 32, 64
    img = np.ones((H, W), np.uint16)
    img[10, W - 1] = 500
    img[10, W - 2] = 500
    img[11, 0] = 500
    img[11, 1] = 500
    ctx = ffs.Context(W, H, np.uint16)
    ctx.set_params(want_strong_mask=1, want_strong_list=1, min_spot_size=1)
    fr = ctx.stream().process(img)[0]
    mask = np.ones((H, W), np.uint8)
    assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)
    assert fr.num_strong_pixels == 4 and fr.n_components == 1
    assert fr.boxes[0]["l"] == 0 and fr.boxes[0]["r"] == W - 1


def test_config1_plumbing_frames(ffs):
    """BASELINE.json configs[0]: 10 x 1024^2 u16 synthetic frames with Poisson spots."""
    from ffs_amd import synth
    p = synth.config1_params()
    mask = synth.config1_mask()
    frames = synth.frames(p, range(10))
    ctx = ffs.Context(1024, 1024, np.uint16, max_batch=10)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    res = ctx.stream().process(frames)
    for fr, img in zip(res, frames):
        _, cc, _ = assert_frame_matches_oracle(fr, img, mask)
        assert cc.num_strong_pixels > 500


@pytest.mark.parametrize("path", [0, 1])
def test_threshold_paths(ffs, path):
    """Both threshold paths (bright windows -> list -> fix-up kernel / -> plane -> exact kernel) must give the
    oracle's result; frames chosen so the lane-group queue wraps many times per wave."""
    rng = np.random.default_rng(99)
    H, W = 700, 1100
    mask = _mask(rng, H, W, dead=300)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    ctx.set_tuning(threshold_path=path)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    frames = np.stack([_spotty(rng, H, W, 0.05, 400, peak=2000), _spotty(rng, H, W, 8.0, 3000, peak=800)])
    for fr, img in zip(ctx.stream().process(frames), frames):
        assert_frame_matches_oracle(fr, img, mask)


def test_params_min_count_and_max_valid(ffs):
    """GPU-reference flavoured parameters: min_count 3 (spotfinder.cuh:18) and a trusted
    maximum on the centre pixel (thresholding.cu:208-215)."""
    from oracle import oracle as O
    rng = np.random.default_rng(21)
    H, W = 60, 90
    img = _spotty(rng, H, W, 0.5, 25, peak=3000)
    mask = _mask(rng, H, W, dead=900)
    ctx = ffs.Context(W, H, np.uint16)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, min_count=3, max_valid=1000)
    fr = ctx.stream().process(img)[0]
    p = O.DispParams(3, 3, 3, 0.0, 6.0, 3.0)
    want = O.dispersion(img, mask, p)
    want[img > 1000] = 0
    np.testing.assert_array_equal(fr.strong_mask, want)
    assert want.sum() > 0


@pytest.mark.parametrize("H,W", [(50, 70), (301, 517)])
def test_uint32_frames_match_oracle(ffs, H, W):
    rng = np.random.default_rng(H + W)
    ctx = ffs.Context(W, H, np.uint32, max_batch=2)
    mask = _mask(rng, H, W)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    a = _spotty(rng, H, W, 5.0, 30, np.uint32, peak=200000)
    b = _spotty(rng, H, W, 300.0, 30, np.uint32, peak=3000000)
    b[5, 5] = (1 << 24) + 5          # >= 2^24: excluded from sums, still a legal centre
    b[H - 2, W - 3] = 0xFFFFFFFF
    frames = np.stack([a, b])
    res = ctx.stream().process(frames)
    for fr, img in zip(res, frames):
        assert_frame_matches_oracle(fr, img, mask)


def test_stack3d_matches_oracle(ffs):
    from oracle import oracle as O
    from ffs_amd import synth
    W, H, NZ = 300, 200, 12
    p = synth.sweep_params(seed=77, n_frames=NZ, n_spots=60, width=W, height=H)
    frames = synth.frames(p, range(NZ))
    mask = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=5)
    ctx.set_params(want_strong_list=1, min_spot_size_3d=4)
    st = ctx.stream()
    stack = ffs.Stack3D(ctx)
    slices = []
    for z0 in range(0, NZ, 5):
        res = st.process(frames[z0:z0 + 5], first_frame_id=100 + z0)
        stack.add_batch(st)
        slices += [(r.strong_k, r.strong_intensity) for r in res]
    refl, n_calc, fs, fp = stack.finish()
    want = O.cc3d(slices, W, H, 4, 2.0)
    assert n_calc == want.n_calculated and fs == want.n_filtered_size and fp == want.n_filtered_sep
    from util import assert_reflections_equal
    assert_reflections_equal(refl, want.reflections)
    assert len(refl) > 5 and (refl["z_max"] > refl["z_min"]).any()
    # per-signal view (Reflection3D::signals_ order) against the oracle's membership
    sig = stack.signals()
    want_sig = O.cc3d_signals(slices, W, H, 4, 2.0)
    assert np.array_equal(sig["reflection"], want_sig) and (want_sig >= 0).any() and (want_sig < 0).any()
    k_all = np.concatenate([k for k, _ in slices]).astype(np.int64)
    assert np.array_equal(sig["x"], k_all % W) and np.array_equal(sig["y"], k_all // W)
    assert np.array_equal(sig["z"], np.concatenate([np.full(len(k), z) for z, (k, _) in enumerate(slices)]))
    assert np.array_equal(sig["intensity"], np.concatenate([i for _, i in slices]))
    counts = np.bincount(sig["reflection"][sig["reflection"] >= 0], minlength=len(refl))
    assert np.array_equal(counts, refl["num_pixels"])


def test_spot_centres_rows(ffs):
    """ffs_stream_spot_centres: the (frame_id, x, y, z) rows fed to the multi-GPU gather equal what
    dist.pack_spots builds from the per-frame results."""
    from ffs_amd import dist as D
    rng = np.random.default_rng(3)
    H, W = 120, 200
    ctx = ffs.Context(W, H, np.uint16, max_batch=3)
    st = ctx.stream()
    frames = np.stack([_spotty(rng, H, W, 1.0, 25) for _ in range(3)])
    res = st.process(frames, first_frame_id=40)
    want = D.pack_spots(res, 500)
    got = np.full((501, 4), -1, np.float32)
    n = st.pack_spot_centres(got, 500)
    assert n == int(want[500].view(np.uint32)[0]) > 5
    assert np.array_equal(got[:n].view(np.uint32), want[:n].view(np.uint32))     # frame ids are bit patterns
    assert np.array_equal(got[500].view(np.uint32), want[500].view(np.uint32))   # (rows written, rows wanted)
    assert (got[n:500] == -1).all()                      # rows beyond the count are left alone
    small = np.zeros((4, 4), np.float32)                 # capacity smaller than the batch: reported, not silent
    with pytest.raises(ffs.FfsError):
        st.pack_spot_centres(small, 3)
    assert np.array_equal(small[:3].view(np.uint32), want[:3].view(np.uint32))
    assert tuple(small[3].view(np.uint32)[:2]) == (3, n)


@pytest.mark.parametrize("transport", ["peer", "rccl"])
def test_stack3d_fed_from_two_contexts(ffs, transport, monkeypatch):
    """The multi-GPU rotation path with both contexts on the one GPU of the test box: frames alternate between
    two contexts, every batch's lists cross over into the first context's 3D stack (ffs_stack3d_add_batch with
    a foreign stream) -- by device copies, and by RCCL send/recv on a one-rank communicator."""
    from oracle import oracle as O
    from ffs_amd import synth
    from util import assert_reflections_equal
    monkeypatch.setenv("FFS_GATHER", transport)
    used = ffs.multi_init([0, 0], transport)
    assert used == ("rccl" if transport == "rccl" else "none")      # one distinct device: nothing to peer with
    W, H, NZ = 300, 200, 12
    p = synth.sweep_params(seed=78, n_frames=NZ, n_spots=60, width=W, height=H)
    frames = synth.frames(p, range(NZ))
    ctxs = [ffs.Context(W, H, np.uint16, max_batch=2) for _ in range(2)]
    for c in ctxs:
        c.set_params(want_strong_list=1, min_spot_size_3d=4)
    streams = [c.stream() for c in ctxs]
    stack = ffs.Stack3D(ctxs[0])
    slices = [None] * NZ
    for b, z0 in enumerate(range(0, NZ, 2)):
        st = streams[b % 2]
        res = st.process(frames[z0:z0 + 2], first_frame_id=z0)
        stack.add_batch(st)
        for j, r in enumerate(res):
            slices[z0 + j] = (r.strong_k.copy(), r.strong_intensity.copy())
    refl, n_calc, fs, fp = stack.finish()
    want = O.cc3d(slices, W, H, 4, 2.0)
    assert (n_calc, fs, fp) == (want.n_calculated, want.n_filtered_size, want.n_filtered_sep)
    assert_reflections_equal(refl, want.reflections)
    assert len(refl) > 5


@pytest.mark.parametrize("transport", ["peer", "rccl"])
def test_stack3d_takes_an_overflowing_frame_from_the_second_context(ffs, transport, monkeypatch):
    """A frame with more strong pixels than the stream's lists hold (an ice ring) is absorbed by ffs_wait on any GPU; its
    list lives on the host afterwards.  When that stream belongs to ANOTHER context than the 3D stack, the batch must
    still reach the stack: the frames that fitted are packed and sent device to device, the overflow frame's list goes up
    from the host -- one such frame landing on GPU 1-7 must not end a sweep the single-GPU path finishes."""
    from oracle import oracle as O
    from ffs_amd import synth
    from util import assert_reflections_equal
    monkeypatch.setenv("FFS_GATHER", transport)
    ffs.multi_init([0, 0], transport)
    W, H, NZ = 200, 120, 8
    p = synth.sweep_params(seed=91, n_frames=NZ, n_spots=30, width=W, height=H)
    frames = synth.frames(p, range(NZ)).copy()
    rng = np.random.default_rng(5)
    for z in (3, 6):                                        # both land in batches of the second context (z // 2 odd)
        frames[z][rng.random((H, W)) < 0.5] += 400          # ~50 % strong pixels: far beyond the lists' 600 entries
    ctxs = [ffs.Context(W, H, np.uint16, max_batch=2, max_strong_per_frame=600) for _ in range(2)]
    for c in ctxs:
        c.set_params(want_strong_list=1, min_spot_size_3d=4)
    streams = [c.stream() for c in ctxs]
    stack = ffs.Stack3D(ctxs[0])
    slices = [None] * NZ
    for b, z0 in enumerate(range(0, NZ, 2)):
        st = streams[b % 2]
        res = st.process(frames[z0:z0 + 2], first_frame_id=z0)
        stack.add_batch(st)
        for j, r in enumerate(res):
            slices[z0 + j] = (r.strong_k.copy(), r.strong_intensity.copy())
    assert len(slices[3][0]) > 5000 and len(slices[6][0]) > 5000 and len(slices[2][0]) < 600
    refl, n_calc, fs, fp = stack.finish()
    want = O.cc3d(slices, W, H, 4, 2.0)
    assert (n_calc, fs, fp) == (want.n_calculated, want.n_filtered_size, want.n_filtered_sep)
    assert_reflections_equal(refl, want.reflections)


def test_resolution_mask_matches_oracle_on_eiger_geometry(ffs):
    """ffs_ctx_apply_resolution_mask against the oracle's float32 restatement of masking.cu:37-73,99-147 on the
    Eiger-16M frame.  Both evaluate sqrtf / atanf / sinf in float32, but the device library and the host libm may
    round the last place differently, which can flip a pixel whose d-spacing sits within a few float32 ulp of dmin or
    dmax.  Stated and asserted: every differing pixel is such a pixel, and there are at most 64 of them in 18 M."""
    from ffs_amd import synth
    from oracle import oracle as O
    W, H = 4148, 4362
    g = dict(wavelength=0.976, distance=0.15, beam_center_x=2074.3, beam_center_y=2181.7,
             pixel_size_x=75e-6, pixel_size_y=75e-6)
    dmin, dmax = 1.45, 25.0
    mask = synth.mask_eiger16m()
    ctx = ffs.Context(W, H, np.uint16, max_batch=1)
    ctx.set_mask(mask)
    ctx.apply_resolution_mask(g["wavelength"], g["distance"], g["beam_center_x"], g["beam_center_y"],
                              g["pixel_size_x"], g["pixel_size_y"], dmin, dmax)
    got = ctx.get_mask()
    want, res = O.resolution_mask(mask, dmin=dmin, dmax=dmax, **g)
    assert 0.2 < want.sum() / mask.sum() < 0.95          # the filter really cuts (corners and beam centre)
    diff = np.argwhere((got != 0) != (want != 0))
    print(f"resolution mask: {len(diff)} of {W * H} pixels differ from the host libm evaluation")
    assert len(diff) <= 64
    for y, x in diff:
        r = np.float32(res[y, x])
        ulp = np.spacing(r)
        assert min(abs(r - np.float32(dmin)), abs(r - np.float32(dmax))) <= 4 * ulp, (y, x, r)
    assert (got[mask == 0] == 0).all()
    # the tables the threshold kernel reads were rebuilt from the new mask: a frame processed now follows it
    img, _ = None, None
    p = synth.eiger16m_params(seed=2003)
    img = synth.frames(p, range(1), threads=8)[0]
    ctx.set_params(want_strong_mask=1)
    fr = ctx.stream().process(img[None])[0]
    strong = O.dispersion(img, got)
    assert np.array_equal(fr.strong_mask, strong) and strong.sum() > 1000
