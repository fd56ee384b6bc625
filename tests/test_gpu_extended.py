"""GPU parity for `-a dispersion_extended` (baseline/spotfinder/baseline.cpp:730-761): every stage of
the HIP path -- first-pass mask, eroded signal region, strong pixels, components, centroids -- against the
oracle on the same inputs, bit-exact, for both flavours, 16- and 32-bit pixels, masks, image edges,
parameter changes and batches; then the full-size detector frame."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_frame_matches_oracle, make_frame

pytestmark = pytest.mark.gpu

CASES = [
    dict(W=97, H=61, dtype=np.uint16, seed=1, n_spots=12),
    dict(W=200, H=150, dtype=np.uint16, seed=2, n_spots=60, masked=True),
    dict(W=130, H=90, dtype=np.uint32, seed=3, n_spots=20, masked=True),
    dict(W=64, H=64, dtype=np.uint16, seed=4, n_spots=200),     # everything is "not background"
    dict(W=40, H=9, dtype=np.uint16, seed=5, n_spots=4),        # shorter than the 11x11 window
    dict(W=517, H=389, dtype=np.uint16, seed=6, n_spots=150, masked=True),
    dict(W=1100, H=75, dtype=np.uint32, seed=7, n_spots=80, masked=True),
    dict(W=56, H=300, dtype=np.uint16, seed=8, n_spots=30),     # exactly one strip wide
    dict(W=57, H=130, dtype=np.uint16, seed=9, n_spots=20),     # one pixel into the second strip
    dict(W=1030, H=64, dtype=np.uint16, seed=10, n_spots=60, masked=True, peak=(2000.0, 60000.0)),  # x >= 8192 windows
    dict(W=300, H=200, dtype=np.uint16, seed=11, n_spots=40, background=400.0),   # bright background
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d-%s" % (c["W"], c["H"], np.dtype(c["dtype"]).name))
@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("variant", [2, 0, 3, 4, 5])
def test_every_stage_matches_oracle(ffs, case, flavour, variant):
    # tuning "ext_first_pass": 2 = the streaming kernel decides the first pass exactly in its drain (16-bit default),
    # 0 = the one-pixel-per-lane first-pass kernel (what 32-bit pixels always use)
    # variant 3: the streaming first pass with the erosion fused into the final pass's tiles (tuning "ext_fused" = 1: an A/B partner,
    # slower than the two kernels it replaces -- DESIGN.md section 4 -- and held to the same planes)
    # variants 4 / 5: the erosion kernels beside the default (tuning "ext_erode": 0 = a lane per word column, 2 = strips of 16 rows;
    # default 1 = strips of 32 rows)
    if variant in (0, 3) and case["dtype"] == np.uint32:
        pytest.skip("32-bit pixels have one first-pass kernel")
    img, mask = make_frame(**case)
    H, W = img.shape
    ctx = ffs.Context(W, H, img.dtype, max_batch=2)
    ctx.set_tuning(ext_first_pass=0 if variant == 0 else 2, ext_fused=1 if variant == 3 else 0,
                   ext_erode={4: 0, 5: 2}.get(variant, 1))
    ctx.set_mask(mask)
    ctx.set_params(algorithm=ffs.ALGO_DISPERSION_EXTENDED, extended_flavour=flavour, want_strong_mask=1,
                   want_strong_list=1, want_reflections=1)
    st = ctx.stream()
    fr = st.process(img)[0]
    strong, first, eroded = O.dispersion_extended(img, mask, flavour=flavour, debug=True)
    got_first, got_eroded = st.debug_bitplane(0, 1), st.debug_bitplane(0, 2)
    d = np.argwhere(got_first != first)
    assert d.size == 0, f"first pass: {len(d)} mismatches, first at (y,x)={d[:5].tolist()}"
    d = np.argwhere(got_eroded != eroded)
    assert d.size == 0, f"erosion: {len(d)} mismatches, first at (y,x)={d[:5].tolist()}"
    assert np.array_equal(st.debug_bitplane(0, 0), strong)
    assert_frame_matches_oracle(fr, img, mask, strong=strong)


def test_batch_of_different_frames_and_parameters(ffs):
    W, H = 260, 140
    frames, masks = zip(*[make_frame(W=W, H=H, seed=20 + i, n_spots=10 + 25 * i, masked=True) for i in range(4)])
    mask = masks[0]
    ctx = ffs.Context(W, H, np.uint16, max_batch=4)
    ctx.set_mask(mask)
    ctx.set_params(algorithm=ffs.ALGO_DISPERSION_EXTENDED, extended_flavour=1, min_count=3, nsig_b=4.0, nsig_s=2.5,
                   max_valid=2000, want_strong_mask=1, want_strong_list=1, min_spot_size=2)
    st = ctx.stream()
    res = st.process(np.stack(frames), first_frame_id=3)
    p = O.DispParams()
    O.lib().ffs_oracle_default_disp_params(O.C.byref(p))
    p.min_count, p.nsig_b, p.nsig_s = 3, 4.0, 2.5
    total = 0
    for fr, img in zip(res, frames):
        strong = O.dispersion_extended(img, mask, p, flavour=1, max_valid=2000.0)
        assert_frame_matches_oracle(fr, img, mask, min_spot_size=2, strong=strong)
        total += int(strong.sum())
    assert total > 0
    # and back to the standard algorithm on the same stream
    ctx.set_params(algorithm=ffs.ALGO_DISPERSION, extended_flavour=0, min_count=2, nsig_b=6.0, nsig_s=3.0, max_valid=-1,
                   min_spot_size=3)
    assert_frame_matches_oracle(st.process(frames[1])[0], frames[1], mask)


@pytest.mark.parametrize("erode,sparse", [(2, 0), (2, 1), (1, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("flavour", [0, 1])
def test_planes_of_successive_batches(ffs, erode, sparse, flavour):
    """The planes of a stream take turns (the first-pass plane always; with tuning "ext_e_sparse" = 1 the signal-region plane too:
    it is then cleared BEHIND the batch before, and the strip erosion stores only the words that hold a pixel of the region):
    every batch's planes must be that batch's, whatever the two batches before left in them (dense frames, then sparse and
    empty ones, batches of different lengths).  All erosion kernels (tuning "ext_erode")."""
    W, H = 700, 260
    specs = [dict(seed=60, n_spots=400), dict(seed=61, n_spots=3), dict(seed=62, n_spots=0), dict(seed=63, n_spots=250, masked=True),
             dict(seed=64, n_spots=1), dict(seed=65, n_spots=120), dict(seed=66, n_spots=0), dict(seed=67, n_spots=300)]
    frames = [make_frame(W=W, H=H, **sp)[0] for sp in specs]
    mask = make_frame(W=W, H=H, seed=63, n_spots=1, masked=True)[1]
    ctx = ffs.Context(W, H, np.uint16, max_batch=3)
    ctx.set_tuning(ext_erode=erode, ext_e_sparse=sparse)
    ctx.set_mask(mask)
    ctx.set_params(algorithm=ffs.ALGO_DISPERSION_EXTENDED, extended_flavour=flavour, want_reflections=1)
    st = ctx.stream()
    want = [O.dispersion_extended(img, mask, flavour=flavour, debug=True) for img in frames]
    for batch in ([0, 1, 2], [3], [4, 5], [6, 7, 0], [2, 6], [0, 3, 7], [1]):
        res = st.process(np.stack([frames[i] for i in batch]))
        for f, i in enumerate(batch):
            strong, first, eroded = want[i]
            assert np.array_equal(st.debug_bitplane(f, 1), first), (batch, f)
            assert np.array_equal(st.debug_bitplane(f, 2), eroded), (batch, f)
            assert_frame_matches_oracle(res[f], frames[i], mask, strong=strong)


def test_empty_and_fully_masked(ffs):
    W, H = 90, 50
    img, _ = make_frame(W=W, H=H, seed=40, n_spots=10)
    ctx = ffs.Context(W, H, np.uint16, max_batch=1)
    ctx.set_params(algorithm=ffs.ALGO_DISPERSION_EXTENDED, want_strong_mask=1)
    st = ctx.stream()
    assert_frame_matches_oracle(st.process(img)[0], img, np.ones((H, W), np.uint8),
                                strong=O.dispersion_extended(img, np.ones((H, W), np.uint8)))
    ctx.set_mask(np.zeros((H, W), np.uint8))
    fr = st.process(img)[0]
    assert fr.num_strong_pixels == 0 and fr.strong_mask.sum() == 0
    ctx.set_mask(np.ones((H, W), np.uint8))
    fr = st.process(np.zeros((H, W), np.uint16))[0]
    assert fr.num_strong_pixels == 0


def test_invalid_algorithm_is_rejected(ffs):
    ctx = ffs.Context(64, 64, np.uint16, max_batch=1)
    with pytest.raises(ffs.FfsError):
        ctx.set_params(algorithm=7)
    ctx.set_params(algorithm=0)
    with pytest.raises(ffs.FfsError):
        ctx.set_params(extended_flavour=2)


def test_eiger16m_frame(ffs):
    """Full detector size, module-gap mask: all three stages and the frame summary."""
    from ffs_amd import synth
    p = synth.eiger16m_params(seed=2100, n_spots=1200)
    mask = synth.mask_eiger16m()
    img = synth.frame(p, 0)
    H, W = img.shape
    ctx = ffs.Context(W, H, np.uint16, max_batch=1)
    ctx.set_mask(mask)
    ctx.set_params(algorithm=ffs.ALGO_DISPERSION_EXTENDED, want_strong_list=1)
    st = ctx.stream()
    fr = st.process(img)[0]
    strong, first, eroded = O.dispersion_extended(img, mask, debug=True)
    assert np.array_equal(st.debug_bitplane(0, 1), first)
    assert np.array_equal(st.debug_bitplane(0, 2), eroded)
    assert_frame_matches_oracle(fr, img, mask, strong=strong)
    assert fr.num_strong_pixels > 1000
