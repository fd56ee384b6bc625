"""The sparse stage in small workgroups (csrc/kernels_band.hpp, tuning `sparse_bands`, the default when nobody reads the pixel lists):
one wave per band of a frame builds the band's components from the wave logs, a merge per frame joins them across the bands' seams.
Held to the oracle on frames made for the seams: components that cross one, several and all band boundaries, components that meet
only in a later band (labels must still follow their FIRST pixel), the reference's row-wrap edge (connected_components.cc:62-70)
inside a band and across a seam, saturated cores on a seam, masked columns, empty bands and frames -- and on bands that overflow
the plan (the batch is run again through the one-workgroup launch inside ffs_wait).  Every test asserts which launches ran
(ffs_stream_last_path): a parity test of a path that silently was not taken proves nothing."""
import numpy as np
import pytest

from util import assert_frame_matches_oracle

pytestmark = pytest.mark.gpu


def seam_frames(rng, W, H, dtype, band_rows, long_bars=True):
    """Five frames whose structures sit on the multiples of band_rows."""
    hi = 60000 if dtype == np.uint16 else 900000
    frames = []
    for i in range(4):
        img = rng.poisson(2.0, (H, W)).astype(dtype)
        for _ in range(W * H // 5000):            # (a wave's log holds 256 groups per strip and band)
            y, x = rng.integers(0, H - 4), rng.integers(0, W - 6)
            img[y:y + rng.integers(1, 4), x:x + rng.integers(1, 6)] = rng.integers(200, 3000)
        frames.append(img)
    a, b, c, d = frames
    seams = [s for s in range(band_rows, H - 30, band_rows)]   # (the shapes around a seam reach 29 rows below it)
    # a: vertical bars through one seam, through all of them; a bar that ends on a seam's last / first row
    for k, s in enumerate(seams):
        a[s - 3:s + 4, 40 + 17 * k] = 700
        a[s - 5:s, 300 + 11 * k] = 650          # ends on the last row of the band above
        a[s:s + 5, 500 + 13 * k] = 640          # starts on the first row of the band below
    if long_bars:
        a[2:H - 2, W - 150] = 900                # through every band
    else:                                        # (tall streaming bands: a bar of 150 rows is 150 groups of a wave's 256)
        for s in seams:
            a[s - 12:s + 12, W - 150] = 900
    # b: a "U": two arms that start in the first band and are joined only at the bottom of the last one -- the label follows the
    # first pixel, the joining happens many bands later.  (A wave's log holds 256 groups per strip and band: long bars keep their
    # distance, here and below.)
    if long_bars:
        b[5:H - 5, 60] = 800
        b[30:H - 5, 460] = 800
        b[H - 6, 60:461] = 800                   # (a band wave holds 768 strong pixels)
    else:
        for s in seams:                          # small U's over every seam
            b[s - 9:s + 9, 60] = 800
            b[s - 5:s + 9, 90] = 800
            b[s + 8, 60:91] = 800
    # c: the row-wrap edge: (W-1, y) -- (0, y+1), inside a band and exactly across each seam; and its one-sided neighbours
    c[20, W - 1] = 900; c[21, 0] = 800
    for s in seams:
        c[s - 1, W - 1] = 900; c[s, 0] = 850                       # across the seam
    c[seams[0] + 7, W - 1] = 900; c[seams[0] + 8, 1] = 850          # not an edge (x = 1)
    c[seams[0] + 20, W - 2] = 900; c[seams[0] + 21, 0] = 850        # not an edge (x = W - 2)
    # ... and a comb: teeth from several bands hanging on one spine, one of them through every seam
    c[10, W // 2 - 50:W // 2 + 51] = 750
    if long_bars:
        c[10:seams[-1] + 9, W // 2] = 750
    else:
        c[seams[-1] - 20:seams[-1] + 9, W // 2] = 750
    for s in seams:
        c[s - 4:s + 5, W // 2 - 24] = 750
        c[s - 4:s + 5, W // 2 + 24] = 750
    c[seams[-1] + 8, W // 2 - 24:W // 2 + 25] = 750                 # the teeth of the last seam meet the long one below it
    # d: saturated cores (windows with sum p >= 65536: decided by the band wave) on and next to the seams, one at the frame's corner
    for k, s in enumerate(seams):
        d[s - 3:s + 3, 200 + 60 * k:206 + 60 * k] = rng.integers(hi // 3, hi)
        d[s + 9:s + 14, 230 + 60 * k:236 + 60 * k] = rng.integers(hi // 3, hi)
    d[0:5, 0:5] = hi
    d[H - 5:H, W - 5:W] = hi
    frames.append(np.zeros((H, W), dtype))
    return np.stack(frames)


@pytest.mark.parametrize("dtype,W,H,target_waves", [(np.uint16, 1000, 300, 0), (np.uint16, 1203, 517, 0), (np.uint32, 1000, 300, 0), (np.uint16, 700, 160, 0),
                                                     (np.uint16, 1000, 300, 22), (np.uint16, 1203, 517, 40)])
def test_bands_match_oracle_on_the_seams(ffs, dtype, W, H, target_waves):
    """target_waves: tuning that makes the streaming kernel's bands taller (as large batches do) -- 150 and 173 rows here, which the band
    waves take as two sub-bands each."""
    rng = np.random.default_rng(505)
    band_rows = -(-H // (H // 72))
    if target_waves == 22:
        band_rows = 75        # two streaming bands of 150 rows, cut in two
    elif target_waves == 40:
        band_rows = 87        # three of 173 (the last: 171), cut in two
    frames = seam_frames(rng, W, H, dtype, band_rows, long_bars=not target_waves)
    mask = np.ones((H, W), np.uint8)
    mask[:, 496:500] = 0
    mask[rng.random((H, W)) < 0.001] = 0
    ctx = ffs.Context(W, H, dtype, max_batch=len(frames))
    if target_waves:
        ctx.set_tuning(target_waves=target_waves)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=0, min_spot_size=1)
    st = ctx.stream()
    for rep in range(3):
        order = np.roll(np.arange(len(frames)), rep)     # (a frame's place in the super row moves: other strips, other neighbours)
        res = st.process(frames[order], first_frame_id=10 * rep)
        path, reruns = st.last_path()
        assert "bands" in path and "wave_logs" in path and reruns == 0, (path, reruns)
        for fr, img in zip(res, frames[order]):
            assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)
    # the same frames with the filters on, one at a time (a one-frame launch has other strips)
    ctx.set_params(want_strong_list=0, min_spot_size=3, max_peak_centroid_separation=2.0)
    for img in frames[:3] if not target_waves else []:
        (fr,) = st.process(img[None])
        assert "bands" in st.last_path()[0]
        assert_frame_matches_oracle(fr, img, mask)
    # against the one-workgroup launch (tuning sparse_bands = 0) and with the lists asked for (never the bands)
    ctx.set_tuning(sparse_bands=0)
    res = st.process(frames)
    assert st.last_path()[0] >= {"wave_logs", "frame_chain"} and "bands" not in st.last_path()[0]
    ctx.set_tuning(sparse_bands=1)
    ctx.set_params(want_strong_list=1, min_spot_size=3)
    res = st.process(frames)
    assert "bands" not in st.last_path()[0]
    for fr, img in zip(res, frames):
        assert_frame_matches_oracle(fr, img, mask)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
@pytest.mark.parametrize("n_bands", [3, 5, 7, 11])
def test_any_number_of_streaming_bands(ffs, dtype, n_bands):
    """Tuning `stream_bands`: the streaming launch's units are dealt to the XCDs in eight contiguous chunks (ffs_device.h, stream_unit), so
    the number of bands need not be a multiple of eight; the logs are addressed by (super row, band, strip).  Odd numbers of bands --
    band heights of 173, 104, 74 and 47 rows, the first two cut into sub-bands by the band waves -- with the seams' frames, through
    the band launches and through the one-workgroup launch."""
    rng = np.random.default_rng(606)
    W, H = 1203, 517
    band_rows = -(-H // n_bands)
    frames = seam_frames(rng, W, H, dtype, band_rows if band_rows <= 96 else band_rows // 2, long_bars=False)
    mask = np.ones((H, W), np.uint8)
    mask[:, 496:500] = 0
    mask[rng.random((H, W)) < 0.001] = 0
    ctx = ffs.Context(W, H, dtype, max_batch=len(frames))
    ctx.set_tuning(stream_bands=n_bands)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=0, min_spot_size=1)
    st = ctx.stream()
    for rep in range(2):
        order = np.roll(np.arange(len(frames)), rep)
        res = st.process(frames[order], first_frame_id=10 * rep)
        path, reruns = st.last_path()
        assert "bands" in path and "wave_logs" in path and reruns == 0, (path, reruns)
        for fr, img in zip(res, frames[order]):
            assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)
    ctx.set_tuning(sparse_bands=0)
    res = st.process(frames)
    assert st.last_path()[0] >= {"wave_logs", "frame_chain"}
    for fr, img in zip(res, frames):
        assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)


def test_bands_beyond_their_plan_fall_back(ffs):
    """A band with more strong pixels, components or seam pixels than the band wave holds raises flag 128: ffs_wait runs the batch again
    through the one-workgroup launch, the stream stays there for its next batches and comes back."""
    rng = np.random.default_rng(9)
    W, H = 1000, 300
    base = rng.poisson(2.0, (H, W)).astype(np.uint16)
    for _ in range(100):
        y, x = rng.integers(0, H - 4), rng.integers(0, W - 6)
        base[y:y + 3, x:x + 4] = rng.integers(200, 3000)
    ones = np.ones((H, W), np.uint8)
    fat = base.copy(); fat[100:130, 200:240] = 3000                  # 1200 strong pixels in one band
    dots = base.copy(); dots[160:220:4, 100:900:40] = 900           # 300 one-pixel components in one band (and no more than 150 groups in a strip: the logs hold them)
    line = base.copy(); line[149, 100:500] = 900                    # 400 strong pixels in a band's last row
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    ctx.set_params(want_strong_list=0, min_spot_size=1)
    st = ctx.stream()
    for img in (fat, dots, line):
        res = st.process(np.stack([base, img]))
        path, reruns = st.last_path()
        assert reruns == 1 and "frame_chain" in path and "bands" not in path, (path, reruns)
        for fr, im in zip(res, (base, img)):
            assert_frame_matches_oracle(fr, im, ones, min_spot_size=1)
        # the stream stays with the one-workgroup launch for a while ...
        res = st.process(np.stack([base, base]))
        assert st.last_path() == ({"wave_logs", "frame_chain"}, 0)
        for _ in range(40):
            res = st.process(np.stack([base, base]))
        # ... and comes back
        assert st.last_path() == ({"wave_logs", "bands"}, 0)
        for fr in res:
            assert_frame_matches_oracle(fr, base, ones, min_spot_size=1)


@pytest.mark.parametrize("dense_overlap", [0, 1])
def test_bands_with_several_batches_in_flight(ffs, dense_overlap):
    """Four streams of one context, their batches in flight together (the bench's shape): every batch through the bands.
    dense_overlap = 1: the A/B partner that hands consecutive streaming kernels over between two HIP streams (off by default:
    measured slower) gives the same results."""
    rng = np.random.default_rng(31)
    W, H, B = 1100, 450, 6
    frames = seam_frames(rng, W, H, np.uint16, 75)
    frames = np.concatenate([frames, frames[:1]])
    ones = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=B)
    ctx.set_tuning(sparse_bands=2, dense_overlap=dense_overlap)      # (bands always: by default the second and third batch of a pipeline that fills take the one-workgroup launch)
    ctx.set_params(want_strong_list=0, min_spot_size=2)
    streams = [ctx.stream() for _ in range(4)]
    want = None
    for rnd in range(3):
        for s in streams:
            s.submit(frames, first_frame_id=0)
        for s in streams:
            res = s.wait()
            assert "bands" in s.last_path()[0]
            if want is None:
                want = [assert_frame_matches_oracle(fr, img, ones, min_spot_size=2) for fr, img in zip(res, frames)]
            else:
                for fr, img, w in zip(res, frames, want):
                    assert_frame_matches_oracle(fr, img, ones, min_spot_size=2, precomputed=w)
    # the default: a context with four streams is a pipeline that fills -- bands at every depth
    ctx.set_tuning(sparse_bands=1)
    for s in streams:
        s.submit(frames, first_frame_id=0)
    for s in streams:
        res = s.wait()
        assert "bands" in s.last_path()[0]
        for fr, img, w in zip(res, frames, want):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=2, precomputed=w)
    # ... and one with two or three streams takes the one-workgroup launch (with its head start) when both / all are in flight
    ctx2 = ffs.Context(W, H, np.uint16, max_batch=B)
    ctx2.set_params(want_strong_list=0, min_spot_size=2)
    two = [ctx2.stream() for _ in range(2)]
    for s in two:
        s.submit(frames, first_frame_id=0)
    paths = []
    for s in two:
        res = s.wait()
        paths.append("bands" in s.last_path()[0])
        for fr, img, w in zip(res, frames, want):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=2, precomputed=w)
    assert paths == [True, False], paths


def test_bands_of_two_heights(ffs):
    """Tuning `band_taper`: the streaming kernel's last bands per XCD are half as tall (50 rows) as the others (100 rows, cut into
    two sub-bands): a short band has ONE sub-band with rows, and its last row meets the first row of the band after the empty one."""
    rng = np.random.default_rng(77)
    W, H = 700, 2400
    frames = []
    for i in range(2):
        img = rng.poisson(2.0, (H, W)).astype(np.uint16)
        for _ in range(300):
            y, x = rng.integers(0, H - 4), rng.integers(0, W - 6)
            img[y:y + rng.integers(1, 4), x:x + rng.integers(1, 6)] = rng.integers(200, 3000)
        img[3:H - 3, 100 + 400 * i] = 700            # through every band and sub-band
        img[49::50, W - 1] = 900                     # the row-wrap edge on every multiple of 50 rows: every kind of seam
        img[50::50, 0] = 850
        frames.append(img)
    frames = np.stack(frames)
    ones = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    ctx.set_tuning(band_taper=50)
    ctx.set_params(want_strong_list=0, min_spot_size=1)
    st = ctx.stream()
    res = st.process(frames)
    assert st.last_path() == ({"wave_logs", "bands"}, 0)
    for fr, img in zip(res, frames):
        assert_frame_matches_oracle(fr, img, ones, min_spot_size=1)
