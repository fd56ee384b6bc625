"""GPU: the rarely taken paths -- dense candidates (LDS queue overflow in the candidate kernel, the
chunked path of the exact kernel), capacity overflow errors, extreme shapes."""
import numpy as np
import pytest

from util import assert_frame_matches_oracle, make_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", [0, 1])
@pytest.mark.parametrize("kind", ["single_photons", "uniform_noise", "stripes", "saturated_blocks"])
def test_dense_candidate_frames(ffs, kind, path):
    rng = np.random.default_rng(hash(kind) % 1000)
    H, W = 300, 1300
    if kind == "single_photons":       # every photon passes the signal test: ~3 % candidates
        img = (rng.random((H, W)) < 0.03).astype(np.uint16)
    elif kind == "uniform_noise":      # huge dispersion everywhere
        img = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    elif kind == "stripes":            # every third column bright: queue bursts in every row
        img = rng.poisson(1.0, (H, W)).astype(np.uint16)
        img[:, ::3] += 40
    else:                              # windows with sum >= 8192 skip the dispersion screen
        img = rng.poisson(3.0, (H, W)).astype(np.uint16)
        for _ in range(60):
            y, x = rng.integers(0, H - 6), rng.integers(0, W - 6)
            img[y:y + rng.integers(1, 6), x:x + rng.integers(1, 6)] = rng.integers(9000, 65536)
    mask = (rng.random((H, W)) > 0.01).astype(np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_strong_per_frame=W * H)
    ctx.set_tuning(threshold_path=path)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    fr = ctx.stream().process(img)[0]
    assert_frame_matches_oracle(fr, img, mask)


def test_capacity_overflow_is_absorbed(ffs):
    """More strong pixels than `max_strong_per_frame`: not an error any more -- the frame is run again with
    room for it and the result is the oracle's (the reference has no capacity at all)."""
    rng = np.random.default_rng(3)
    H, W = 120, 200
    img = rng.poisson(1.0, (H, W)).astype(np.uint16)
    img[::7, ::5] = 900                      # ~700 isolated strong pixels
    mask = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_strong_per_frame=100)
    ctx.set_params(want_strong_list=1)
    st = ctx.stream()
    fr = st.process(img)[0]
    assert fr.num_strong_pixels > 600
    assert_frame_matches_oracle(fr, img, mask)
    # the stream stays usable
    quiet = rng.poisson(1.0, (H, W)).astype(np.uint16)
    quiet[50, 60:63] = 300
    fr = st.process(quiet)[0]
    assert 1 <= fr.num_strong_pixels <= 100
    assert_frame_matches_oracle(fr, quiet, mask)


@pytest.mark.parametrize("H,W", [(1, 1), (3, 5), (7, 9), (6, 700), (2000, 8), (40, 4097), (16, 10240), (5000, 48)])
def test_extreme_shapes(ffs, H, W):
    rng = np.random.default_rng(H * 31 + W)
    img = rng.poisson(2.0, (H, W)).astype(np.uint16)
    img[rng.integers(0, H, 5), rng.integers(0, W, 5)] += 500
    mask = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16)
    ctx.set_params(want_strong_mask=1, want_strong_list=1, min_spot_size=1)
    fr = ctx.stream().process(img)[0]
    assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)


def test_bad_arguments_are_rejected(ffs):
    with pytest.raises(ffs.FfsError):
        ffs.Context(0, 10)
    with pytest.raises(ffs.FfsError):
        ffs.Context(20000, 10)
    ctx = ffs.Context(64, 64, max_batch=2)
    st = ctx.stream()
    with pytest.raises(ffs.FfsError):
        st.wait()                                            # nothing submitted
    with pytest.raises(ffs.FfsError):
        st.submit(np.zeros((3, 64, 64), np.uint16))          # more than max_batch
    with pytest.raises(ffs.FfsError):
        ctx.set_params(min_count=1)                          # standalone.cc:60 asserts min_count > 1
    st.submit(np.zeros((64, 64), np.uint16))
    with pytest.raises(ffs.FfsError):
        st.submit(np.zeros((64, 64), np.uint16))             # batch already in flight
    assert st.wait()[0].num_strong_pixels == 0


@pytest.mark.parametrize("direct", [1, 0])
def test_many_components_per_frame_both_record_paths(ffs, direct):
    """Thousands of components per frame: more than the 256 per frame the copy path brings back
    speculatively (so its top-up copy runs) and enough to exercise the direct-to-host path too."""
    from util import assert_frame_matches_oracle
    rng = np.random.default_rng(8)
    H, W = 300, 400
    frames = rng.poisson(0.05, (2, H, W)).astype(np.uint16)          # lonely photons: thousands of tiny components
    frames[rng.random((2, H, W)) < 0.01] += 9
    mask = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    ctx.set_tuning(direct_records=direct)                               # (before the first stream)
    ctx.set_params(min_spot_size=1, want_strong_list=1, max_peak_centroid_separation=50.0)
    st = ctx.stream()
    for rep in range(2):                                                # 2nd pass: speculative size has grown
        res = st.process(frames)
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, mask, min_spot_size=1, max_sep=50.0)
        assert min(fr.n_components for fr in res) > 600


def test_stream_closed_with_a_compressed_batch_in_flight(ffs):
    """Destroying a stream joins the helper thread of ffs_submit_compressed and drains the GPU work."""
    from ffs_amd import bslz4
    from util import make_frame
    img, _ = make_frame(W=300, H=200, seed=2, n_spots=30)
    ctx = ffs.Context(300, 200, np.uint16, max_batch=2)
    st = ctx.stream()
    st.submit_compressed([bslz4.compress(img), bslz4.compress(img)])
    st.close()                                                          # no wait()
    st2 = ctx.stream()
    assert st2.process(img)[0].num_strong_pixels > 0                    # the context is still usable


def _checker_frame(W, H, lo=0, hi=1000):
    img = np.full((H, W), lo, np.uint16)
    img[::2, ::2] = hi
    img[1::2, 1::2] = hi
    return img


def test_frame_beyond_the_list_capacity_inside_a_batch(ffs):
    """One frame with far more strong pixels (and components) than the stream's lists hold, between two
    normal frames: the batch must come back complete and equal to the oracle (the reference's
    ConnectedComponents has no capacity, connected_components.cc:24-32)."""
    W, H = 300, 200
    normal0, mask = make_frame(W=W, H=H, seed=11, n_spots=25)
    normal1, _ = make_frame(W=W, H=H, seed=12, n_spots=25)
    dense = _checker_frame(W, H)                       # every other pixel strong, each its own component
    frames = np.stack([normal0, dense, normal1])
    ctx = ffs.Context(W, H, np.uint16, max_batch=3, max_strong_per_frame=2000)   # lists far too small for `dense`
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    st = ctx.stream()
    res = st.process(frames, first_frame_id=7)
    assert [r.frame_id for r in res] == [7, 8, 9]
    assert res[1].num_strong_pixels > 20000 and res[1].n_components > 20000
    for fr, img in zip(res, frames):
        assert_frame_matches_oracle(fr, img, mask)
    # the same stream keeps working, and a second overflowing batch reuses the one-frame stream
    res2 = st.process(frames[::-1].copy())
    for fr, img in zip(res2, frames[::-1]):
        assert_frame_matches_oracle(fr, img, mask)
    # components alone over the limit: many 1-pixel components, few enough strong pixels for the list
    ctx2 = ffs.Context(W, H, np.uint16, max_batch=2, max_strong_per_frame=40000)
    ctx2.set_params(want_strong_list=1)
    st2 = ctx2.stream()
    for fr, img in zip(st2.process(np.stack([dense, normal0])), (dense, normal0)):
        assert_frame_matches_oracle(fr, img, mask)


def test_default_capacity_overflow_full_size_lists(ffs):
    """Default capacities (2^18 strong pixels, 65536 components per frame) exceeded by a frame that is half
    strong pixels."""
    W, H = 1024, 640
    dense = _checker_frame(W, H)
    normal, mask = make_frame(W=W, H=H, seed=5, n_spots=40)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    st = ctx.stream()
    res = st.process(np.stack([normal, dense]))
    assert res[1].num_strong_pixels > (1 << 18)
    for fr, img in zip(res, (normal, dense)):
        assert_frame_matches_oracle(fr, img, mask)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
def test_bright_window_list_overflow_falls_back(ffs, dtype):
    """Windows whose sums leave the streaming kernel's exact range (16-bit: sum p >= 65536; 32-bit: a pixel
    >= 2^24 nearby) go onto a list for the gather kernel.  With the list shrunk to 8 entries it overflows, and
    the batch must come back right all the same (re-run inside ffs_wait with those windows marked in the plane as
    candidates for the exact kernel)."""
    rng = np.random.default_rng(12)
    H, W = 240, 400
    img = rng.poisson(3.0, (H, W)).astype(dtype)
    top = 65535 if dtype == np.uint16 else (1 << 24) + 1000
    for _ in range(40):
        y, x = rng.integers(0, H - 6), rng.integers(0, W - 6)
        img[y:y + rng.integers(1, 5), x:x + rng.integers(1, 5)] = rng.integers(top // 2, top + 1)
    mask = (rng.random((H, W)) > 0.01).astype(np.uint8)
    ctx = ffs.Context(W, H, dtype, max_batch=2)
    ctx.set_tuning(bright_cap=8)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    st = ctx.stream()
    for fr, im in zip(st.process(np.stack([img, img[::-1].copy()])), (img, img[::-1])):
        assert_frame_matches_oracle(fr, im, mask)
    quiet = rng.poisson(3.0, (H, W)).astype(dtype)       # and the stream is fine afterwards
    assert_frame_matches_oracle(st.process(quiet[None])[0], quiet, mask)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
@pytest.mark.parametrize("group", [1, 3])
def test_super_row_groups(ffs, dtype, group):
    """The streaming kernels lay the frames of a batch side by side in one super row, as many as keep a group's
    buffers below 2 GiB (58 Eiger-16M frames); a batch beyond that is cut into several groups.  Tuning
    "frames_per_group" forces small groups so that the cut (3 + 3 + 2 frames, and one frame per group) is exercised on small frames."""
    W, H, B = 333, 77, 8
    frames, mask = [], None
    for i in range(B):
        img, mask = make_frame(W=W, H=H, dtype=dtype, seed=40 + i, n_spots=12, masked=True)
        frames.append(img)
    ctx = ffs.Context(W, H, dtype, max_batch=B)
    ctx.set_tuning(frames_per_group=group)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    st = ctx.stream()
    for fr, img in zip(st.process(np.stack(frames)), frames):
        assert_frame_matches_oracle(fr, img, mask)
    for fr, img in zip(st.process(np.stack(frames[:5])), frames[:5]):      # a partial last group
        assert_frame_matches_oracle(fr, img, mask)


@pytest.mark.parametrize("ccl", [1, 2])
@pytest.mark.parametrize("sched", [0, 3])
@pytest.mark.parametrize("path", [0, 1])
def test_sparse_stage_variants_agree(ffs, ccl, sched, path):
    """Tuning "sparse_stage": 2 = one launch per batch, a workgroup per frame (k_frame_chain: forest in LDS up to 20480
    strong pixels, global arrays beyond), 1 = four grid-wide kernels; "sched": shared dense / sparse / upload streams per
    context (3) or one stream per ffs_stream (0); "threshold_path".  Every combination must give the oracle's result: a
    sparse frame, a frame beyond the LDS forest, an empty frame and a frame with a row-wrap pair."""
    rng = np.random.default_rng(21)
    W, H = 640, 480
    sparse, mask = make_frame(W=W, H=H, seed=31, n_spots=40)
    dense = rng.poisson(1.0, (H, W)).astype(np.uint16)
    dense[rng.random((H, W)) < 0.09] += 60                      # ~27 k strong pixels: beyond kChainLdsEntries
    empty = np.zeros((H, W), np.uint16)
    wrap = rng.poisson(1.0, (H, W)).astype(np.uint16)
    wrap[100, W - 1] = 500
    wrap[101, 0] = 400                                             # (W-1, y) -- (0, y+1): one component for the reference
    wrap[200:203, 300:303] = 300
    frames = np.stack([sparse, dense, empty, wrap])
    ones = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=4, max_strong_per_frame=60000)
    ctx.set_tuning(sparse_stage=ccl, sched=sched, threshold_path=path)
    ctx.set_params(want_strong_mask=1, want_strong_list=1, min_spot_size=1, max_peak_centroid_separation=3.0)
    st = ctx.stream()
    for rep in range(2):                                           # (second pass: every buffer has been used once)
        res = st.process(frames, first_frame_id=10 * rep)
        assert res[1].num_strong_pixels > 20480 and res[2].num_strong_pixels == 0
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=1, max_sep=3.0)
    # and without the dense mask asked for (the default of the hot path)
    ctx.set_params(want_strong_mask=0, want_strong_list=1, min_spot_size=1, max_peak_centroid_separation=3.0)
    for fr, img in zip(st.process(frames), frames):
        assert_frame_matches_oracle(fr, img, ones, min_spot_size=1, max_sep=3.0)


def test_streams_of_one_context_from_several_threads(ffs):
    """The reference runs one worker thread per CUDA stream (spotfinder.cc:725-752).  Here the streams of a context
    share its HIP streams (one for the dense kernels, two for the sparse launches, one for uploads): four threads,
    each with its own ffs_stream, submit raw, device-resident-style and compressed batches at the same time and
    every result must be the oracle's."""
    import threading
    from ffs_amd import bslz4
    W, H, B = 320, 240, 3
    frames, mask = [], None
    for i in range(12):
        img, mask = make_frame(W=W, H=H, seed=70 + i, n_spots=20, masked=True)
        frames.append(img)
    ctx = ffs.Context(W, H, np.uint16, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=1)
    errors = []

    def worker(t):
        try:
            st = ctx.stream()
            for rep in range(6):
                mine = [frames[(t * 3 + rep + j) % len(frames)] for j in range(B)]
                if (rep + t) % 2:
                    res = st.process_compressed([bslz4.compress(f) for f in mine], first_frame_id=100 * t + rep)
                else:
                    res = st.process(np.stack(mine), first_frame_id=100 * t + rep)
                assert [r.frame_id for r in res] == [100 * t + rep + j for j in range(B)]
                for fr, img in zip(res, mine):
                    assert_frame_matches_oracle(fr, img, mask)
            st.close()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
def test_random_shapes_and_densities(ffs, dtype):
    """Small frames of random shape, mask and strong-pixel density (from none to a quarter of the frame), in
    batches of random size: plane rows of one to five 16-byte segments, tiles cut by the frame's last rows, waves of
    the per-frame workgroup with nothing to do, runs crossing word and segment boundaries, row-wrap pairs."""
    rng = np.random.default_rng(2024 if dtype == np.uint16 else 2025)
    for case in range(24):
        W = int(rng.integers(1, 640))
        H = int(rng.integers(1, 300))
        B = int(rng.integers(1, 5))
        mask = (rng.random((H, W)) > rng.choice([0.0, 0.02, 0.3])).astype(np.uint8)
        ctx = ffs.Context(W, H, dtype, max_batch=B, max_strong_per_frame=W * H)
        ctx.set_mask(mask)
        mss, sep = int(rng.integers(0, 4)), float(rng.choice([0.0, 1.5, 20.0]))
        ctx.set_params(want_strong_mask=int(case % 2), want_strong_list=1, min_spot_size=mss, max_peak_centroid_separation=sep)
        st = ctx.stream()
        frames = []
        for f in range(B):
            img = rng.poisson(rng.choice([0.2, 2.0, 30.0]), (H, W)).astype(dtype)
            dens = rng.choice([0.0, 0.001, 0.02, 0.25])
            hot = rng.random((H, W)) < dens
            img[hot] += rng.integers(50, 4000, hot.sum()).astype(dtype)
            if W > 1 and H > 1 and f == 0:
                img[0, W - 1] = 3000
                img[1, 0] = 2500
            frames.append(img)
        res = st.process(np.stack(frames))
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, mask, min_spot_size=mss, max_sep=sep)
        st.close()


def _blob_frame(W, H, seed, n_blobs, rmin=3, rmax=7):
    """Fat spots on a quiet background: many strong pixels in few runs (what the extended algorithm's final mask looks like)."""
    rng = np.random.default_rng(seed)
    img = rng.poisson(1.0, (H, W)).astype(np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_blobs):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(rmin, rmax + 1)
        y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, H), max(cx - r, 0), min(cx + r + 1, W)
        sel = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][sel] = rng.integers(200, 4000, sel.sum()).astype(np.uint16)
    return img


@pytest.mark.parametrize("want_list", [1, 0])
@pytest.mark.parametrize("chain_runs", [1, 0, 2])
def test_run_based_sparse_stage(ffs, chain_runs, want_list):
    """Frames beyond the LDS forest of pixels (20480) whose RUNS fit (16384): from the stream's second dense batch on the one
    launch builds its forest over runs (k_frame_chain<uint16_t, true>).  Fat spots, spots across 32-pixel word boundaries and
    frame edges, rows that are one long run (32 word-runs chained), the reference's row-wrap edge between fat runs, equal
    peak intensities (ties go to the smallest (y, x)), an empty and a sparse frame in the same batch; `chain_runs` 0 (the four
    grid-wide kernels for such batches) must agree."""
    W, H = 1000, 700     # (W not a multiple of 32: the last word of a row is cut)
    a = _blob_frame(W, H, 1, 500)
    b = _blob_frame(W, H, 2, 420)
    b[300, :] = 900                       # a whole row: 32 word-runs, joined with everything it touches
    b[301, 0:40] = 900                    # row wrap: (W-1, 300) -- (0, 301)
    b[400:440, 31:33] = 700               # a vertical bar across a word boundary
    b[500, 64:96] = 1234                  # exactly one full word
    b[502, 63:97] = 1234                  # one pixel more on both sides
    b[0, 0:5] = 800; b[H - 1, W - 5:W] = 800; b[0, W - 3:W] = 800; b[1, 0:3] = 800   # corners; (W-1, 0) -- (0, 1) wraps
    c = _blob_frame(W, H, 3, 480)
    c[c > 150] = 2000                     # every strong pixel the same value: peaks decided by position alone
    sparse, _ = make_frame(W=W, H=H, seed=33, n_spots=30)
    empty = np.zeros((H, W), np.uint16)
    frames = np.stack([a, b, sparse, empty, c])
    ones = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=5, max_strong_per_frame=90000)
    ctx.set_tuning(chain_runs=chain_runs)
    # want_list 0: what the library does by default -- the run-based launch keeps the pixel lists inside LDS (need_lists = 0)
    ctx.set_params(want_strong_mask=want_list, want_strong_list=want_list, min_spot_size=1, max_peak_centroid_separation=3.0)
    st = ctx.stream()
    for rep in range(3):                  # (the first batch cannot know that it is dense; the later ones take the run-based launch)
        res = st.process(frames, first_frame_id=10 * rep)
        assert res[0].num_strong_pixels > 20480 and res[1].num_strong_pixels > 20480 and res[4].num_strong_pixels > 20480
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=1, max_sep=3.0)
    # the reference's default filters, no dense mask asked for, a masked detector
    mask = (np.random.default_rng(9).random((H, W)) > 0.002).astype(np.uint8)
    mask[:, 500:504] = 0
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=0, want_strong_list=want_list, min_spot_size=3, max_peak_centroid_separation=2.0)
    for fr, img in zip(st.process(frames), frames):
        assert_frame_matches_oracle(fr, img, mask)


@pytest.mark.parametrize("want_list", [1, 0])
def test_run_based_sparse_stage_overflow_falls_back(ffs, want_list):
    """A dense frame with more runs than the run-based launch holds in LDS (isolated strong pixels: as many runs as pixels)
    raises its flag; the batch is run again through the grid-wide kernels inside ffs_wait() and the stream keeps to them."""
    rng = np.random.default_rng(4)
    W, H = 640, 480
    noisy = rng.poisson(1.0, (H, W)).astype(np.uint16)
    noisy[rng.random((H, W)) < 0.09] += 60                      # ~27 k isolated strong pixels
    fat = _blob_frame(W, H, 5, 520)
    frames = np.stack([noisy, fat])
    ones = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2, max_strong_per_frame=60000)
    ctx.set_params(want_strong_list=want_list, min_spot_size=1)
    st = ctx.stream()
    for rep in range(3):
        res = st.process(frames)
        assert res[0].num_strong_pixels > 20480
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=1)
    fat_only = np.stack([fat, fat[::-1].copy()])
    for fr, img in zip(st.process(fat_only), fat_only):
        assert_frame_matches_oracle(fr, img, ones, min_spot_size=1)


@pytest.mark.parametrize("want_list", [1, 0])
@pytest.mark.parametrize("strong_log", [1, 0])
def test_wave_logs_and_bit_plane_agree(ffs, strong_log, want_list):
    """Tuning `strong_log`: the streaming kernel's per-wave logs merged by the sparse launch (1, default) against the bit plane
    (0).  Frames chosen for the merge: strong pixels in every strip and band, groups that straddle strips, a frame pair
    that shares a strip (frame width not a multiple of the strip), bright windows (decided by the sparse launch: the cores
    of saturated spots), the row-wrap pair, rows with hundreds of strong groups (a wave's log overflows: the batch falls back
    to the plane inside ffs_wait and the stream stays there), an empty frame; then the dense byte mask."""
    rng = np.random.default_rng(77)
    W, H, B = 1000, 300, 5
    frames = []
    for i in range(B - 1):
        img = rng.poisson(2.0, (H, W)).astype(np.uint16)
        for _ in range(150):
            y, x = rng.integers(0, H - 4), rng.integers(0, W - 4)
            img[y:y + rng.integers(1, 4), x:x + rng.integers(1, 6)] = rng.integers(200, 3000)
        for _ in range(12):                                   # saturated cores: windows with sum p >= 65536
            y, x = rng.integers(0, H - 8), rng.integers(0, W - 8)
            img[y:y + 6, x:x + 6] = rng.integers(20000, 65535)
        frames.append(img)
    frames[1][100, W - 1] = 900; frames[1][101, 0] = 800     # (W-1, y) -- (0, y+1)
    frames.append(np.zeros((H, W), np.uint16))
    frames = np.stack(frames)
    mask = np.ones((H, W), np.uint8)
    mask[:, 496:500] = 0
    mask[rng.random((H, W)) < 0.001] = 0
    ctx = ffs.Context(W, H, np.uint16, max_batch=B)
    ctx.set_tuning(strong_log=strong_log)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=want_list, min_spot_size=1)
    st = ctx.stream()
    for rep in range(2):
        for fr, img in zip(st.process(frames, first_frame_id=rep), frames):
            assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)
    ctx.set_params(want_strong_mask=1, want_strong_list=want_list, min_spot_size=3)
    for fr, img in zip(st.process(frames[:3]), frames[:3]):
        assert_frame_matches_oracle(fr, img, mask)
    # a row of strong groups: more entries than a wave's log holds -> the plane takes over, results unchanged
    busy = rng.poisson(1.0, (H, W)).astype(np.uint16)
    busy[20:60, ::5] = 500
    ctx.set_params(want_strong_mask=0, want_strong_list=want_list, min_spot_size=3)
    for rep in range(2):
        for fr, img in zip(st.process(np.stack([busy, frames[0]])), (busy, frames[0])):
            assert_frame_matches_oracle(fr, img, mask)


@pytest.mark.parametrize("dtype", [np.uint16, np.uint32])
@pytest.mark.parametrize("want_list", [1, 0])
def test_wave_log_of_several_register_runs(ffs, dtype, want_list):
    """A streaming wave keeps its log entries in registers, 64 at a time (kernels_stream.hpp `lbuf`), and writes a run when the next
    drain would not fit and when it ends.  Frames whose waves hold 100-200 entries each -- more than one run, fewer than the 256 a
    log takes -- must come out as the oracle's without a second pass: the path bits say the logs served, `reruns` says once."""
    rng = np.random.default_rng(12)
    W, H, B = 1000, 300, 3
    frames = []
    for i in range(B):
        img = rng.poisson(2.0, (H, W)).astype(dtype)
        for _ in range(520):
            y, x = rng.integers(0, H - 3), rng.integers(0, W - 3)
            img[y:y + rng.integers(1, 3), x:x + rng.integers(1, 4)] = rng.integers(200, 3000)
        for _ in range(6):
            y, x = rng.integers(0, H - 8), rng.integers(0, W - 8)
            img[y:y + 6, x:x + 6] = rng.integers(20000, 65535)
        frames.append(img)
    frames = np.stack(frames)
    mask = np.ones((H, W), np.uint8)
    mask[rng.random((H, W)) < 0.001] = 0
    ctx = ffs.Context(W, H, dtype, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=want_list, min_spot_size=1)
    st = ctx.stream()
    for rep in range(2):
        res = st.process(frames, first_frame_id=rep)
        path, reruns = st.last_path()
        assert "wave_logs" in path and reruns == 0, (path, reruns)
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, mask, min_spot_size=1)
    assert res[0].num_strong_pixels > 1500


@pytest.mark.parametrize("want_list", [1, 0])
def test_sparse_and_dense_batches_alternate_on_one_stream(ffs, want_list):
    """Which sparse stage a batch gets follows what the stream's previous batch held: wave logs for sparse data, the plane with
    the run-based launch for dense data, and a dense batch that arrives on the logs is run again through the plane inside
    ffs_wait (flag 64) without switching the logs off for good.  Sparse / dense / dense / sparse / sparse / dense."""
    W, H = 1000, 700
    sparse = [make_frame(W=W, H=H, seed=300 + i, n_spots=40)[0] for i in range(2)]
    dense = [_blob_frame(W, H, 40 + i, 500) for i in range(2)]
    ones = np.ones((H, W), np.uint8)
    ctx = ffs.Context(W, H, np.uint16, max_batch=2, max_strong_per_frame=90000)
    ctx.set_params(want_strong_list=want_list, min_spot_size=2)
    st = ctx.stream()
    for k, batch in enumerate([sparse, dense, dense, sparse, sparse, dense, [sparse[0], dense[1]], sparse]):
        res = st.process(np.stack(batch), first_frame_id=10 * k)
        for fr, img in zip(res, batch):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=2)


def test_stack_created_after_the_batch_is_refused(ffs):
    """Tuning `device_lists` (default 2): a batch leaves its strong-pixel lists on the device while a 3D stack is alive or the
    host asked for them.  A stack created AFTER the batch was submitted cannot take that batch (an error that says so, not
    garbage); the next batch is fine, and with `want_strong_list` the lists are always there."""
    from ffs_amd.api import FfsError
    W, H = 400, 300
    frames = np.stack([make_frame(W=W, H=H, seed=500 + i, n_spots=25)[0] for i in range(2)])
    ctx = ffs.Context(W, H, np.uint16, max_batch=2)
    st = ctx.stream()
    st.process(frames, first_frame_id=0)              # no stack alive, no list asked for
    stack = ffs.Stack3D(ctx)
    with pytest.raises(FfsError, match="strong-pixel lists"):
        stack.add_batch(st)
    st.process(frames, first_frame_id=0)
    stack.add_batch(st)                                # submitted while the stack was alive
    refl, n_calc, _, _ = stack.finish()
    from oracle import oracle as O
    lists = []
    for img in frames:
        strong = O.dispersion(img, np.ones((H, W), np.uint8))
        k = np.flatnonzero(strong).astype(np.uint64)
        lists.append((k, img.ravel()[k].astype(np.uint32)))
    want = O.cc3d(lists, W, H, 3, 2.0)
    assert n_calc == want.n_calculated
    del stack
    ctx2 = ffs.Context(W, H, np.uint16, max_batch=2)
    ctx2.set_params(want_strong_list=1)
    st2 = ctx2.stream()
    st2.process(frames, first_frame_id=0)
    stack2 = ffs.Stack3D(ctx2)
    stack2.add_batch(st2)                              # the host asked for the lists: they are on the device
    assert stack2.finish()[1] == want.n_calculated


@pytest.mark.parametrize("kind", ["dense_on_logs", "runs_overflow"])
def test_stack_takes_a_batch_that_was_run_again(ffs, kind):
    """`lists_valid` after a re-run inside ffs_wait: a 3D stack is alive (tuning `device_lists` = 2 -> the lists stay on the
    device), nobody asked for the lists on the host, and the batch is one the first launch cannot serve -- dense frames arriving
    on the wave logs (flag 64: again through the plane and the run-based launch) or a frame with more runs than the run-based
    launch holds (flag 16: again through the grid-wide kernels).  ffs_stack3d_add_batch must then read the lists of the
    SECOND pass: the 3D result is the oracle's for the oracle's own lists."""
    from oracle import oracle as O
    from util import assert_reflections_equal
    rng = np.random.default_rng(17)
    W, H = 640, 480
    ones = np.ones((H, W), np.uint8)
    if kind == "dense_on_logs":
        frames = np.stack([_blob_frame(W, H, 60, 520), _blob_frame(W, H, 61, 520)])
    else:
        noisy = rng.poisson(1.0, (H, W)).astype(np.uint16)
        noisy[rng.random((H, W)) < 0.09] += 60
        frames = np.stack([noisy, _blob_frame(W, H, 62, 520)])
    ctx = ffs.Context(W, H, np.uint16, max_batch=2, max_strong_per_frame=90000)
    ctx.set_params(min_spot_size=1, min_spot_size_3d=2)       # want_strong_list = 0
    st = ctx.stream()
    stack = ffs.Stack3D(ctx)
    lists = []
    for rep in range(3):                                       # sparse-path first batch, then the stream knows its data is dense
        res = st.process(frames, first_frame_id=2 * rep)
        assert max(r.num_strong_pixels for r in res) > 20480
        stack.add_batch(st)
        for fr, img in zip(res, frames):
            assert_frame_matches_oracle(fr, img, ones, min_spot_size=1)
            k = np.flatnonzero(O.dispersion(img, ones)).astype(np.uint64)
            lists.append((k, img.ravel()[k].astype(np.uint32)))
    refl, n_calc, fs, fp = stack.finish()
    want = O.cc3d(lists, W, H, 2, 2.0)
    assert (n_calc, fs, fp) == (want.n_calculated, want.n_filtered_size, want.n_filtered_sep)
    assert_reflections_equal(refl, want.reflections)
