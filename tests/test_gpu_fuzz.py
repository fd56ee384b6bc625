"""GPU: randomised parity sweep -- frame sizes around the kernels' strip / tile / block boundaries,
both pixel widths, random masks, background levels from empty to bright, saturated pixels, random
algorithm parameters, both algorithms and flavours, batches, raw and compressed input.  Everything the
C ABI returns is compared with the oracle, bit for bit.  Seeds are fixed: a failure names its case."""
import numpy as np
import pytest

from ffs_amd import bslz4
from oracle import oracle as O
from util import assert_frame_matches_oracle

pytestmark = pytest.mark.gpu

# widths around 496-px strips (u16), 240-px strips (u32), 56-px strips (extended first pass), 128-px pitch
WIDTHS = [8, 31, 55, 56, 57, 127, 128, 129, 239, 241, 495, 496, 497, 512, 991, 993, 1030]
HEIGHTS = [1, 2, 6, 7, 8, 9, 15, 17, 33, 64, 101]


def random_case(seed):
    rng = np.random.default_rng(seed)
    W = int(rng.choice(WIDTHS))
    H = int(rng.choice(HEIGHTS))
    dtype = np.uint16 if rng.random() < 0.7 else np.uint32
    lam = float(rng.choice([0.0, 0.05, 0.5, 2.0, 20.0, 300.0]))
    n = int(rng.integers(1, 4))
    frames = rng.poisson(lam, (n, H, W)).astype(np.int64)
    for f in frames:                                     # spots, some saturated, some on the borders
        for _ in range(int(rng.integers(0, max(2, W * H // 400)))):
            cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
            s = rng.uniform(0.5, 2.0)
            pk = rng.choice([20, 200, 5000, 70000 if dtype == np.uint16 else 3e6])
            y0, y1, x0, x1 = max(cy - 5, 0), min(cy + 6, H), max(cx - 5, 0), min(cx + 6, W)
            yy, xx = np.mgrid[y0:y1, x0:x1]
            f[y0:y1, x0:x1] += rng.poisson(pk * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)))
    top = np.iinfo(np.uint16).max if dtype == np.uint16 else (1 << 26)   # u32: some pixels above 2^24
    frames = np.minimum(frames, top).astype(dtype)
    mask = np.ones((H, W), np.uint8)
    kind = rng.integers(0, 4)
    if kind == 1:
        mask[rng.random((H, W)) < 0.03] = 0
    elif kind == 2:
        mask[:, W // 2:W // 2 + max(1, W // 20)] = 0
        mask[H // 3:H // 3 + max(1, H // 10), :] = 0
    elif kind == 3:
        mask[rng.random((H, W)) < 0.5] = 0
    params = dict(min_count=int(rng.choice([2, 2, 3, 5, 10])), nsig_b=float(rng.choice([6.0, 6.0, 3.0, 1.5])),
                  nsig_s=float(rng.choice([3.0, 3.0, 2.0, 5.5])), threshold=float(rng.choice([0.0, 0.0, 4.0])),
                  min_spot_size=int(rng.choice([3, 1, 0, 6])),
                  max_peak_centroid_separation=float(rng.choice([2.0, 0.5, 10.0])))
    algo = int(rng.integers(0, 2))
    flavour = int(rng.integers(0, 2)) if algo else 0
    max_valid = int(rng.choice([-1, -1, 1000]))
    return W, H, dtype, frames, mask, params, algo, flavour, max_valid, bool(rng.random() < 0.4)


@pytest.mark.parametrize("seed", range(240))
def test_random_case(ffs, seed, tuning=None, want_list=1, want_mask=1, passes=1):
    W, H, dtype, frames, mask, prm, algo, flavour, max_valid, compressed = random_case(seed)
    ctx = ffs.Context(W, H, dtype, max_batch=len(frames))
    if tuning:
        ctx.set_tuning(**tuning)
    ctx.set_mask(mask)
    ctx.set_params(algorithm=algo, extended_flavour=flavour, max_valid=max_valid, want_strong_mask=want_mask,
                   want_strong_list=want_list, want_reflections=1, **prm)
    st = ctx.stream()
    for _ in range(passes):   # (a stream's later batches follow what its earlier ones held: dense data changes the sparse launch)
        res = (st.process_compressed([bslz4.compress(f) for f in frames], first_frame_id=5) if compressed
               else st.process(frames, first_frame_id=5))
    p = O.DispParams()
    O.lib().ffs_oracle_default_disp_params(O.C.byref(p))
    p.min_count, p.nsig_b, p.nsig_s, p.threshold = prm["min_count"], prm["nsig_b"], prm["nsig_s"], prm["threshold"]
    for i, (fr, img) in enumerate(zip(res, frames)):
        assert fr.frame_id == 5 + i
        if algo:
            strong = O.dispersion_extended(img, mask, p, flavour=flavour, max_valid=float(max_valid))
        else:
            strong = O.dispersion(img, mask, p)
            if max_valid >= 0:
                strong = strong & (img <= max_valid)          # thresholding.cu:208-215
        assert_frame_matches_oracle(fr, img, mask, min_spot_size=prm["min_spot_size"],
                                    max_sep=prm["max_peak_centroid_separation"], strong=strong)


@pytest.mark.parametrize("seed", range(12))
def test_random_stack3d(ffs, seed):
    """Random sweeps: blobs that persist over a random number of frames, gaps in the frame numbering,
    slices added out of order; 3D components, their reflections and the per-signal view against the oracle."""
    from util import assert_reflections_equal
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.choice([64, 130, 300])), int(rng.choice([40, 90, 200]))
    NZ = int(rng.integers(2, 9))
    vol = rng.poisson(float(rng.choice([0.2, 2.0])), (NZ, H, W)).astype(np.int64)
    for _ in range(int(rng.integers(3, 40))):
        cz, cy, cx = rng.integers(0, NZ), rng.integers(0, H), rng.integers(0, W)
        sz, s, pk = rng.uniform(0.3, 2.5), rng.uniform(0.6, 1.8), rng.choice([30, 300, 4000])
        zz, yy, xx = np.mgrid[0:NZ, max(cy - 5, 0):min(cy + 6, H), max(cx - 5, 0):min(cx + 6, W)]
        vol[:, max(cy - 5, 0):min(cy + 6, H), max(cx - 5, 0):min(cx + 6, W)] += rng.poisson(
            pk * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s) - (zz - cz) ** 2 / (2 * sz * sz)))
    frames = np.minimum(vol, 65535).astype(np.uint16)
    mask = np.ones((H, W), np.uint8)
    if rng.random() < 0.5:
        mask[rng.random((H, W)) < 0.02] = 0
    min3d, sep = int(rng.choice([1, 3, 5])), float(rng.choice([2.0, 1.0, 50.0]))
    ctx = ffs.Context(W, H, np.uint16, max_batch=3)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_list=1, min_spot_size_3d=min3d, max_peak_centroid_separation=sep)
    st = ctx.stream()
    stack = ffs.Stack3D(ctx)
    ids = np.sort(rng.choice(np.arange(100, 100 + 3 * NZ), NZ, replace=False))     # gaps in the numbering
    lists = {}
    for z0 in range(0, NZ, 3):
        res = st.process(frames[z0:z0 + 3])
        for j, r in enumerate(res):
            lists[int(ids[z0 + j])] = (r.strong_k.copy(), r.strong_intensity.copy())
    for fid in rng.permutation(list(lists)):                                        # out of order
        stack.add_slice(int(fid), *lists[int(fid)])
    refl, n_calc, fs, fp = stack.finish()
    slices = [lists[int(f)] for f in ids]
    want = O.cc3d(slices, W, H, min3d, sep)
    assert (n_calc, fs, fp) == (want.n_calculated, want.n_filtered_size, want.n_filtered_sep)
    assert_reflections_equal(refl, want.reflections)
    assert np.array_equal(stack.signals()["reflection"], O.cc3d_signals(slices, W, H, min3d, sep))


@pytest.mark.parametrize("seed", range(16))
def test_random_chunks_decode(ffs, seed):
    """Decoder against data with very different LZ4 structure: sparse bit planes (many short sequences),
    ramps and periodic patterns (overlapping matches of every small offset), noise, constant frames."""
    if bslz4.liblz4() is None:
        pytest.skip("no liblz4")
    rng = np.random.default_rng(2000 + seed)
    W, H = int(rng.choice([64, 97, 333, 1030])), int(rng.choice([17, 64, 129]))
    dtype = np.uint16 if seed % 3 else np.uint32
    kind = seed % 8
    if kind == 0:
        img = (rng.random((H, W)) < 0.01) * rng.integers(1, 4, (H, W))
    elif kind == 1:
        img = np.arange(W * H).reshape(H, W) % int(rng.integers(2, 40))
    elif kind == 2:
        img = np.repeat(rng.integers(0, 1 << 12, (H, 1)), W, axis=1)
    elif kind == 3:
        img = rng.integers(0, np.iinfo(dtype).max, (H, W), endpoint=True)
    elif kind == 4:
        img = np.zeros((H, W)); img[H // 2, W // 2] = np.iinfo(dtype).max
    elif kind == 5:
        img = rng.poisson(0.05, (H, W))
    elif kind == 6:
        img = rng.poisson(40.0, (H, W))
    else:
        img = np.tile(rng.integers(0, 300, (1, int(rng.integers(1, 9)))), (H, W))[:H, :W]
    img = np.ascontiguousarray(img).astype(dtype)
    ctx = ffs.Context(W, H, dtype, max_batch=2)
    _, got = ctx.stream().decode_only([bslz4.compress(img, "lz4"), bslz4.compress(img[::-1].copy(), "lz4")])
    assert np.array_equal(got[0], img) and np.array_equal(got[1], img[::-1])


@pytest.mark.parametrize("seed", range(0, 240, 6))
def test_random_case_alternative_kernels(ffs, seed):
    """The same sweep through the paths that are not the default: bright windows marked in the plane for the exact
    kernel (both pixel widths), the one-pixel-per-lane extended first pass, the four grid-wide sparse kernels."""
    test_random_case(ffs, seed, tuning=dict(threshold_path=1, ext_first_pass=0, sparse_stage=1))


@pytest.mark.parametrize("seed", range(1, 240, 4))
def test_random_case_run_based_sparse_stage(ffs, seed):
    """The same sweep with every 16-bit frame's connected components built over runs of strong pixels (tuning
    `chain_runs` = 2: the launch dense frames take by themselves, kernels_chain.hpp): random shapes, masks, algorithms,
    densities and filters through phases E' / U' / P' / R'."""
    test_random_case(ffs, seed, tuning=dict(chain_runs=2))


@pytest.mark.parametrize("seed", range(2, 240, 4))
def test_random_case_bit_plane_instead_of_wave_logs(ffs, seed):
    """The same sweep with tuning `strong_log` = 0: the 16-bit streaming kernel scatters plane bytes, counters and occupancy
    bits and lists bright windows for k_bright_fix, the sparse launch compacts the plane (the default since round 3c is the
    wave logs, which the sweeps above go through)."""
    test_random_case(ffs, seed, tuning=dict(strong_log=0))


@pytest.mark.parametrize("seed", range(0, 240, 3))
def test_random_case_without_lists(ffs, seed):
    """The same sweep as the library runs it by default and as bench.py times it: nobody asks for the strong-pixel lists or the
    byte mask (`want_strong_list = 0`, `want_strong_mask = 0`), so the sparse launch keeps the lists inside LDS and skips their
    stores (`need_lists = 0`, kernels_chain.hpp).  Counts, boxes and reflections against the oracle; twice on the same stream."""
    test_random_case(ffs, seed, want_list=0, want_mask=0, passes=2)


@pytest.mark.parametrize("seed", range(1, 240, 6))
def test_random_case_without_lists_run_based(ffs, seed):
    """... and through the run-based launch (tuning `chain_runs` = 2), whose list stores are skipped the same way."""
    test_random_case(ffs, seed, tuning=dict(chain_runs=2), want_list=0, want_mask=0, passes=2)


@pytest.mark.parametrize("seed", range(2, 240, 12))
def test_random_case_without_lists_bit_plane(ffs, seed):
    """... through the bit plane (tuning `strong_log` = 0) and with the lists forced to stay on the device (`device_lists` = 1)."""
    test_random_case(ffs, seed, tuning=dict(strong_log=0), want_list=0, want_mask=0, passes=2)
    test_random_case(ffs, seed, tuning=dict(device_lists=1), want_list=0, want_mask=0)


@pytest.mark.parametrize("seed", range(3, 240, 8))
def test_random_case_gather_path(ffs, seed):
    """Tuning `threshold_path` = 2, the partner of `spotfinder --validate`: no streaming kernel -- the plane starts as the
    valid-pixel mask and k_exact gathers the window of every valid pixel.  Must give the oracle's result on its own."""
    test_random_case(ffs, seed, tuning=dict(threshold_path=2))
    test_random_case(ffs, seed, tuning=dict(threshold_path=2, sparse_stage=1, strong_log=0), want_list=0, want_mask=1)
