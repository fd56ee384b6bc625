"""Shared helpers for the parity tests: the oracle is the checker, never the thing tested."""
import numpy as np

from oracle import oracle as O


def make_frame(W, H, dtype=np.uint16, seed=0, n_spots=20, masked=False, background=2.0, peak=(30.0, 3000.0)):
    """One deterministic synthetic frame (+ mask: module gaps, dead pixels and a rectangle when masked)."""
    from ffs_amd import synth
    p = synth.params(W, H, dtype, seed=seed, background=background, n_spots=n_spots, sigma=(0.7, 1.8), peak=peak,
                     max_value=65535 if dtype == np.uint16 else (1 << 20))
    img = synth.frame(p, 0)
    mask = np.ones((H, W), np.uint8)
    if masked:
        mask = synth.mask_modules(W, H, max(W // 3, 8), max(H // 2, 8), 3, 4)
        mask = synth.mask_dead_pixels(mask, seed + 100, max(W * H // 400, 1))
        mask = synth.mask_rect(mask, W // 5, W // 5 + 9, H // 4, H // 4 + 7)
    return img, mask


def oracle_frame(img, mask, min_spot_size=3, max_sep=2.0, strong=None):
    """Everything the reference's worker would know about one frame, from the oracle."""
    if strong is None:
        strong = O.dispersion(img, mask)
    cc = O.cc2d(strong, img, min_spot_size)
    refl = O.cc2d_reflections(cc.k, cc.intensity, img.shape[1], img.shape[0], min_spot_size, max_sep)
    return strong, cc, refl


REFL_FIELDS = ["x_min", "x_max", "y_min", "y_max", "z_min", "z_max", "num_pixels",
               "peak_x", "peak_y", "peak_z", "peak_intensity", "sum_intensity"]
REFL_FLOATS = ["com_x", "com_y", "com_z", "peak_centroid_distance"]


def assert_reflections_equal(got, want, tol=0.0):
    assert len(got) == len(want), (len(got), len(want))
    for f in REFL_FIELDS:
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for f in REFL_FLOATS:
        if tol == 0.0:
            # bit-exact: compare the float32 bit patterns
            np.testing.assert_array_equal(got[f].view(np.uint32), want[f].view(np.uint32), err_msg=f)
        else:
            np.testing.assert_allclose(got[f], want[f], rtol=0, atol=tol, err_msg=f)


def assert_frame_matches_oracle(fr, img, mask, min_spot_size=3, max_sep=2.0, strong=None, precomputed=None):
    """precomputed: what oracle_frame() returned for this frame (batches that repeat a frame ask the oracle once)."""
    strong, cc, refl = precomputed if precomputed is not None else oracle_frame(img, mask, min_spot_size, max_sep, strong)
    if fr.strong_mask is not None:
        diff = np.argwhere(fr.strong_mask != strong)
        assert diff.size == 0, f"{len(diff)} strong-mask mismatches, first at (y,x)={diff[:5].tolist()}"
    assert fr.num_strong_pixels == cc.num_strong_pixels
    if fr.strong_k is not None:
        np.testing.assert_array_equal(fr.strong_k.astype(np.uint64), cc.k)
        np.testing.assert_array_equal(fr.strong_intensity, cc.intensity)
    assert fr.n_components == cc.n_unfiltered_boxes
    assert fr.num_strong_pixels_filtered == cc.num_strong_pixels_filtered
    assert len(fr.boxes) == len(cc.boxes)
    for f in ("l", "t", "r", "b", "num_pixels"):
        np.testing.assert_array_equal(fr.boxes[f], cc.boxes[f], err_msg=f)
    if fr.reflections is not None:
        # centroids: north_star tolerance is 1e-6; we hold them to bit-exact float32
        assert_reflections_equal(fr.reflections, refl.reflections)
        assert fr.n_filtered_size == refl.n_filtered_size
        assert fr.n_filtered_sep == refl.n_filtered_sep
    return strong, cc, refl
