"""CPU: the oracle's extended dispersion (restated from baseline/spotfinder/baseline.cpp:325-776)
against an independent numpy/scipy formulation of the same published algorithm -- window sums by
uniform_filter-style box sums instead of a summed-area table, erosion by maximum/minimum filters
instead of a chamfer distance transform.  "parity unpinned" by the reference itself (its class needs
DIALS); this is the cross-check that the restatement says what the text says."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import oracle as O
from util import make_frame


def box_sum(a, r):
    """Sum over the (2r+1)^2 window clipped to the image."""
    a = a.astype(np.float64)
    k = np.ones((2 * r + 1, 2 * r + 1))
    return ndimage.convolve(a, k, mode="constant", cval=0.0)


def numpy_extended(img, mask, min_count=2, nsig_b=6.0, nsig_s=3.0, threshold=0.0, flavour=0, max_valid=-1):
    src = img.astype(np.float64)
    valid = (mask != 0) & (src < (1 << 24))
    m = box_sum(valid, 3)
    x = box_sum(valid * src, 3)
    y = box_sum(valid * src * src, 3)
    a = m * y - x * x - x * (m - 1)
    with np.errstate(invalid="ignore"):
        c = x * nsig_b * np.sqrt(2 * (m - 1))
        first = (mask != 0) & (m >= min_count) & (a > c)
    if max_valid >= 0:
        first &= src <= max_valid
    if flavour == 0:
        # distance to the nearest pixel that is not "first" (masked pixels are not): eroded away
        # when that distance is <= 2; pixels outside the image do not count
        near_other = ndimage.minimum_filter(first.astype(np.uint8), size=5, mode="constant", cval=1) == 0
    else:
        bg_valid = (~first) & (mask != 0)
        near_other = ndimage.maximum_filter(bg_valid.astype(np.uint8), size=5, mode="constant", cval=0) == 1
    signal_region = first & ~near_other
    bg = (mask != 0) & ~signal_region
    bgv = bg & (src < (1 << 24))
    m2 = box_sum(bgv, 5)
    x2 = box_sum(bgv * src, 5)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = np.where(m2 >= 2, x2 / np.maximum(m2, 1), 0.0)
        strong = signal_region & (src > threshold) & (src >= mean + nsig_s * np.sqrt(mean))
    if flavour == 1:
        strong &= m2 > 0
    if max_valid >= 0:
        strong &= src <= max_valid
    return strong.astype(np.uint8), first.astype(np.uint8), signal_region.astype(np.uint8)


CASES = [
    dict(W=97, H=61, dtype=np.uint16, seed=1, n_spots=12),
    dict(W=200, H=150, dtype=np.uint16, seed=2, n_spots=60, masked=True),
    dict(W=130, H=90, dtype=np.uint32, seed=3, n_spots=20, masked=True),
    dict(W=64, H=64, dtype=np.uint16, seed=4, n_spots=200),          # crowded: blobs merge, deep erosion
    dict(W=40, H=9, dtype=np.uint16, seed=5, n_spots=4),             # shorter than the 11x11 window
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("flavour", [0, 1])
def test_extended_matches_independent_formulation(case, flavour):
    img, mask = make_frame(**case)
    got, first, eroded = O.dispersion_extended(img, mask, flavour=flavour, debug=True)
    want, wfirst, wsignal = numpy_extended(img, mask, flavour=flavour)
    assert np.array_equal(first, wfirst)
    assert np.array_equal(eroded, wsignal)
    assert np.array_equal(got, want)
    if case["n_spots"] in (12, 20, 60):
        assert got.sum() > 0
    assert not (got & ~first).any() and not (got & (mask == 0)).any()


def test_extended_parameters_and_max_valid():
    img, mask = make_frame(W=160, H=120, dtype=np.uint16, seed=9, n_spots=40, masked=True)
    p = O.DispParams()
    O.lib().ffs_oracle_default_disp_params(O.C.byref(p))
    p.min_count = 3
    p.nsig_b = 4.0
    p.nsig_s = 2.5
    got = O.dispersion_extended(img, mask, p, flavour=1, max_valid=2000.0)
    want, _, _ = numpy_extended(img, mask, min_count=3, nsig_b=4.0, nsig_s=2.5, flavour=1, max_valid=2000)
    assert np.array_equal(got, want)
    assert not (got & (img > 2000)).any()


def test_extended_finds_more_of_each_spot_than_standard():
    """The point of the algorithm: a background estimate that excludes the spot itself, so weak
    shoulders of strong spots pass.  Every standard-dispersion pixel well inside a spot stays."""
    img, mask = make_frame(W=300, H=200, dtype=np.uint16, seed=11, n_spots=50)
    std = O.dispersion(img, mask)
    ext = O.dispersion_extended(img, mask)
    assert ext.sum() > 0 and std.sum() > 0
    assert ext.sum() >= 0.8 * std.sum()


def test_all_masked_and_flat_images():
    img = np.full((30, 40), 5, np.uint16)
    assert O.dispersion_extended(img, np.ones_like(img, np.uint8)).sum() == 0
    assert O.dispersion_extended(img, np.zeros_like(img, np.uint8)).sum() == 0
    for fl in (0, 1):
        assert O.dispersion_extended(np.zeros((12, 12), np.uint16), np.ones((12, 12), np.uint8), flavour=fl).sum() == 0
