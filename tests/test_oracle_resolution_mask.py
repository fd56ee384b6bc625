"""CPU: the oracle's restatement of the resolution mask (spotfinder/kernels/masking.cu:37-73,99-147) against
known answers and an independent float64 formulation."""
import numpy as np


GEOM = dict(wavelength=0.976, distance=0.15, beam_center_x=2074.3, beam_center_y=2181.7,
            pixel_size_x=75e-6, pixel_size_y=75e-6)


def d_spacing_f64(W, H, g):
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    dx = (x + 0.5 - g["beam_center_x"]) * g["pixel_size_x"]
    dy = (y + 0.5 - g["beam_center_y"]) * g["pixel_size_y"]
    r = np.hypot(dx, dy)
    with np.errstate(divide="ignore"):
        return g["wavelength"] / (2.0 * np.sin(0.5 * np.arctan(r / g["distance"])))


def test_known_answers_and_float64_agreement():
    from oracle import oracle as O
    W, H = 600, 500
    g = dict(GEOM, beam_center_x=300.0, beam_center_y=250.0)
    mask = np.ones((H, W), np.uint8)
    mask[10:20, 30:40] = 0                                   # already masked pixels stay masked
    out, res = O.resolution_mask(mask, dmin=4.0, dmax=30.0, **g)
    want = d_spacing_f64(W, H, g)
    # float32 evaluation of a well-conditioned formula: a few ulp of float32 at most
    rel = np.abs(res.astype(np.float64) - want) / want
    assert rel.max() < 8 * 2.0 ** -24
    # a pixel 1000 px from the centre along x: r = 1000.5 px * 75 um -> d = lambda / (2 sin(atan(r / D) / 2))
    r = (599 + 0.5 - 300.0) * 75e-6
    assert abs(res[250, 599] - 0.976 / (2 * np.sin(0.5 * np.arctan(np.hypot(r, 0.5 * 75e-6) / 0.15)))) < 1e-4
    inside = (want >= 4.0) & (want <= 30.0)
    sure = (np.abs(want - 4.0) > 1e-4) & (np.abs(want - 30.0) > 1e-3)   # away from the two thresholds
    assert np.array_equal(out[sure] != 0, (inside & (mask != 0))[sure])
    assert (out[10:20, 30:40] == 0).all()
    assert 0 < out.sum() < mask.sum()
    # each limit only if > 0
    only_min, _ = O.resolution_mask(mask, dmin=4.0, **g)
    only_max, _ = O.resolution_mask(mask, dmax=30.0, **g)
    assert np.array_equal(out, only_min & only_max)
    none, _ = O.resolution_mask(mask, **g)
    assert np.array_equal(none, mask)
