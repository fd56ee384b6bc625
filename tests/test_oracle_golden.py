"""CPU: the oracle restatement (oracle/ffs_oracle.c) against golden vectors produced by the
reference's own baseline/spotfinder/standalone.cc, and -- when oracle/_ref is built -- against the
compiled reference directly.  This is what pins the oracle."""
import numpy as np
import pytest

import golden_util as G
from oracle import oracle as O


@pytest.mark.parametrize("name,img,mask,strong", list(G.small_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_small_golden(name, img, mask, strong):
    got = O.dispersion(img, mask)
    np.testing.assert_array_equal(got, strong)


def test_config1_golden():
    total = 0
    for i, img, mask, strong_k in G.config1():
        got = np.flatnonzero(O.dispersion(img, mask)).astype(np.uint32)
        np.testing.assert_array_equal(got, strong_k, err_msg=f"frame {i}")
        total += len(got)
    assert total > 5000


@pytest.mark.parametrize("idx", [2, 5])
def test_reference_sample_images_golden(idx):
    """The reference's own generated sample images (h5read.c:203-276) + module-gap mask."""
    for i, img, mask, strong_k in G.samples([idx]):
        got = np.flatnonzero(O.dispersion(img, mask)).astype(np.uint32)
        np.testing.assert_array_equal(got, strong_k)
        assert len(got) > 1000


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(6))
def test_port_equals_compiled_reference(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(20, 200)), int(rng.integers(20, 300))
    lam = float(rng.choice([0.05, 1.0, 7.0, 200.0, 5000.0]))
    img = rng.poisson(lam, (H, W)).astype(np.uint32)
    for _ in range(10):
        y, x = rng.integers(0, H), rng.integers(0, W)
        img[y, x] += rng.integers(10, 60000)
    if seed % 2:
        img[rng.integers(0, H), rng.integers(0, W)] = (1 << 24) + 17
    mask = (rng.random((H, W)) > 0.03).astype(np.uint8)
    want = O.RefSpotfinder(W, H)(img, mask)
    np.testing.assert_array_equal(O.dispersion(img, mask), want)
    np.testing.assert_array_equal(O.dispersion(img.astype(np.float64), mask), want)


def test_persistent_context_matches_one_shot():
    rng = np.random.default_rng(3)
    img = rng.poisson(2.0, (60, 80)).astype(np.uint16)
    img[30, 40] = 900
    mask = np.ones((60, 80), np.uint8)
    ps = O.PortSpotfinder(80, 60)
    dst = np.empty((60, 80), np.uint8)
    for _ in range(2):
        ps.run_f64(img.astype(np.float64), mask, dst)
        np.testing.assert_array_equal(dst, O.dispersion(img, mask))
