"""GPU: the HIP path against the committed golden vectors (expected outputs produced by the
reference's own standalone.cc; see tests/golden/make_golden.py)."""
import numpy as np
import pytest

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,img,mask,strong", list(G.small_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_small_golden(ffs, name, img, mask, strong):
    H, W = img.shape
    ctx = ffs.Context(W, H, img.dtype)
    ctx.set_mask(mask)
    ctx.set_params(want_strong_mask=1, want_strong_list=1)
    fr = ctx.stream().process(img)[0]
    np.testing.assert_array_equal(fr.strong_mask, strong)
    np.testing.assert_array_equal(fr.strong_k, np.flatnonzero(strong).astype(np.uint32))
    np.testing.assert_array_equal(fr.strong_intensity, img.reshape(-1)[fr.strong_k].astype(np.uint32))


def test_config1_golden(ffs):
    cases = list(G.config1())
    frames = np.stack([c[1] for c in cases])
    ctx = ffs.Context(1024, 1024, np.uint16, max_batch=10)
    ctx.set_mask(cases[0][2])
    ctx.set_params(want_strong_list=1)
    res = ctx.stream().process(frames)
    for fr, (i, img, mask, strong_k) in zip(res, cases):
        np.testing.assert_array_equal(fr.strong_k, strong_k, err_msg=f"frame {i}")


def test_reference_sample_images_golden(ffs):
    """All six generated sample images of the reference (Eiger-16M, module-gap mask) in one batch."""
    cases = list(G.samples(range(6)))
    frames = np.stack([c[1] for c in cases])
    ctx = ffs.Context(4148, 4362, np.uint16, max_batch=6)
    ctx.set_mask(cases[0][2])
    ctx.set_params(want_strong_list=1, want_strong_mask=1)
    res = ctx.stream().process(frames)
    for fr, (i, img, mask, strong_k) in zip(res, cases):
        np.testing.assert_array_equal(fr.strong_k, strong_k, err_msg=f"sample {i}")
        assert int(fr.strong_mask.sum()) == len(strong_k)
        assert not fr.strong_mask[mask == 0].any()
