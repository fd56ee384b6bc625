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


@pytest.mark.parametrize("workload,algorithm,frame", [("eiger16m", "dispersion", 0), ("eiger16m", "dispersion", 31),
                                                      ("eiger16m", "dispersion_extended", 7), ("jungfrau9m", "dispersion", 19)])
def test_bench_workload_fixture(workload, algorithm, frame):
    """tests/golden/bench_workloads.npz (what bench.py and tests/test_gpu_bench_config.py hold the HIP path to, generated with
    the reference's standalone.cc as the threshold): the restatement reproduces a frame's entry -- input hash, counts and
    the digest of its boxes and reflections -- so the fixture and the generator cannot drift apart unnoticed."""
    import hashlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from ffs_amd import fixtures, synth
    exp = fixtures.load_expected(workload, algorithm, 0, 32)
    assert exp is not None
    z = np.load(os.path.normpath(fixtures.GOLDEN))
    p = synth.eiger16m_params(seed=2000) if workload == "eiger16m" else synth.jungfrau9m_params(seed=4000)
    _, mask = bench.make_inputs(workload, 0, 0)
    img = synth.frame(p, frame)
    assert hashlib.sha256(img.tobytes()).digest() == z[fixtures.key(workload, algorithm, 0) + "/input_sha256"][frame].tobytes()
    strong = O.dispersion_extended(img, mask) if algorithm == "dispersion_extended" else O.dispersion(img, mask)
    cc = O.cc2d(strong, img, 3)
    refl = O.cc2d_reflections(cc.k, cc.intensity, img.shape[1], img.shape[0], 3, 2.0)
    assert (cc.num_strong_pixels, len(cc.boxes), cc.n_unfiltered_boxes, len(refl.reflections)) == (
        exp["num_strong_pixels"][frame], exp["n_boxes"][frame], exp["n_components"][frame], exp["n_reflections"][frame])
    assert fixtures.frame_digest(cc.boxes, refl.reflections) == exp["digest"][frame].tobytes()


def test_sweep_fixture_frames():
    """The rotation-sweep entry of tests/golden/bench_workloads.npz (BASELINE.json configs[4], bench.py --workload sweep16m): two of
    its 100 frames through the restatement give the committed strong-pixel and box counts, and the table's counts are consistent
    (found = calculated - filtered) -- the whole sweep is re-derived on the GPU box by tests/test_gpu_fullsize.py."""
    from ffs_amd import fixtures, synth
    exp = fixtures.load_expected_sweep("sweep16m")
    assert exp is not None and len(exp["num_strong_pixels"]) == 100 and len(exp["digest"]) == 32
    assert exp["n_reflections"] == exp["n_calculated"] - exp["n_filtered_size"] - exp["n_filtered_sep"] and exp["n_reflections"] > 300
    p = synth.sweep_params(seed=5000, n_frames=100, n_spots=800)
    mask = synth.mask_eiger16m()
    for z in (3, 57):
        img = synth.frame(p, z)
        cc = O.cc2d(O.dispersion(img, mask), img, 3)
        assert (cc.num_strong_pixels, len(cc.boxes)) == (exp["num_strong_pixels"][z], exp["n_boxes"][z])
