"""CPU: the C-ABI library loads and exports every symbol include/ffs_hip.h declares (no compute
calls without a GPU), the synthetic generator is deterministic, and the reference's sample images
are reproduced byte for byte."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ffs_[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol(ffs):
    from ffs_amd import api
    lib = ffs.load_library()
    declared = _declared("ffs_hip.h")
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"libffs_hip.so lacks {missing}"
    assert sorted(api.EXPORTS) == declared, "python binding list out of sync with the header"


def test_hip_library_exports_no_other_function():
    """Built with -fvisibility=hidden: the C ABI of include/ffs_hip.h is every FUNCTION the library exports (the kernels'
    launch handles are data symbols the HIP runtime registers)."""
    import subprocess
    so = os.path.join(ROOT, "fast-feedback-service_amd", "libffs_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    funcs = sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "Tt")
    assert funcs == _declared("ffs_hip.h"), [f for f in funcs if not f.startswith("ffs_")]


def test_product_library_has_no_experiment_switch():
    """The timing experiments that break results (phases switched off, kernels stopped half way) are compiled only with
    -DFFS_EXPERIMENTS (make experiments -> libffs_hip_exp.so): the product library neither reads their environment
    variables nor carries their kernel."""
    so = open(os.path.join(ROOT, "fast-feedback-service_amd", "libffs_hip.so"), "rb").read()
    for needle in (b"FFS_EXP_", b"FFS_K1_DEBUG", b"FFS_CHAIN_SKIP", b"FFS_CHAIN_STOP", b"FFS_DUMMY", b"k_dummy_spin",
                   b"FFS_K1_VARIANT", b"FFS_CCL", b"FFS_SCHED", b"FFS_EMIT"):
        assert needle not in so, needle


def test_synth_library_exports_every_declared_symbol():
    from ffs_amd import synth
    lib = synth.lib()
    for s in _declared("ffs_synth.h"):
        assert hasattr(lib, s), s


def test_no_gpu_means_loud_failure(ffs):
    """Without a GPU the product path must fail loudly, not fall back to the CPU."""
    if ffs.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ffs.FfsError) as e:
        ffs.Context(64, 64)
    assert "no CPU fallback" in str(e.value) or e.value.code == -5


def test_default_params_are_the_oracles(ffs):
    from ffs_amd import api
    p = api.default_params()
    assert (p.min_count, p.nsig_b, p.nsig_s, p.threshold) == (2, 6.0, 3.0, 0.0)   # standalone.cc:16-20
    assert (p.min_spot_size, p.min_spot_size_3d) == (3, 3)                        # spotfinder.cc:318-328
    assert p.max_peak_centroid_separation == 2.0 and p.max_valid == -1


def test_synth_is_deterministic_and_pinned():
    from ffs_amd import synth
    p = synth.config1_params()
    a, b = synth.frame(p, 3), synth.frame(p, 3)
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a, synth.frame(p, 4))
    list(G.config1())   # asserts SHA-256 of all ten frames against the fixture
    par = synth.frames(p, [3, 4, 5], threads=3)
    np.testing.assert_array_equal(par[0], a)


def test_reference_sample_images_byte_exact():
    """SHA-256 recorded from the reference's own h5read_generate_samples() output."""
    for i, img, mask, strong_k in G.samples(range(6)):
        assert img.shape == (4362, 4148)


def test_sweep_spots_persist_across_frames():
    from ffs_amd import synth
    p = synth.sweep_params(seed=9, n_frames=8, n_spots=30, width=200, height=150)
    fr = synth.frames(p, range(8), threads=4).astype(np.int64)
    bright = (fr > 40)
    assert bright.any()
    # a bright pixel is usually bright on the neighbouring frame too (rocking curve)
    both = (bright[:-1] & bright[1:]).sum()
    assert both > 0
