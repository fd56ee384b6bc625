"""Access to the committed golden fixtures (tests/golden/*.npz, generated from the reference's
own standalone.cc by tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def small_cases():
    z = np.load(os.path.join(GOLD, "dispersion_small.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    for n in names:
        img, mask = z[n + "/image"], z[n + "/mask"]
        strong = np.unpackbits(z[n + "/strong"])[: img.size].reshape(img.shape)
        yield n, img, mask, strong


def config1():
    from ffs_amd import synth
    z = np.load(os.path.join(GOLD, "dispersion_config1.npz"))
    p, mask = synth.config1_params(), synth.config1_mask()
    assert sha(mask) == str(z["mask_sha256"])
    for i in range(10):
        img = synth.frame(p, i)
        assert sha(img) == str(z[f"frame{i}/input_sha256"]), "synthetic generator drifted from the fixture"
        yield i, img, mask, z[f"frame{i}/strong_k"]


def samples(indices=range(6)):
    from ffs_amd import synth
    z = np.load(os.path.join(GOLD, "dispersion_samples.npz"))
    mask = synth.mask_eiger16m()
    assert sha(mask) == str(z["mask_sha256"])
    for i in indices:
        img = synth.reference_sample(i)
        assert sha(img) == str(z[f"sample{i}/input_sha256"])
        yield i, img, mask, z[f"sample{i}/strong_k"]
