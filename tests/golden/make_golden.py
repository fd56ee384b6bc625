#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):
    make -C oracle            # builds oracle/_ref/libffs_ref.so from the reference's standalone.cc
    python tests/golden/make_golden.py

What is recorded (data only -- inputs, or seeds + SHA-256 of inputs, and expected outputs):
  dispersion_small.npz     small u16/u32 frames stored verbatim + the reference's strong masks
  dispersion_config1.npz   BASELINE.json configs[0]: 10 x 1024^2 synthetic frames (seeded),
                           SHA-256 of each input, the reference's strong-pixel lists
  dispersion_samples.npz   the reference's own generated sample images 0..5 (h5read.c:203-276,
                           Eiger-16M + module-gap mask): SHA-256 of the images as produced by the
                           reference's generator (compiled into /tmp from h5read.c when HDF5 headers
                           are available), and the reference's strong-pixel lists for them
The expected outputs come from StandaloneSpotfinder<double>::standard_dispersion
(baseline/spotfinder/standalone.cc:258-270) through oracle/ref_shim.cc.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python"))
sys.path.insert(0, ROOT)

from ffs_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_mask(img, mask):
    H, W = img.shape
    return O.RefSpotfinder(W, H)(img.astype(np.float64), mask)


def small_cases():
    rng = np.random.default_rng(20240711)
    cases = {}

    def spotty(H, W, lam, n, dtype, peak):
        img = rng.poisson(lam, (H, W)).astype(np.int64)
        for _ in range(n):
            cy, cx = rng.integers(0, H), rng.integers(0, W)
            s = rng.uniform(0.7, 1.8)
            yy, xx = np.mgrid[max(cy - 6, 0):min(cy + 7, H), max(cx - 6, 0):min(cx + 7, W)]
            img[yy, xx] += rng.poisson(rng.uniform(10, peak) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)))
        return np.minimum(img, np.iinfo(dtype).max).astype(dtype)

    def mask(H, W, dead):
        m = np.ones((H, W), np.uint8)
        m[:, W // 2 - 2:W // 2 + 2] = 0
        m[H // 3:H // 3 + 3, :] = 0
        m[rng.integers(0, H, dead), rng.integers(0, W, dead)] = 0
        return m

    cases["u16_sparse_64x48"] = (spotty(48, 64, 0.2, 6, np.uint16, 300), mask(48, 64, 10))
    cases["u16_lambda2_131x97"] = (spotty(97, 131, 2.0, 15, np.uint16, 800), mask(97, 131, 30))
    cases["u16_bright_200x120"] = (spotty(120, 200, 400.0, 20, np.uint16, 30000), mask(120, 200, 50))
    sat = spotty(60, 90, 1.0, 5, np.uint16, 100)
    sat[[0, 0, 59, 59, 30], [0, 89, 0, 89, 45]] = 65535
    cases["u16_saturated_corners_90x60"] = (sat, np.ones((60, 90), np.uint8))
    cases["u16_all_masked_40x30"] = (spotty(30, 40, 3.0, 3, np.uint16, 100), np.zeros((30, 40), np.uint8))
    cases["u16_zero_frame_40x30"] = (np.zeros((30, 40), np.uint16), np.ones((30, 40), np.uint8))
    cases["u16_single_photons_50x50"] = ((rng.random((50, 50)) < 0.01).astype(np.uint16), np.ones((50, 50), np.uint8))
    big = spotty(70, 110, 50.0, 10, np.uint32, 2000000)
    big[5, 5] = (1 << 24) + 9
    big[40, 60] = 0xFFFFFFFF
    big[41, 60] = (1 << 24) - 1
    cases["u32_over_2p24_110x70"] = (big, mask(70, 110, 20))
    cases["u32_lambda5_100x80"] = (spotty(80, 100, 5.0, 12, np.uint32, 100000), mask(80, 100, 15))
    return cases


def reference_samples():
    """The reference's own sample images from its own generator (h5read.c), if it can be built."""
    src = os.path.join("/root/reference", "h5read", "src", "h5read.c")
    inc = os.path.join("/root/reference", "h5read", "include")
    if not (os.path.exists(src) and os.path.exists("/opt/conda/include/hdf5.h")):
        return None
    d = tempfile.mkdtemp(prefix="h5r_")
    prog = r'''
#include <stdio.h>
#include "h5read.h"
int main(void){ h5read_handle*h=h5read_generate_samples();
  size_t n=h5read_get_image_slow(h)*h5read_get_image_fast(h);
  for(int i=0;i<6;i++){ image_t*im=h5read_get_image(h,i); char fn[64]; sprintf(fn,"sample_%d.u16",i);
    FILE*f=fopen(fn,"wb"); fwrite(im->data,2,n,f); fclose(f);
    if(i==0){f=fopen("mask.u8","wb"); fwrite(im->mask,1,n,f); fclose(f);} h5read_free_image(im);} return 0; }
'''
    open(os.path.join(d, "gen.c"), "w").write(prog)
    subprocess.run(["gcc", "-O1", "-DHAVE_HDF5", "-I/opt/conda/include", f"-I{inc}", "gen.c", src,
                    "-L/opt/conda/lib", "-lhdf5", "-Wl,-rpath,/opt/conda/lib", "-o", "gen"],
                   cwd=d, check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["./gen"], cwd=d, check=True)
    imgs = [np.fromfile(os.path.join(d, f"sample_{i}.u16"), np.uint16).reshape(4362, 4148) for i in range(6)]
    mask = np.fromfile(os.path.join(d, "mask.u8"), np.uint8).reshape(4362, 4148)
    return imgs, mask


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref/libffs_ref.so missing: run `make -C oracle` where /root/reference exists")

    out = {}
    for name, (img, m) in small_cases().items():
        out[name + "/image"] = img
        out[name + "/mask"] = m
        out[name + "/strong"] = np.packbits(ref_mask(img, m), axis=None)
    np.savez_compressed(os.path.join(HERE, "dispersion_small.npz"), **out)

    p = synth.config1_params()
    m = synth.config1_mask()
    out = {"mask_sha256": np.array(sha(m))}
    for i in range(10):
        img = synth.frame(p, i)
        strong = ref_mask(img, m)
        out[f"frame{i}/input_sha256"] = np.array(sha(img))
        out[f"frame{i}/strong_k"] = np.flatnonzero(strong).astype(np.uint32)
    np.savez_compressed(os.path.join(HERE, "dispersion_config1.npz"), **out)

    rs = reference_samples()
    out = {}
    if rs is not None:
        imgs, mask = rs
        out["generated_by"] = np.array("reference h5read_generate_samples() compiled from h5read/src/h5read.c")
    else:
        imgs = [synth.reference_sample(i) for i in range(6)]
        mask = synth.mask_eiger16m()
        out["generated_by"] = np.array("ffs_synth_reference_sample (restatement)")
    out["mask_sha256"] = np.array(sha(mask))
    for i, img in enumerate(imgs):
        strong = ref_mask(img, mask)
        out[f"sample{i}/input_sha256"] = np.array(sha(img))
        out[f"sample{i}/strong_k"] = np.flatnonzero(strong).astype(np.uint32)
        print(f"sample {i}: {int(strong.sum())} strong")
    np.savez_compressed(os.path.join(HERE, "dispersion_samples.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
