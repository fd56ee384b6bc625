#!/usr/bin/env python3
"""Soak of the sparse stage in small workgroups (csrc/kernels_band.hpp): random frame shapes, band heights (tuning target_waves /
band_taper: one band, many, sub-bands, bands of two heights), pixel widths, masks, batch sizes and contents -- sparse spots, bars
along and across the band boundaries, row-wrap pairs, saturated cores -- against the oracle, every frame, twice per stream.
Counts which launches ran (a data set beyond a band's plan falls back inside ffs_wait: still compared).
    python tools/soak_bands.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffs_amd
from util import assert_frame_matches_oracle

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, took, t0 = [], {"bands": 0, "fallback": 0, "other": 0}, time.time()
for k, seed in enumerate(range(first, first + n)):
    rng = np.random.default_rng(9100 + seed)
    W, H = int(rng.integers(260, 2100)), int(rng.integers(150, 1300))
    dt = np.uint32 if rng.random() < 0.25 else np.uint16
    B = int(rng.integers(1, 5))
    hi = 60000 if dt == np.uint16 else 800000
    frames = []
    for _ in range(B):
        img = rng.poisson(float(rng.choice([0.5, 2.0, 5.0])), (H, W)).astype(dt)
        for _ in range(int(W * H / rng.choice([3000, 8000, 20000]))):
            y, x = rng.integers(0, H - 4), rng.integers(0, W - 6)
            img[y:y + rng.integers(1, 5), x:x + rng.integers(1, 7)] = rng.integers(150, 4000)
        for _ in range(int(rng.integers(0, 4))):                      # bars across many rows (and band boundaries)
            x, y0 = rng.integers(0, W), rng.integers(0, H - 20)
            img[y0:y0 + rng.integers(10, 120), x] = rng.integers(300, 2000)
        for _ in range(int(rng.integers(0, 3))):                      # bars along a row
            y, x0 = rng.integers(0, H), rng.integers(0, W - 30)
            img[y, x0:x0 + rng.integers(8, 200)] = rng.integers(300, 2000)
        for _ in range(int(rng.integers(0, 6))):                      # row-wrap pairs
            y = rng.integers(0, H - 1)
            img[y, W - 1] = 900; img[y + 1, 0] = 850
        for _ in range(int(rng.integers(0, 5))):                      # saturated cores
            y, x = rng.integers(0, H - 7), rng.integers(0, W - 7)
            img[y:y + 6, x:x + 6] = rng.integers(hi // 3, hi)
        frames.append(img)
    frames = np.stack(frames)
    mask = np.ones((H, W), np.uint8)
    if rng.random() < 0.6:
        mask[rng.random((H, W)) < 0.002] = 0
        c0 = int(rng.integers(0, W - 6)); mask[:, c0:c0 + 4] = 0
    tuning = dict(sparse_bands=2)
    r = rng.random()
    if r < 0.35:
        tuning["target_waves"] = int(rng.choice([4, 12, 30, 80, 300]))
    elif r < 0.45 and H >= 2304 // 2:
        tuning["band_taper"] = int(rng.choice([40, 60]))
    prm = dict(min_spot_size=int(rng.choice([1, 3, 5])), max_peak_centroid_separation=float(rng.choice([2.0, 0.0, 4.0])))
    try:
        ctx = ffs_amd.Context(W, H, dt, max_batch=B)
        ctx.set_tuning(**tuning)
        ctx.set_mask(mask)
        ctx.set_params(want_strong_list=0, want_strong_mask=0, want_reflections=1, **prm)
        st = ctx.stream()
        want = None
        for rep in range(2):
            res = st.process(frames, first_frame_id=rep * 100)
            path, reruns = st.last_path()
            took["bands" if "bands" in path else ("fallback" if reruns else "other")] += 1
            if want is None:
                want = [assert_frame_matches_oracle(fr, img, mask, min_spot_size=prm["min_spot_size"], max_sep=prm["max_peak_centroid_separation"]) for fr, img in zip(res, frames)]
            else:
                for fr, img, w in zip(res, frames, want):
                    assert_frame_matches_oracle(fr, img, mask, min_spot_size=prm["min_spot_size"], max_sep=prm["max_peak_centroid_separation"], precomputed=w)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("FAIL", seed, (W, H, B, np.dtype(dt).name, tuning, prm), repr(e)[:300], flush=True)
        continue
    if k % 20 == 19:
        print(f"{k + 1} seeds, {len(bad)} failures, batches by launch {took}, {time.time() - t0:.0f} s", flush=True)
print("done:", n, "seeds,", len(bad), "failures", bad, "batches by launch", took)
sys.exit(1 if bad else 0)
