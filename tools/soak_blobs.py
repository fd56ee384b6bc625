#!/usr/bin/env python3
"""Soak: frames of fat spots (tens of thousands of strong pixels in thousands of runs) of random shape, mask, algorithm and
filters through the dense paths -- the run-based one-launch sparse stage, its overflow fall-back, the four-pixel final pass of
the extended algorithm -- against the oracle.   python tools/soak_blobs.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffs_amd
from oracle import oracle as O
from util import assert_frame_matches_oracle

def blob_frame(rng, W, H, n_blobs, rmax):
    img = rng.poisson(float(rng.choice([0.3, 1.0, 4.0])), (H, W)).astype(np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n_blobs):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, rmax + 1)
        y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, H), max(cx - r, 0), min(cx + r + 1, W)
        sel = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
        hi = int(rng.choice([300, 3000, 60000]))
        img[y0:y1, x0:x1][sel] = rng.integers(hi // 3, hi, sel.sum()).astype(np.uint16)
    if rng.random() < 0.3:
        img[rng.integers(0, H), :] = 900                    # a whole row
    return img

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad, t0 = [], time.time()
for k, seed in enumerate(range(first, first + n)):
    rng = np.random.default_rng(7000 + seed)
    W, H = int(rng.integers(300, 2100)), int(rng.integers(200, 1300))
    area = W * H
    rmax = int(rng.integers(2, 9))
    n_blobs = int(area / (3.0 * rmax * rmax) * rng.uniform(0.15, 0.6))
    B = int(rng.integers(1, 4))
    frames = np.stack([blob_frame(rng, W, H, n_blobs, rmax) for _ in range(B)])
    mask = np.ones((H, W), np.uint8)
    if rng.random() < 0.6:
        mask[rng.random((H, W)) < 0.003] = 0
        mask[:, int(rng.integers(0, W - 4)):][:, :3] = 0
    algo = int(rng.random() < 0.5)
    prm = dict(min_spot_size=int(rng.choice([1, 3, 6])), max_peak_centroid_separation=float(rng.choice([2.0, 0.0, 5.0])))
    tuning = [None, dict(chain_runs=2), dict(chain_runs=0)][seed % 3]
    try:
        ctx = ffs_amd.Context(W, H, np.uint16, max_batch=B, max_strong_per_frame=min(area, 400000))
        if tuning:
            ctx.set_tuning(**tuning)
        ctx.set_mask(mask)
        want_list = int(rng.random() < 0.5)    # 0: the default -- the dense launches keep their pixel lists in LDS (need_lists = 0)
        ctx.set_params(algorithm=algo, want_strong_mask=int(rng.random() < 0.5) if want_list else 0, want_strong_list=want_list, want_reflections=1, **prm)
        st = ctx.stream()
        p = O.DispParams()
        O.lib().ffs_oracle_default_disp_params(O.C.byref(p))
        want = [O.dispersion_extended(f, mask, p) if algo else O.dispersion(f, mask, p) for f in frames]
        for rep in range(3):          # (the second batch of a stream is the one that knows the data is dense)
            res = st.process(frames, first_frame_id=rep)
            for fr, img, strong in zip(res, frames, want):
                assert_frame_matches_oracle(fr, img, mask, min_spot_size=prm["min_spot_size"], max_sep=prm["max_peak_centroid_separation"], strong=strong)
        ns = [int(w.sum()) for w in want]
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("FAIL", seed, (W, H, B, algo, tuning, prm), repr(e)[:300], flush=True)
        continue
    print(f"seed {seed}: {W}x{H} x{B} algo {algo} tuning {tuning} strong {ns} ok ({time.time() - t0:.0f} s)", flush=True)
print("done:", n, "seeds,", len(bad), "failures", bad)
sys.exit(1 if bad else 0)
