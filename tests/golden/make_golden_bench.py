#!/usr/bin/env python3
"""Generates tests/golden/bench_workloads.npz: what bench.py's own timed batches must produce.

For every workload bench.py times (eiger16m / dispersion, eiger16m / dispersion_extended, jungfrau9m / dispersion), every rank
0..7 (rank r's frames have seed base + 1000 r, bench.make_inputs) and every one of the 32 unique frames of a step: the SHA-256
of the input frame, num_strong_pixels, n_components, n_boxes, n_reflections and the digest (ffs_amd.fixtures.frame_digest) of
the boxes and reflections -- all from the ORACLE: the threshold of the standard algorithm by the reference's own standalone.cc
(oracle/_ref, compiled from /root/reference where it lies; falls back to the restatement, which tests/test_oracle_golden.py
holds to it), the extended algorithm by the restatement of baseline.cpp (DIALS absent: unpinned), connected components and
reflections by the restatement (Boost.Graph absent: unpinned).

Also the rotation sweep of bench.py --workload sweep16m (BASELINE.json configs[4]): per frame the strong pixels and boxes, and
the 3D reflection table's counts and digest (--only sweep16m).

Run in the build container:   make oracle synth && python tests/golden/make_golden_bench.py [--ranks 8]
About 25 CPU-minutes for all of it on 8 cores.  Data only: seeds, hashes, counts, digests.
"""
import argparse
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python"))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (make_inputs: the frames bench.py times)
from ffs_amd import fixtures  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [("eiger16m", "dispersion"), ("eiger16m", "dispersion_extended"), ("jungfrau9m", "dispersion")]
N_FRAMES = 56     # (covers 32, 48 and 56 frames per step: bench.py --batch)


def one_frame(img, mask, algorithm, use_ref):
    H, W = img.shape
    if algorithm == "dispersion_extended":
        strong = O.dispersion_extended(img, mask)
    elif use_ref:
        strong = O.RefSpotfinder(W, H)(img, mask)
    else:
        strong = O.dispersion(img, mask)
    cc = O.cc2d(strong, img, 3)
    refl = O.cc2d_reflections(cc.k, cc.intensity, W, H, 3, 2.0)
    return (hashlib.sha256(img.tobytes()).digest(), cc.num_strong_pixels, cc.n_unfiltered_boxes, len(cc.boxes),
            len(refl.reflections), fixtures.frame_digest(cc.boxes, refl.reflections))


def sweep_case(out, threads, use_ref):
    """BASELINE.json configs[4] (bench.py --workload sweep16m, tests/3d_connected_components.sh:27-37's shape): the 100-frame
    Eiger-16M sweep of seed 5000 -- per frame the oracle's strong pixels and boxes (min_spot_size 3), then the oracle's 3D
    labelling of all slices (min_spot_size_3d 15, separation 2.0): counts and the digest of the reflection table."""
    from ffs_amd import synth
    W, H, NZ, B = 4148, 4362, 100, 25
    p = synth.sweep_params(seed=5000, n_frames=NZ, n_spots=800)
    mask = synth.mask_eiger16m()
    slices, ns, nbx = [], [], []
    t0 = time.time()

    def one(img):
        strong = O.RefSpotfinder(W, H)(img, mask) if use_ref else O.dispersion(img, mask)
        cc = O.cc2d(strong, img, 3)
        return cc.k, cc.intensity, cc.num_strong_pixels, len(cc.boxes)
    for z0 in range(0, NZ, B):
        frames = synth.frames(p, range(z0, z0 + B), threads=threads)
        with ThreadPoolExecutor(threads) as ex:
            rows = list(ex.map(one, frames))
        slices += [(r[0], r[1]) for r in rows]
        ns += [r[2] for r in rows]
        nbx += [r[3] for r in rows]
        print(f"sweep16m: frames {z0}..{z0 + B - 1} done, {time.time() - t0:.0f} s", flush=True)
    want = O.cc3d(slices, W, H, 15, 2.0)
    out["sweep16m/n_reflections"] = np.array(len(want.reflections), np.uint32)
    out["sweep16m/n_calculated"] = np.array(want.n_calculated, np.uint32)
    out["sweep16m/n_filtered_size"] = np.array(want.n_filtered_size, np.uint32)
    out["sweep16m/n_filtered_sep"] = np.array(want.n_filtered_sep, np.uint32)
    out["sweep16m/digest"] = np.frombuffer(fixtures.reflections_digest(want.reflections), np.uint8)
    out["sweep16m/num_strong_pixels"] = np.array(ns, np.uint32)
    out["sweep16m/n_boxes"] = np.array(nbx, np.uint32)
    print(f"sweep16m: {len(want.reflections)} reflections of {want.n_calculated} calculated "
          f"({want.n_filtered_size} below the size, {want.n_filtered_sep} beyond the separation), {sum(ns)} strong pixels", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--only", default="", help="workload/algorithm, e.g. eiger16m/dispersion")
    args = ap.parse_args()
    path = os.path.join(HERE, "bench_workloads.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    use_ref = O.have_ref()
    out["threshold_oracle"] = np.array("reference standalone.cc (oracle/_ref)" if use_ref else "oracle restatement")
    if not args.only or args.only == "sweep16m":
        sweep_case(out, args.threads, use_ref)
        np.savez_compressed(path, **out)
    for workload, algorithm in CASES:
        if args.only and args.only != f"{workload}/{algorithm}":
            continue
        for rank in range(args.ranks):
            t0 = time.time()
            frames, mask = bench.make_inputs(workload, N_FRAMES, rank)
            with ThreadPoolExecutor(args.threads) as ex:
                rows = list(ex.map(lambda img: one_frame(img, mask, algorithm, use_ref), frames))
            k = fixtures.key(workload, algorithm, rank)
            out[k + "/input_sha256"] = np.frombuffer(b"".join(r[0] for r in rows), np.uint8).reshape(N_FRAMES, 32)
            out[k + "/num_strong_pixels"] = np.array([r[1] for r in rows], np.uint32)
            out[k + "/n_components"] = np.array([r[2] for r in rows], np.uint32)
            out[k + "/n_boxes"] = np.array([r[3] for r in rows], np.uint32)
            out[k + "/n_reflections"] = np.array([r[4] for r in rows], np.uint32)
            out[k + "/digest"] = np.frombuffer(b"".join(r[5] for r in rows), np.uint8).reshape(N_FRAMES, 32)
            print(f"{k}: {N_FRAMES} frames, {int(out[k + '/n_boxes'].sum())} boxes, "
                  f"{int(out[k + '/num_strong_pixels'].sum())} strong pixels, {time.time() - t0:.0f} s", flush=True)
            np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
