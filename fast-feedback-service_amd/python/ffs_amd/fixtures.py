"""Per-frame result digests: what bench.py and the parity tests compare with tests/golden/bench_workloads.npz.

A digest is the SHA-256 over the named fields of a frame's boxes and reflections in a fixed order (integers and the
float32 bit patterns as they are), so that one committed 32-byte value per frame pins every number the C ABI returns for
it.  The expected values are generated from the oracle by tests/golden/make_golden_bench.py; nothing here imports it.
(The reference's own self-check compares per frame too: spotfinder/spotfinder.cc:1012-1053.)"""
from __future__ import annotations

import hashlib
import os

import numpy as np

BOX_FIELDS = ("l", "t", "r", "b", "num_pixels")
REFL_FIELDS = ("x_min", "x_max", "y_min", "y_max", "z_min", "z_max", "num_pixels", "com_x", "com_y", "com_z",
               "peak_x", "peak_y", "peak_z", "peak_intensity", "peak_centroid_distance", "sum_intensity")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden", "bench_workloads.npz")


def frame_digest(boxes: np.ndarray, reflections: np.ndarray | None) -> bytes:
    h = hashlib.sha256()
    h.update(np.uint32(len(boxes)).tobytes())
    for f in BOX_FIELDS:
        h.update(np.ascontiguousarray(boxes[f]).tobytes())
    if reflections is not None:
        h.update(np.uint32(len(reflections)).tobytes())
        for f in REFL_FIELDS:
            h.update(np.ascontiguousarray(reflections[f]).tobytes())
    return h.digest()


def key(workload: str, algorithm: str, rank: int) -> str:
    return f"{workload}/{algorithm}/rank{rank}"


def load_expected(workload: str, algorithm: str, rank: int, n_frames: int):
    """-> dict(num_strong_pixels, n_boxes, n_components, n_reflections: uint32[n_frames]; digest: (n_frames, 32) uint8)
    or None when the fixture holds nothing for this workload / rank / batch."""
    path = os.path.normpath(GOLDEN)
    if not os.path.exists(path):
        return None
    z = np.load(path)
    k = key(workload, algorithm, rank)
    if k + "/num_strong_pixels" not in z.files:
        return None
    out = {f: z[f"{k}/{f}"] for f in ("num_strong_pixels", "n_boxes", "n_components", "n_reflections", "digest")}
    if len(out["n_boxes"]) < n_frames:
        return None
    return {f: v[:n_frames] for f, v in out.items()}


def reflections_digest(reflections: np.ndarray) -> bytes:
    """One 32-byte value for a whole reflection table (a rotation sweep's 3D reflections, in label order)."""
    h = hashlib.sha256()
    h.update(np.uint32(len(reflections)).tobytes())
    for f in REFL_FIELDS:
        h.update(np.ascontiguousarray(reflections[f]).tobytes())
    return h.digest()


def load_expected_sweep(name: str = "sweep16m"):
    """BASELINE.json configs[4], bench.py --workload sweep16m: -> dict(n_reflections, n_calculated, n_filtered_size, n_filtered_sep: int;
    digest: bytes (reflections_digest of the 3D table); num_strong_pixels, n_boxes: uint32[frames]) or None."""
    path = os.path.normpath(GOLDEN)
    if not os.path.exists(path):
        return None
    z = np.load(path)
    if f"{name}/digest" not in z.files:
        return None
    out = {f: int(z[f"{name}/{f}"]) for f in ("n_reflections", "n_calculated", "n_filtered_size", "n_filtered_sep")}
    out["digest"] = z[f"{name}/digest"].tobytes()
    out["num_strong_pixels"] = z[f"{name}/num_strong_pixels"]
    out["n_boxes"] = z[f"{name}/n_boxes"]
    return out
