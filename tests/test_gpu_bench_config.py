"""GPU: the configuration bench.py TIMES, held to the oracle at its full size.

bench.py runs `ffs_submit_device` on 32 frames resident in HBM, four batches in flight, with the library's DEFAULT
parameters: no strong-pixel list, no dense byte mask (`want_strong_list = 0`, `want_strong_mask = 0` -> `need_lists = 0`:
the sparse launch keeps the lists inside LDS and skips their stores, kernels_chain.hpp).  Every other full-size test asks for
the list or the mask, which switches those stores back on -- so this file runs exactly what is timed and compares every frame
of every batch (counts, boxes, reflections bit for bit) with the oracle's result for that frame, and with the committed
fixture bench.py itself checks against (tests/golden/bench_workloads.npz, tests/golden/make_golden_bench.py).
BASELINE.json configs[1]: "results checked per unique frame" (seeds 2000 .. 2031); configs[3]: Jungfrau-9M uint32; SURVEY 8f-1:
the extended algorithm, whose dense frames take the run-based launch without lists from the stream's second batch on."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch  # noqa: F401  (before libffs_hip.so is loaded: torch brings its own copy of the HIP runtime, and two runtimes in one
#                            process cannot both have the GPU -- bench.py imports torch first for the same reason)

from util import assert_frame_matches_oracle, oracle_frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

CASES = [("eiger16m", "dispersion"), ("eiger16m", "dispersion_extended"), ("jungfrau9m", "dispersion")]


def _resident(ctx, frames):
    """The frames in the library's pitched device layout, as bench.py keeps them (torch: device memory only)."""
    import torch
    pitch, fstride = ctx.device_layout()
    B, H, W = frames.shape
    host = np.zeros((B, H, pitch // frames.dtype.itemsize), frames.dtype)
    host[:, :, :W] = frames
    return torch.from_numpy(host.view(np.uint8).reshape(-1)).to("cuda:0"), pitch, fstride


def _pipeline(streams, ptr, pitch, fstride, B, steps, check):
    """bench.py's run_steps: `streams` batches in flight; check(step, results) on every batch."""
    inflight = []
    for step in range(steps + len(streams)):
        if step < steps:
            s = streams[step % len(streams)]
            if len(inflight) == len(streams):
                st0, done = inflight.pop(0)
                check(st0, done.wait())
            s.submit_device(ptr, pitch, fstride, B, first_frame_id=step * B)
            inflight.append((step, s))
        elif inflight:
            st0, done = inflight.pop(0)
            check(st0, done.wait())


@pytest.mark.parametrize("workload,algorithm", CASES)
def test_timed_configuration_against_the_oracle(ffs, workload, algorithm):
    import bench
    from ffs_amd import fixtures
    from oracle import oracle as O
    W, H, dt, _ = bench.WORKLOADS[workload]
    B, n_streams, steps = 32, 4, 12
    frames, mask = bench.make_inputs(workload, B, 0)            # rank 0's frames: seeds 2000 .. 2031 / 4000 .. 4031
    ext = algorithm == "dispersion_extended"

    def oracle(img):
        return oracle_frame(img, mask, strong=O.dispersion_extended(img, mask) if ext else None)
    with ThreadPoolExecutor(16) as ex:                           # (the C oracle runs without the interpreter lock)
        want = list(ex.map(oracle, frames))

    ctx = ffs.Context(W, H, dt, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=1, algorithm=1 if ext else 0)   # exactly bench.py's call: every other field at its default
    assert ctx.params.want_strong_list == 0 and ctx.params.want_strong_mask == 0
    d_frames, pitch, fstride = _resident(ctx, frames)
    streams = [ctx.stream() for _ in range(n_streams)]
    expected = fixtures.load_expected(workload, algorithm, 0, B)
    assert expected is not None, "tests/golden/bench_workloads.npz holds nothing for this workload"
    seen = []

    def check(step, res):
        assert [r.frame_id for r in res] == list(range(step * B, step * B + B))
        for f, fr in enumerate(res):
            assert fr.strong_k is None and fr.strong_mask is None      # nothing asked for, nothing returned
            assert_frame_matches_oracle(fr, frames[f], mask, precomputed=want[f])
            assert fixtures.frame_digest(fr.boxes, fr.reflections) == expected["digest"][f].tobytes(), (step, f)
        assert np.array_equal([r.num_strong_pixels for r in res], expected["num_strong_pixels"])
        assert np.array_equal([len(r.boxes) for r in res], expected["n_boxes"])
        seen.append(step)

    _pipeline(streams, d_frames.data_ptr(), pitch, fstride, B, steps, check)
    assert seen == list(range(steps))
    if ext:
        assert min(r[1].num_strong_pixels for r in want) > 20480    # dense frames: the run-based launch from the 2nd batch of a stream on
    # the strong-pixel lists switched on afterwards give the same frames (and the lists themselves are the oracle's)
    ctx.set_params(want_reflections=1, algorithm=1 if ext else 0, want_strong_list=1)
    streams[0].submit_device(d_frames.data_ptr(), pitch, fstride, B, first_frame_id=0)
    for f, fr in enumerate(streams[0].wait()):
        assert_frame_matches_oracle(fr, frames[f], mask, precomputed=want[f])


def test_partial_batches_and_two_in_flight_without_lists(ffs):
    """The same default parameters with fewer frames than max_batch and with two batches in flight (tuning `chain_first`:
    the sparse launch then also works off the bright-window list): Eiger-16M, 5 of 8 frames."""
    import bench
    W, H, dt, _ = bench.WORKLOADS["eiger16m"]
    frames, mask = bench.make_inputs("eiger16m", 8, 0)
    with ThreadPoolExecutor(8) as ex:
        want = list(ex.map(lambda img: oracle_frame(img, mask), frames))
    ctx = ffs.Context(W, H, dt, max_batch=8)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=1)
    d_frames, pitch, fstride = _resident(ctx, frames)
    streams = [ctx.stream() for _ in range(2)]

    def check(step, res):
        for f, fr in enumerate(res):
            assert_frame_matches_oracle(fr, frames[f + 3], mask, precomputed=want[f + 3])
    _pipeline(streams, d_frames.data_ptr() + 3 * fstride, pitch, fstride, 5, 6, check)
