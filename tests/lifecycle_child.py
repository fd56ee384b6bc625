"""Child process of tests/test_gpu_lifecycle.py: builds contexts, streams and a 3D stack through the C ABI, runs a batch on each, and
leaves in the way `mode` names.  The parent asserts exit code 0 and an empty stderr.  (The ABI is driven through raw ctypes handles
where the point is the ORDER of destroy calls, which the Python wrapper's reference counting would otherwise rearrange.)

    python tests/lifecycle_child.py <mode> [torch]
"""
import os
import sys

mode = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] == "torch":
    import torch  # noqa: F401  (first, as tests/conftest.py and bench.py do: it brings its own HIP runtime)
    torch.zeros(8, device="cuda:0").sum().item()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python"))
sys.path.insert(0, ROOT)
import ctypes as C

import numpy as np

import ffs_amd
from ffs_amd import api

W, H, B = 640, 400, 8          # (8 frames x >= 1100 components: ffs_wait's helper threads are in play)
rng = np.random.default_rng(5)
frames = rng.poisson(2.0, (B, H, W)).astype(np.uint16)
ys, xs = rng.integers(4, H - 4, 1400), rng.integers(4, W - 4, 1400)
for f in range(B):
    frames[f, ys, xs] += 500
    frames[f, ys + 1, xs] += 300

ctxs, streams = [], []
for c in range(3):
    ctx = ffs_amd.Context(W, H, np.uint16, max_batch=B)
    ctx.set_params(want_reflections=1, min_spot_size=1)
    ctxs.append(ctx)
    streams += [ctx.stream(), ctx.stream()]
stack = ffs_amd.Stack3D(ctxs[0])
counts = []
for s in streams:
    s.submit(frames, first_frame_id=0)
for s in streams:
    res = s.wait()
    counts.append(sum(len(r.boxes) for r in res))
stack.add_batch(streams[0])
assert len(set(counts)) == 1 and counts[0] >= B * 1100, counts
lib = api.load_library()

if mode == "leak":
    # nothing is closed: handles alive as globals when the interpreter goes down (the shape of tools/soak_blobs.py at sys.exit)
    print("ok", counts[0], flush=True)
    sys.exit(0)

if mode == "leak_hard":
    # ... and the finalisers never run either (os._exit skips them AND the exit handlers: nothing of ours may be needed then)
    print("ok", counts[0], flush=True)
    os._exit(0)

if mode == "inflight":
    # exit with a batch in flight on every stream, one of them compressed (its helper thread indexes blocks), nothing waited for
    from ffs_amd import bslz4
    chunks = [bslz4.compress(f) for f in frames]
    streams[0].submit_compressed(chunks, first_frame_id=0)
    for s in streams[1:]:
        s.submit(frames, first_frame_id=8)
    print("ok", counts[0], flush=True)
    sys.exit(0)

if mode == "ctx_first":
    # the contexts are destroyed FIRST, then their streams and the stack (dead handles: no-ops), then everything a second time
    hs = [C.c_void_p(s._h.value) for s in streams]
    hc = [C.c_void_p(c._h.value) for c in ctxs]
    hk = C.c_void_p(stack._h.value)
    streams[2].submit(frames, first_frame_id=16)       # (one of them with a batch in flight)
    for h in hc:
        lib.ffs_ctx_destroy(h)
    rc = lib.ffs_wait(hs[0], None, None)                # a dead handle is refused, not followed
    assert rc == -1, rc
    assert b"stale ffs_stream handle" in lib.ffs_last_error(None)
    rc = lib.ffs_submit(hs[1], frames.ctypes.data_as(C.c_void_p), B, 0)
    assert rc == -1, rc
    for _ in range(2):
        for h in hs:
            lib.ffs_stream_destroy(h)
        lib.ffs_stack3d_destroy(hk)
        for h in hc:
            lib.ffs_ctx_destroy(h)
    for o in streams + ctxs + [stack]:
        o._h = None
    # the library is still usable afterwards
    ctx = ffs_amd.Context(W, H, np.uint16, max_batch=B)
    ctx.set_params(want_reflections=1, min_spot_size=1)
    st = ctx.stream()
    assert sum(len(r.boxes) for r in st.process(frames)) == counts[0]
    print("ok", counts[0], flush=True)
    sys.exit(0)

if mode == "reverse_gc":
    # finalisers in the order a module teardown might pick: contexts, then the stack, then the streams
    for c in ctxs:
        c.close()
    stack.close()
    for s in streams:
        s.close()
    print("ok", counts[0], flush=True)
    sys.exit(0)

raise SystemExit(f"unknown mode {mode}")
