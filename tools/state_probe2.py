#!/usr/bin/env python3
"""Which buffer's placement moves the streaming kernel's time?  mode frames: ONE context and its streams, eight copies of the frames
held at once (different physical places), the pipeline over each copy in turn, twice.  mode ctx: one copy of the frames, a fresh
context (mask tables, wave logs, band scratch somewhere else) per trial, the old ones kept alive.   python tools/state_probe2.py frames|ctx"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT)
import ffs_amd
import bench

mode = sys.argv[1] if len(sys.argv) > 1 else "frames"
W, H, dt, _ = bench.WORKLOADS["eiger16m"]
B = 32
frames, mask = bench.make_inputs("eiger16m", B, 0)
dev = torch.device("cuda", 0)

def make_ctx():
    ctx = ffs_amd.Context(W, H, dt, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=1)
    return ctx, [ctx.stream() for _ in range(4)]

def make_frames(ctx):
    pitch, fstride = ctx.device_layout()
    host = np.zeros((B, H, pitch // 2), dt)
    host[:, :, :W] = frames
    return torch.from_numpy(host.view(np.uint8).reshape(-1)).to(dev), pitch, fstride

def run(streams, ptr, pitch, fstride, k):
    thr, infl = [], []
    for step in range(k + 4):
        if step < k:
            s = streams[step % 4]
            if len(infl) == 4:
                d = infl.pop(0); d.wait_counts(); thr.append(d.timings()["threshold"])
            s.submit_device(ptr, pitch, fstride, B, first_frame_id=step * B)
            infl.append(s)
        elif infl:
            d = infl.pop(0); d.wait_counts(); thr.append(d.timings()["threshold"])
    return thr

def measure(streams, ptr, pitch, fstride):
    run(streams, ptr, pitch, fstride, 8)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); thr = run(streams, ptr, pitch, fstride, 60); torch.cuda.synchronize(dev)
    return float(np.mean(thr[8:])), (time.perf_counter() - t0) / 60 * 1e3

if mode == "frames":
    ctx, streams = make_ctx()
    copies = [make_frames(ctx) for _ in range(8)]
    for rnd in range(2):
        for i, (t, pitch, fstride) in enumerate(copies):
            k, st = measure(streams, t.data_ptr(), pitch, fstride)
            print(f"round {rnd} frames copy {i} at {t.data_ptr():#x}: kernel (events) {k:.4f} ms  step {st:.4f} ms", flush=True)
else:
    keep = []
    t = None
    for i in range(8):
        ctx, streams = make_ctx()
        keep.append((ctx, streams))
        if t is None:
            t, pitch, fstride = make_frames(ctx)
        k, st = measure(streams, t.data_ptr(), pitch, fstride)
        print(f"context {i}: kernel (events) {k:.4f} ms  step {st:.4f} ms", flush=True)
    for rnd in range(1):
        for i, (ctx, streams) in enumerate(keep):
            k, st = measure(streams, t.data_ptr(), pitch, fstride)
            print(f"again context {i}: kernel (events) {k:.4f} ms  step {st:.4f} ms", flush=True)
