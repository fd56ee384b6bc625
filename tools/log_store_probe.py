#!/usr/bin/env python3
"""What do the wave-log stores cost the streaming kernel, and are THEY what moves with a context's place in memory?  Experiments
build only: contexts are created alternately with FFS_EXP_K1_DEBUG = 0 and = <bit> (512: the log stays unwritten and empty; results
are wrong then), all kept alive, one copy of the frames; the streaming kernel's own events over 60 pipelined batches each.
    FFS_HIP_LIB=.../libffs_hip_exp.so python tools/log_store_probe.py [bit] [contexts per kind]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT)
import ffs_amd
import bench

bit = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_each = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, dt, _ = bench.WORKLOADS["eiger16m"]
B = 32
frames, mask = bench.make_inputs("eiger16m", B, 0)
dev = torch.device("cuda", 0)

def make_ctx(dbg):
    os.environ["FFS_EXP_K1_DEBUG"] = str(dbg)
    ctx = ffs_amd.Context(W, H, dt, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=1)
    return ctx, [ctx.stream() for _ in range(4)]

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

keep, t = [], None
res = {0: [], bit: []}
for i in range(2 * n_each):
    dbg = bit if i % 2 else 0
    ctx, streams = make_ctx(dbg)
    keep.append((dbg, ctx, streams))
    if t is None:
        pitch, fstride = ctx.device_layout()
        host = np.zeros((B, H, pitch // 2), dt); host[:, :, :W] = frames
        t = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(dev)
    k, st = measure(streams, t.data_ptr(), pitch, fstride)
    res[dbg].append(k)
    print(f"context {i} dbg {dbg}: kernel (events) {k:.4f} ms  step {st:.4f} ms", flush=True)
for i, (dbg, ctx, streams) in enumerate(keep):
    k, st = measure(streams, t.data_ptr(), pitch, fstride)
    res[dbg].append(k)
    print(f"again context {i} dbg {dbg}: kernel (events) {k:.4f} ms  step {st:.4f} ms", flush=True)
for dbg, v in res.items():
    print(f"dbg {dbg}: n {len(v)}  min {min(v):.4f}  mean {np.mean(v):.4f}  max {max(v):.4f} ms")
