#!/usr/bin/env python3
"""What ffs_wait() costs the host once the GPU is done (result assembly), and what the tail of a K-step run is made of."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python"))
import torch
import bench as Bn
import ffs_amd
W, H, dt, bpp = Bn.WORKLOADS["eiger16m"]
B = 32
frames, mask = Bn.make_inputs("eiger16m", 4, 0)
ctx = ffs_amd.Context(W, H, dt, max_batch=B, device=0)
ctx.set_mask(mask)
pitch, fstride = ctx.device_layout()
host = np.zeros((B, H, pitch // np.dtype(dt).itemsize), dt)
for i in range(B):
    host[i, :, :W] = frames[i % 4]
d = torch.from_numpy(host.view(np.uint8).reshape(-1)).to("cuda:0")
ptr = d.data_ptr()
streams = [ctx.stream() for _ in range(4)]
for s in streams:
    s.submit_device(ptr, pitch, fstride, B, 0); s.wait(copy=False)
for want in (0, 1):
    ctx.set_params(want_reflections=want)
    ts = []
    for rep in range(10):
        streams[0].submit_device(ptr, pitch, fstride, B, 0)
        time.sleep(0.005)                       # the GPU is long done
        t = time.perf_counter(); streams[0].wait(copy=False); ts.append(time.perf_counter() - t)
    print(f"want_reflections={want}: wait() after the GPU is done: {np.median(ts)*1e6:.0f} us (min {min(ts)*1e6:.0f})")
ctx.set_params(want_reflections=0)
# tail of a K-step run: timestamps of the last waits relative to the last submit
K = 20
for rep in range(3):
    infl = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    subs, dones = [], []
    for step in range(K):
        s = streams[step % 4]
        if len(infl) == 4:
            infl.pop(0).wait_counts(); dones.append(time.perf_counter() - t0)
        s.submit_device(ptr, pitch, fstride, B, step); infl.append(s); subs.append(time.perf_counter() - t0)
    for h in infl:
        h.wait_counts(); dones.append(time.perf_counter() - t0)
    print(f"K={K}: per step {1e3*dones[-1]/K:.4f} ms; batches done at (ms): " + " ".join(f"{1e3*d:.2f}" for d in dones))
    print("      submits at (ms): " + " ".join(f"{1e3*d:.2f}" for d in subs))
