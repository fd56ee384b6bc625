#!/usr/bin/env python3
"""Where a 100-frame sweep's host time goes (stack creation, batches, finish, destroy)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as Bn
import ffs_amd
from ffs_amd import synth

W, H, NZ, B = 4148, 4362, 100, 25
p = synth.sweep_params(seed=5000, n_frames=NZ, n_spots=800)
mask = synth.mask_eiger16m()
ctx = ffs_amd.Context(W, H, np.uint16, max_batch=B, device=0)
ctx.set_mask(mask)
ctx.set_params(want_reflections=0, min_spot_size=3, min_spot_size_3d=15)
pitch, fstride = ctx.device_layout()
d = torch.empty(NZ * fstride, dtype=torch.uint8, device="cuda:0")
host = np.zeros((B, H, pitch // 2), np.uint16)
for z0 in range(0, NZ, B):
    host[:, :, :W] = synth.frames(p, range(z0, z0 + B), threads=16)
    d[z0 * fstride:(z0 + B) * fstride].copy_(torch.from_numpy(host.view(np.uint8).reshape(-1)))
streams = [ctx.stream() for _ in range(4)]
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
reps = 10
for rep in range(reps + 2):
    if rep == 2:
        T.clear(); torch.cuda.synchronize(); t_all = time.perf_counter()
    t = time.perf_counter(); stack = ffs_amd.Stack3D(ctx); tick("create", t)
    t = time.perf_counter()
    for b in range(NZ // B):
        streams[b].submit_device(d.data_ptr() + b * B * fstride, pitch, fstride, B, first_frame_id=b * B)
    tick("submit x4", t)
    for b in range(NZ // B):
        t = time.perf_counter(); streams[b].wait(copy=False); tick("wait", t)
        t = time.perf_counter(); stack.add_batch(streams[b]); tick("add_batch", t)
    t = time.perf_counter(); stack.finish(); tick("finish", t)
    t = time.perf_counter(); stack.close(); tick("close", t)
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print(f"sweep {tot / reps * 1e3:.3f} ms; " + ", ".join(f"{k} {v / reps * 1e3:.3f}" for k, v in T.items()))
