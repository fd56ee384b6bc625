#!/usr/bin/env python3
"""Where a bench step goes on the host: time inside submit_device / wait per batch, for N batches in flight."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as Bn
import ffs_amd

ap = argparse.ArgumentParser()
ap.add_argument("--streams", default="4")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--workload", default="eiger16m")
a = ap.parse_args()
W, H, dt, bpp = Bn.WORKLOADS[a.workload]
B = a.batch
frames, mask = Bn.make_inputs(a.workload, B, 0)
ctx = ffs_amd.Context(W, H, dt, max_batch=B, device=0)
ctx.set_mask(mask)
ctx.set_params(want_reflections=1)
pitch, fstride = ctx.device_layout()
host = np.zeros((B, H, pitch // np.dtype(dt).itemsize), dt)
host[:, :, :W] = frames
d = torch.from_numpy(host.view(np.uint8).reshape(-1)).to("cuda:0")
ptr = d.data_ptr()
for ns in [int(x) for x in a.streams.split(",")]:
    streams = [ctx.stream() for _ in range(ns)]
    for rep in range(2):
        t_sub = t_wait = t_py = 0.0
        inflight = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(a.steps + ns):
            if step < a.steps:
                s = streams[step % ns]
                if len(inflight) == ns:
                    dn = inflight.pop(0)
                    t = time.perf_counter(); res = dn.wait(copy=False); t_wait += time.perf_counter() - t
                    t = time.perf_counter(); _ = sum(len(r.boxes) for r in res) + sum(r.num_strong_pixels for r in res); t_py += time.perf_counter() - t
                t = time.perf_counter(); s.submit_device(ptr, pitch, fstride, B, first_frame_id=step * B); t_sub += time.perf_counter() - t
                inflight.append(s)
            elif inflight:
                dn = inflight.pop(0)
                t = time.perf_counter(); res = dn.wait(copy=False); t_wait += time.perf_counter() - t
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    print(f"streams {ns}: {el / a.steps * 1e3:.4f} ms/step  {a.steps * B / el:.0f} fps | submit {t_sub / a.steps * 1e3:.4f}  wait {t_wait / a.steps * 1e3:.4f}  python sums {t_py / a.steps * 1e3:.4f} ms/step", flush=True)
    tm = streams[0].timings()
    print("   last batch stage ms:", {k: round(v, 3) for k, v in tm.items()}, flush=True)
    for s in streams:
        s.close()
