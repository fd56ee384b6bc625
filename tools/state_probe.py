#!/usr/bin/env python3
"""Does the step time depend on where the buffers lie?  One process, several trials: the bench's frames (torch allocation, behind a
pad of varying size), a fresh context and four streams each time, 60 steps of the resident pipeline; prints the frames' device
address, the mean of the streaming kernel's own events and the step time.   python tools/state_probe.py [trials]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT)
import ffs_amd
import bench

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H, dt, _ = bench.WORKLOADS["eiger16m"]
B = 32
frames, mask = bench.make_inputs("eiger16m", B, 0)
dev = torch.device("cuda", 0)
pads = [0, 1 << 20, 3 << 20, 64 << 10, 5 << 20, 0, 17 << 20, 2 << 20, 512 << 10, 33 << 20, 0, 9 << 20]
keep = []
for t in range(trials):
    pad = pads[t % len(pads)]
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device=dev))
    ctx = ffs_amd.Context(W, H, dt, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=1)
    pitch, fstride = ctx.device_layout()
    host = np.zeros((B, H, pitch // 2), dt)
    host[:, :, :W] = frames
    d_frames = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(dev)
    ptr = d_frames.data_ptr()
    streams = [ctx.stream() for _ in range(4)]

    def run(k):
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
    run(8)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); thr = run(60); torch.cuda.synchronize(dev); el = time.perf_counter() - t0
    print(f"trial {t}: pad {pad >> 10:6d} KiB  frames at {ptr:#x} (mod 2 MiB {ptr % (2 << 20):#x}, mod 1 GiB {(ptr % (1 << 30)) >> 20} MiB)  "
          f"kernel (events) {np.mean(thr[8:]):.4f} ms  step {el / 60 * 1e3:.4f} ms", flush=True)
    for s in streams:
        s.close()
    ctx.close()
    if os.environ.get("FFS_PROBE_HOLD"):   # keep this copy: the next trial's frames land somewhere else
        keep.append(d_frames)
    del d_frames, host
    torch.cuda.empty_cache()
