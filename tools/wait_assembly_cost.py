#!/usr/bin/env python3
"""Host time of ffs_wait when the batch has long been finished on the GPU: what the result assembly (wire records ->
boxes, reflections, centre rows) costs per batch of 32 Eiger-16M frames.   python tools/wait_assembly_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import ffs_amd, bench
W, H, dt, _ = bench.WORKLOADS["eiger16m"]
B = 32
frames, mask = bench.make_inputs("eiger16m", B, 0)
for refl in (1, 0):
    ctx = ffs_amd.Context(W, H, dt, max_batch=B); ctx.set_mask(mask); ctx.set_params(want_reflections=refl)
    pitch, fstride = ctx.device_layout()
    host = np.zeros((B, H, pitch // 2), dt); host[:, :, :W] = frames
    d = torch.from_numpy(host.view(np.uint8).reshape(-1)).to("cuda:0")
    st = ctx.stream()
    ts = []
    for rep in range(6):
        st.submit_device(d.data_ptr(), pitch, fstride, B)
        time.sleep(0.01)
        t0 = time.perf_counter(); n, nb, ns = st.wait_counts(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"want_reflections={refl}: ffs_wait after the GPU is done: {['%.3f' % t for t in ts]} ms ({nb} boxes)")
