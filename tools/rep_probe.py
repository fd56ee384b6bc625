import os, sys, time, json
import numpy as np
import torch
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT",".")
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT)
import ffs_amd, bench
W, H, dt, _ = bench.WORKLOADS["eiger16m"]; B=32
frames, mask = bench.make_inputs("eiger16m", B, 0)
dev = torch.device("cuda", 0)
ctx = ffs_amd.Context(W, H, dt, max_batch=B); ctx.set_mask(mask); ctx.set_params(want_reflections=1)
pitch, fstride = ctx.device_layout()
host = np.zeros((B, H, pitch // 2), dt); host[:, :, :W] = frames
d = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(dev); ptr = d.data_ptr()
streams = [ctx.stream() for _ in range(4)]
def run(k):
    thr, infl = [], []
    for step in range(k + 4):
        if step < k:
            s = streams[step % 4]
            if len(infl) == 4:
                x = infl.pop(0); x.wait_counts(); thr.append(x.timings()["threshold"])
            s.submit_device(ptr, pitch, fstride, B, first_frame_id=step * B); infl.append(s)
        elif infl:
            x = infl.pop(0); x.wait_counts(); thr.append(x.timings()["threshold"])
    return thr
run(5)
for rep in range(10):
    torch.cuda.synchronize(dev); t0=time.perf_counter(); thr=run(20); torch.cuda.synchronize(dev); el=time.perf_counter()-t0
    print(f"rep {rep}: {el/20*1e3:.4f} ms/step, kernel events mean {np.mean(thr):.4f} first5 {np.round(thr[:5],4).tolist()} last5 {np.round(thr[-5:],4).tolist()}", flush=True)
    if rep == 5: time.sleep(0.5); print("(slept 0.5 s)")
