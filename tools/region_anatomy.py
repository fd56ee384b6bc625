#!/usr/bin/env python3
"""Anatomy of the bench's timed regions from a rocprofv3 kernel trace: regions = runs of streaming kernels separated by idle time.
    rocprofv3 --kernel-trace -d gpurun_out/prof --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streamed --no-cli-e2e
    python3 tools/region_anatomy.py gpurun_out/prof [steps]"""
import csv, glob, sys
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(f"{d}/*/*kernel_trace.csv")[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ffsamd::", "").replace("ffsamd::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:24], r["Queue_Id"]))
ev.sort()
ks = [e for e in ev if e[2].startswith("k_stream_u16<3") or e[2].startswith("k_stream_u32")]
# split into regions: a gap of more than 150 us between consecutive streaming kernels
regions, cur = [], []
for e in ks:
    if cur and e[0] - cur[-1][1] > 150_000:
        regions.append(cur); cur = []
    cur.append(e)
if cur: regions.append(cur)
shown = 0
for reg in regions:
    if len(reg) != steps:
        continue
    t0, t1 = reg[0][0], reg[-1][1]
    sparse = [e for e in ev if ("k_band" in e[2] or "k_frame" in e[2]) and t0 <= e[0] <= t1 + 1_000_000]
    tail = (max(e[1] for e in sparse) - t1) / 1e3 if sparse else float("nan")
    dur = [(e[1] - e[0]) / 1e3 for e in reg]
    gaps = [(reg[i + 1][0] - reg[i][1]) / 1e3 for i in range(len(reg) - 1)]
    print(f"region of {steps} steps: first streaming kernel's start -> last one's end {(t1 - t0) / 1e3:8.1f} us; last sparse kernel ends {tail:6.1f} us later; "
          f"streaming kernels: mean {sum(dur) / len(dur):6.1f} us (first four {[round(x) for x in dur[:4]]}, last {round(dur[-1])}); "
          f"gap between consecutive streaming kernels: mean {sum(gaps) / len(gaps):5.1f} us, max {max(gaps):5.1f}")
    shown += 1
    last = reg
print(f"({shown} regions of {steps} steps)")
if shown:
    t0 = last[-4][0]
    print("\n# the end of the last such region (us from the fourth-last streaming kernel's start): start, duration, end, kernel, queue")
    for s, e, n, q in ev:
        if t0 <= s <= last[-1][1] + 400_000:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(e - t0) / 1e3:9.1f}  {n:24s} q{q}")
