#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel trace as a timeline (start, duration, kernel, queue)."""
import csv, glob, sys
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
f = glob.glob(f"{d}/*/*kernel_trace.csv")[0]
ev = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("void ffsamd::", "").replace("ffsamd::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:28], r["Queue_Id"], r.get("Stream_Id", "")))
ev.sort()
k1 = [e for e in ev if "k_stream" in e[2]]
print("kernels", len(ev), "K1", len(k1))
if len(k1) > 40:
    a, b = k1[-31][0], k1[-1][0]
    print("steady step us:", (b - a) / 30 / 1e3)
w = ev[skip:skip + count]
t0 = w[0][0]
for s, e, n, q, st in w:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n:28s} q{q} s{st}")
