#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass) into a per-kernel JSON.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit the TCC slots
together), are in KiB, and on gfx950 FETCH_SIZE reports half of a wide (16 B/lane) streaming
read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

  python tools/summarize_pmc.py gpurun_out/pmcA2 gpurun_out/pmcB2 ... > profiles/rNN_pmc.json
"""
import collections
import csv
import glob
import json
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        if "rocclr" in k:
            continue
        m = {c: sum(x) / len(x) for c, x in v.items()}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
            # quad-cycles of VALU issue per SIMD vs cycles per XCD
            m["valu_busy_frac"] = m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
        out[k] = {c: round(x, 3) for c, x in m.items()}
        out[k]["launches_averaged"] = len(next(iter(v.values())))
    import os
    if os.environ.get("FFS_COMMIT"):
        out["_commit"] = os.environ["FFS_COMMIT"]
    else:
      try:
        import subprocess
        out["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
      except Exception:
        pass
    # what the profiled kernels were compiled from: bench.py compares these hashes with the files it finds and says so when a
    # kernel source changed after the counters were taken (the GPU box has no .git to ask)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = {}
    for f in sorted(glob.glob(os.path.join(root, "fast-feedback-service_amd", "csrc", "kernels_*.hpp")) + [os.path.join(root, "fast-feedback-service_amd", "csrc", "ffs_device.h")]):
        src[os.path.basename(f)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    out["_sources_sha256_16"] = src
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
