#!/usr/bin/env python3
"""Soak: the random sweep of tests/test_gpu_fuzz.py over seeds outside the test-suite's ranges, through the default paths,
the run-based sparse stage and the alternative kernels.   python tools/soak_fuzz.py [first_seed] [n_seeds]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffs_amd
import test_gpu_fuzz as T
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
bad = []
t0 = time.time()
for k, seed in enumerate(range(first, first + n)):
    for name, tuning in (("default", None), ("runs", dict(chain_runs=2)), ("alt", dict(threshold_path=1, ext_first_pass=0, sparse_stage=1))):
        try:
            # every other (seed, path) runs as the library does by default and as bench.py times it: no strong-pixel list, no
            # byte mask asked for (need_lists = 0: the sparse launch keeps the lists in LDS), twice on the same stream
            bare = (seed + len(name)) % 2 == 0
            T.test_random_case(ffs_amd, seed, tuning=tuning, want_list=0 if bare else 1, want_mask=0 if bare else 1, passes=2 if bare else 1)
        except Exception as e:  # noqa: BLE001
            bad.append((seed, name, repr(e)[:200]))
            print("FAIL", seed, name, repr(e)[:300], flush=True)
    if k % 50 == 49:
        print(f"{k + 1} seeds, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print("done:", n, "seeds,", len(bad), "failures")
sys.exit(1 if bad else 0)
