#!/usr/bin/env python3
"""Profiling harness: only the two threshold kernels on resident synthetic frames
(ffs_bench_threshold), for `rocprofv3 --kernel-trace --stats` and `--pmc` passes.

  rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_threshold.py --iters 20
  rocprofv3 --pmc FETCH_SIZE -d out --output-format csv -- python3 tools/prof_threshold.py --iters 5
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--unique", type=int, default=4, help="unique synthetic frames (cycled)")
    ap.add_argument("--workload", default="eiger16m")
    ap.add_argument("--paths", default="0", help="comma list of tuning threshold_path values to A/B (0 = bright list, 1 = bright plane + exact kernel)")
    ap.add_argument("--exp", default="", help="comma list of FFS_EXP_K1_DEBUG values (experiments build only: FFS_HIP_LIB=.../libffs_hip_exp.so); "
                                              "1 = no group ever flagged, 2 = drains do nothing, 4 = no exact predicate, 8/16 = dense mask off/on, 32 = queue not written, "
                                              "64 = every row reads one row of the mask table, 128 = no tile-count / occupancy atomics, 256 = no plane byte stores; "
                                              "a value may be repeated: one context each (kernel time follows where a context's buffers lie)")
    ap.add_argument("--dense", action="store_true", help="ask for the dense byte mask (want_strong_mask)")
    ap.add_argument("--tune", default="", help="A/B over tuning sets, ';'-separated, each 'key=value,key=value' (ffs_ctx_set_tuning), e.g. "
                                               "'rows_ahead=2;rows_ahead=3;rows_ahead=4'")
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--algorithm", default="dispersion", choices=["dispersion", "dispersion_extended"])
    ap.add_argument("--decode", action="store_true", help="also run the bitshuffle-LZ4 decode kernel on the batch")
    args = ap.parse_args()
    import torch
    import ffs_amd
    from bench import WORKLOADS, make_inputs
    W, H, dt, bpp = WORKLOADS[args.workload]
    frames, mask = make_inputs(args.workload, args.unique, 0)
    B = args.batch
    def make_ctx():
        c = ffs_amd.Context(W, H, dt, max_batch=B)
        c.set_mask(mask)
        c.set_params(algorithm=1 if args.algorithm == "dispersion_extended" else 0, want_strong_mask=1 if args.dense else 0)
        return c

    ctx = make_ctx()
    pitch, fstride = ctx.device_layout()
    host = np.zeros((B, H, pitch // np.dtype(dt).itemsize), dt)
    for i in range(B):
        host[i, :, :W] = frames[i % len(frames)]
    d = torch.from_numpy(host.view(np.uint8).reshape(-1)).cuda()
    print(f"image buffer at {d.data_ptr():#x}", flush=True)
    alg = float(W) * H * bpp * B
    variants = [("path", int(v)) for v in args.paths.split(",")] if not args.exp else [("exp", int(v)) for v in args.exp.split(",")]
    if args.tune:
        variants = [("tune", t) for t in args.tune.split(";")]
    variants = [(k, v, i) for i, (k, v) in enumerate(variants)]   # (a value may be given several times: one context each)
    if args.decode:
        from ffs_amd import bslz4
        st = ctx.stream()
        uniq = [np.frombuffer(bslz4.compress(f), np.uint8) for f in frames]
        chunks = [uniq[i % len(uniq)] for i in range(B)]
        ms, _ = st.decode_only(chunks, iters=args.iters, want_frames=False)
        raw = float(W) * H * np.dtype(dt).itemsize * B
        print(f"decode: {ms*1e3:.1f} us/launch, {raw/ms/1e6:.0f} GB/s of pixels written, "
              f"{sum(c.size for c in chunks)/ms/1e6:.0f} GB/s of chunks read, batch {B}", flush=True)
        del st
    # one context (and stream) per variant: tuning is per context, the experiment switches are read when a context is created
    streams = {}
    for kind, v, idx in variants:
        if kind == "exp":
            os.environ["FFS_EXP_K1_DEBUG"] = str(v)
        c = make_ctx()
        if kind == "path":
            c.set_tuning(threshold_path=v)
        if kind == "tune":
            c.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in v.split(",") if kv})
        streams[(kind, v, idx)] = (c, c.stream())
    for rnd in range(args.rounds):          # interleaved A/B rounds in one process
        for key in variants:
            a, b = streams[key][1].bench_threshold(d.data_ptr(), pitch, fstride, B, args.iters)
            print(f"round {rnd} {key[0]} {key[1]} #{key[2]}: dense kernel {a*1e3:.1f} us/launch ({alg/a/1e6:.0f} GB/s algorithmic, "
                  f"{alg/a/1e6/8000:.3f} of 8 TB/s), rest of the stage {b*1e3:.1f} us/launch, batch {B}", flush=True)


if __name__ == "__main__":
    main()
