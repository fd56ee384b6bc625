#!/usr/bin/env python3
"""bench.py -- detector frames/s of the spot-finder hot path on MI355X.

One "step" = one pass of the whole hot path (dispersion threshold -> strong-pixel
compaction -> 2D connected components -> centroids/filters -> results on the host) over
one batch of synthetic frames that are already resident in HBM.  N=1 workload =
BASELINE.json configs[1]: Eiger-2XE 16M (4148 x 4362 uint16), 7x7 window ("3x3 kernel"
half-widths), synthetic frames with Poisson background + Gaussian spots and the Eiger
module-gap mask.

`python3 bench.py --gpus N` works when invoked plainly: with N > 1 and no WORLD_SIZE in the
environment it starts its N ranks itself (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relays rank 0's one JSON line and
exits with the worst child's code -- without ever touching the GPU in the launching process.
Under torch.distributed.run it is a rank as before.  Every rank processes its own shard of
the frame queue (weak scaling; frames are independent, spotfinder/spotfinder.cc:686,752) and
the per-frame spot lists are gathered to rank 0 once per `--gather-every` batches: an
all_gather of the ranks' row counts, then exactly the written rows by RCCL send / recv
(`--gather padded`: round 3's all_gather of fixed-size blocks to every rank, for A/B).
Every batch waited for is checked against committed oracle results ("results_checked").  `--single-process` drives the N GPUs from ONE process instead --
one context and one host thread per GPU behind ffs_multi_init, the C++ driver's model
(`spotfinder --gpus N`) -- so both designs get a curve.

Prints ONE JSON line (rank 0).  torch is used only for device memory and
torch.distributed; the product is libffs_hip.so behind include/ffs_hip.h.
"""
import argparse
import json
import os
import re
import shutil
import socket
import statistics
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "fast-feedback-service_amd")
sys.path.insert(0, os.path.join(PKG, "python"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (width, height, dtype, algorithmic bytes per pixel = pixel + mask read + mask write)
    "eiger16m": (4148, 4362, np.uint16, 4),
    "jungfrau9m": (3072, 3072, np.uint32, 6),
    "plumbing1k": (1024, 1024, np.uint16, 4),
}


# ---------------------------------------------------------------------------------------------------
# host description (CPU model beside nproc; NUMA layout for the streamed legs)
def host_info():
    model, nodes = None, 0
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        nodes = len([d for d in os.listdir("/sys/devices/system/node") if re.fullmatch(r"node\d+", d)])
    except OSError:
        pass
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "affinity_cores": share, "numa_nodes": nodes}


def visible_gpus():
    """GPUs this job may use, counted WITHOUT initialising HIP (the launching process must never touch the
    GPU: its children are exec'ed).  KFD topology nodes with SIMDs, cut by the *_VISIBLE_DEVICES lists."""
    if os.environ.get("FFS_BENCH_ASSUME_GPUS"):          # CPU rehearsal of the launcher (tests)
        return int(os.environ["FFS_BENCH_ASSUME_GPUS"])
    n = 0
    top = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(top):
            try:
                with open(os.path.join(top, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        n = -1
    if n < 0:
        try:
            import torch
            n = torch.cuda.device_count()    # (does not create a HIP context on this image)
        except Exception:
            n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """--gpus N > 1 invoked plainly: N child ranks of this same script, rank 0's JSON line relayed."""
    n = args.gpus
    have = visible_gpus()
    if have < n:
        print(f"bench.py: --gpus {n} asked for, {have} GPU(s) visible on this host: not started "
              f"(run on a node with {n} GPUs, or lower --gpus)", file=sys.stderr)
        return 3
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FFS_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    line = None
    deadline = time.time() + args.launch_timeout
    out0 = []

    def pump():
        for ln in procs[0].stdout:
            out0.append(ln)
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    worst = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                worst = worst or rc
                # one rank down: the others would sit in a collective for ever
                for q in alive:
                    procs[q].terminate()
        if time.time() > deadline:
            print(f"bench.py: ranks still running after {args.launch_timeout:.0f} s: stopping them", file=sys.stderr)
            for q in alive:
                procs[q].kill()
            worst = worst or 4
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(10)
        except subprocess.TimeoutExpired:
            p.kill()
    th.join(5)
    for ln in out0:
        if ln.lstrip().startswith("{"):
            line = ln.strip()
    if line and worst == 0:
        print(line, flush=True)
        return 0
    print(f"bench.py: the {n}-rank run failed (worst exit code {worst})", file=sys.stderr)
    return worst or 5


# ---------------------------------------------------------------------------------------------------
def make_inputs(workload, n_unique, rank):
    from ffs_amd import synth
    if workload == "eiger16m":
        p = synth.eiger16m_params(seed=2000 + 1000 * rank)
        mask = synth.mask_eiger16m()
    elif workload == "jungfrau9m":
        p = synth.jungfrau9m_params(seed=4000 + 1000 * rank)
        mask = synth.mask_modules(3072, 3072, 1024, 512, 0, 0)
    else:
        p = synth.config1_params(seed=1000 + 1000 * rank)
        mask = synth.config1_mask()
    frames = synth.frames(p, range(n_unique), threads=min(16, os.cpu_count() or 1))
    return frames, mask


def stale_note(summary, kernel_name):
    """Says when the sources that define the profiled kernel are not the ones the counters were taken with: by the hashes the
    summary carries (tools/summarize_pmc.py), else by `git diff` against its commit stamp where there is a repository."""
    import hashlib
    defining = ["kernels_stream.hpp", "kernels_threshold.hpp", "ffs_device.h"] + (["kernels_extended.hpp"] if "<2, true" in kernel_name else [])
    csrc = os.path.join(PKG, "csrc")
    then = summary.get("_sources_sha256_16")
    if then:
        changed = [f for f in defining if f in then and os.path.exists(os.path.join(csrc, f))
                   and hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16] != then[f]]
        return f" -- STALE: {', '.join(changed)} changed since these counters were taken" if changed else " -- kernel sources unchanged since"
    if summary.get("_commit") and os.path.isdir(os.path.join(ROOT, ".git")):
        try:
            r = subprocess.run(["git", "-C", ROOT, "diff", "--name-only", summary["_commit"], "HEAD", "--"] + [os.path.join(csrc, f) for f in defining],
                               capture_output=True, text=True, timeout=20)
            changed = [os.path.basename(x) for x in r.stdout.split()]
            if r.returncode == 0:
                return f" -- STALE: {', '.join(changed)} changed since that commit" if changed else " -- kernel sources unchanged since"
        except Exception:
            pass
    return " -- whether the kernel sources changed since is unknown here (no source hashes in the summary, no repository)"


def pmc_traffic(workload, batch):
    """HBM bytes per launch of the threshold kernel from the newest committed rocprofv3 PMC summary
    for this workload and batch (profiles/*pmc_threshold_<workload>_b<batch>.json, produced by
    tools/summarize_pmc.py from separate FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2
    correction on FETCH_SIZE).  None if no such profile exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_threshold_{workload}_b{batch}.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        for k, v in d.items():
            if ("k_stream" in k or "k_candidates" in k) and "hbm_bytes_per_launch" in v:
                src = os.path.relpath(files[-1], ROOT)
                if d.get("_commit"):
                    src += f" (rocprofv3 PMC passes taken at commit {d['_commit']}; not re-measured in this run)"
                src += stale_note(d, k)
                return int(v["hbm_bytes_per_launch"]), src
    except Exception:
        pass
    return None, None


def cpu_baseline(frames, mask, ext=False, budget_s=12.0):
    """Reference CPU path timed on this box's host cores: dispersion threshold by the
    reference's own standalone.cc when oracle/_ref is present (else our restatement), then the
    oracle's connected components; one frame per thread, the reference's threading model
    (spotfinder/spotfinder.cc:725-752).  Two legs: one core, and every
    core this process may use."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    H, W = mask.shape
    hi = host_info()
    nproc = hi["affinity_cores"]
    kind = "reference" if O.have_ref() and not ext else "port"   # baseline.cpp's extended class needs DIALS

    def worker(idx_list):
        sf = O.RefSpotfinder(W, H) if kind == "reference" else O.PortSpotfinder(W, H)
        dst = np.empty((H, W), np.uint8)
        done = 0
        t_end = time.perf_counter() + budget_s
        for i in idx_list:
            img = frames[i]
            if ext:
                dst = O.dispersion_extended(img, mask)
            else:
                f64 = img.astype(np.float64)          # the reference converts too (spotfinder.cc:1024)
                sf.run_f64(f64, mask, dst)
            O.cc2d(dst, img, 3)
            done += 1
            if time.perf_counter() > t_end:
                break
        return done

    def leg(cores, per):
        jobs = [[(c * per + j) % len(frames) for j in range(per)] for c in range(cores)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            done = sum(ex.map(worker, jobs))
        dt = time.perf_counter() - t0
        return done, dt

    done1, dt1 = leg(1, 8)                       # ~8 x 0.13 s on one core
    # "all cores" = this job's CPU share: a 1-GPU box of the pool shows every core of the host (nproc, reported)
    # but grants 16 per GPU; FFS_BENCH_CPU_THREADS overrides (256 threads measured 32 frames/s against 40 with
    # 16: the summed-area tables of 256 frames do not fit the caches)
    cores = max(1, min(nproc, int(os.environ.get("FFS_BENCH_CPU_THREADS", "16"))))
    done, dt = leg(cores, 4)
    what = ("reference baseline/spotfinder/standalone.cc (oracle/_ref)" if kind == "reference"
            else "oracle port" + (" of baseline.cpp DispersionExtendedThreshold" if ext else ""))
    return {
        "value": round(done / dt, 3), "unit": "frames/s", "cores": cores, "kind": kind, "nproc": nproc,
        "cpu_model": hi["cpu_model"], "host_cores": hi["nproc"], "numa_nodes": hi["numa_nodes"],
        "single_core": {"value": round(done1 / dt1, 3), "unit": "frames/s", "cores": 1,
                        "ms_per_frame": round(dt1 / done1 * 1e3, 1), "sample": f"{done1} frames, {dt1:.1f} s wall"},
        "ms_per_frame_per_core": round(dt * cores / done * 1e3, 1),
        "sample": f"{done} {W}x{H} frames of the same workload, one frame per thread on {cores} threads "
                  f"(the CPU share of one GPU on this pool; the host shows {hi['nproc']} cores); threshold = {what}"
                  f", connected components = oracle port (Boost.Graph absent); {dt:.1f} s wall",
    }


# ---------------------------------------------------------------------------------------------------
def cli_e2e(n_images=1000, threads=None, batch=None, cpu_decode_images=256, keep_dir=None, long_images=4096):
    """End-to-end rate of the drop-in binary, as the Zocalo service launches it (`spotfinder <stream dir> --threads N
    --pipe_fd FD`, src/ffs/service.py:419-440): bin/spotfinder on an Eiger-stream directory of `n_images` bitshuffle-LZ4
    frames (BASELINE.json configs[1]'s count; 32 distinct frames, the rest are directory entries pointing at them),
    JSON lines read from the pipe by this harness, frames/s from the binary's own last line
    (spotfinder/spotfinder.cc:1308-1322).  Two legs: chunks decoded on the GPU (default) and on the worker threads
    (--cpu-decode, the reference's way).  Runs BEFORE this process touches the GPU."""
    exe = os.path.join(PKG, "bin", "spotfinder")
    tool = os.path.join(PKG, "bin", "ffs_hosttool")
    if not (os.path.exists(exe) and os.path.exists(tool)):
        return {"error": "bin/spotfinder or bin/ffs_hosttool not built (make cli)"}
    hi = host_info()
    if threads is None:
        threads = max(2, min(hi["affinity_cores"], int(os.environ.get("FFS_BENCH_CPU_THREADS", "16"))))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    work = keep_dir or tempfile.mkdtemp(prefix="ffs_e2e_", dir=base)
    shm = os.path.join(work, "stream")
    out = {"images": n_images, "threads": threads, "batch": batch or "default (4)", "unique_frames": 32,
           "source": "Eiger-stream directory (start_1/4/5 + image_%06d_2 bitshuffle-LZ4 chunks) in " + (base or "tmp"),
           "cpu_model": hi["cpu_model"], "host_cores": hi["nproc"], "affinity_cores": hi["affinity_cores"],
           "numa_nodes": hi["numa_nodes"]}
    try:
        t0 = time.perf_counter()
        r = subprocess.run([tool, "mkshm", "synth:eiger16m:32", shm], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": "ffs_hosttool mkshm failed: " + (r.stdout + r.stderr)[-300:]}
        # every frame its own file, as a detector writes them (symbolic links to 32 files -- round 2 and 3a -- kept the whole
        # data set in the page cache's hot end and half of it in L3).  tmpfs pages are charged to this job's memory: the long run
        # is shrunk (or dropped) when the files would not fit what the file system and the memory cgroup have left, and a copier
        # thread's error ends the leg at once instead of leaving spotfinder waiting for images that never come.
        chunk = os.path.getsize(os.path.join(shm, "image_000000_2"))
        room = shutil.disk_usage(work).free
        for lim in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                v = open(lim).read().strip()
                if v != "max":
                    used = 0
                    for cur in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                        if os.path.exists(cur):
                            used = int(open(cur).read().strip())
                            break
                    room = min(room, int(v) - used)
                break
            except (OSError, ValueError):
                continue
        fit = int(room * 0.6 // max(chunk, 1))          # (leave room for the GPU legs that follow: pinned frames, torch)
        if long_images > n_images and long_images > fit:
            out["long_run_shrunk"] = f"{long_images} -> {max(fit, 0)} images: {room / 1e9:.1f} GB left in {base or 'tmp'} / the memory cgroup"
            long_images = fit if fit > n_images else 0
        if n_images > fit:
            return dict(out, error=f"{n_images} chunk files of {chunk / 1e6:.1f} MB do not fit the {room / 1e9:.1f} GB left in {base or 'tmp'} / the memory cgroup")
        n_files = max(n_images, long_images)
        copy_errors = []

        def copy_range(lo, hi):
            try:
                for i in range(lo, hi):
                    if copy_errors:
                        return
                    shutil.copyfile(os.path.join(shm, f"image_{i % 32:06d}_2"), os.path.join(shm, f"image_{i:06d}_2"))
            except OSError as e:
                copy_errors.append(f"{type(e).__name__}: {e}")
        step = (n_files - 32 + 7) // 8
        copiers = [threading.Thread(target=copy_range, args=(32 + k * step, min(32 + (k + 1) * step, n_files))) for k in range(8)]
        for t in copiers:
            t.start()
        for t in copiers:
            t.join()
        if copy_errors:
            return dict(out, error="writing the chunk files failed: " + copy_errors[0])
        hdr = open(os.path.join(shm, "start_1")).read()
        open(os.path.join(shm, "start_1"), "w").write(hdr.replace('"nimages": 32', f'"nimages": {max(n_images, long_images)}'))
        out["prepare_s"] = round(time.perf_counter() - t0, 1)
        out["chunk_MB"] = round(chunk / 1e6, 2)

        def run(extra, images):
            rfd, wfd = os.pipe()
            lines = []

            def reader():
                with os.fdopen(rfd, "r") as f:
                    for ln in f:
                        lines.append(ln)
            th = threading.Thread(target=reader, daemon=True)
            th.start()
            t1 = time.perf_counter()
            p = subprocess.run([exe, shm, "--threads", str(threads), "--images", str(images), "--pipe_fd", str(wfd)]
                               + (["--batch", str(batch)] if batch else []) + extra, pass_fds=(wfd,), capture_output=True,
                               text=True, timeout=600, cwd=work)
            wall = time.perf_counter() - t1
            os.close(wfd)
            th.join(10)
            m = re.search(r"(\d+) images in ([0-9.]+) s .*?\(\x1b\[1;34m([0-9.]+) fps", p.stdout)
            if p.returncode != 0 or not m:
                return {"error": f"rc {p.returncode}: " + (p.stdout + p.stderr)[-300:]}
            ok = 0
            for ln in lines:
                try:
                    d = json.loads(ln)
                    ok += int("n_spots_total" in d and "num_strong_pixels" in d and "file-number" in d)
                except ValueError:
                    pass
            return {"frames_per_s": float(m.group(3)), "images": int(m.group(1)), "binary_s": float(m.group(2)),
                    "chunks_read_GBps": round(int(m.group(1)) * chunk / 1e9 / max(float(m.group(2)), 1e-9), 1),
                    "wall_s": round(wall, 3), "wall_frames_per_s": round(int(m.group(1)) / max(wall, 1e-9), 1),
                    "start_up_and_tear_down_s": round(wall - float(m.group(2)), 3), "json_lines": ok, "stderr_bytes": len(p.stderr)}

        run([], 32)          # untimed warm-up of the binary (first GPU context of the process tree, page cache of the libraries)
        # first pass over files nobody has read yet (what a live data set is): bounded by the kernel -- the first read() after the
        # write() moves every page to the active list, one lock for all readers: 12-24 GB/s per host whatever reads
        # (tools/ubench/cold_read.cc) -- the later passes are what the driver itself can do
        out["first_pass_over_fresh_files"] = run([], n_images)
        # (a 1000-image run lasts 0.17 s: one run moves +-8 % with whatever else the host's memory system is doing -- the median of three)
        runs = [run([], n_images) for _ in range(3)]
        good = sorted((r for r in runs if "frames_per_s" in r), key=lambda r: r["frames_per_s"])
        gpu = dict(good[len(good) // 2], runs_frames_per_s=[r.get("frames_per_s") for r in runs], value_from="median of three runs") if good else runs[0]
        out["gpu_decode"] = gpu
        out["frames_per_s"] = gpu.get("frames_per_s")
        out["cpu_decode"] = run(["--cpu-decode"], min(n_images, cpu_decode_images))
        if long_images > n_images:   # the fixed set-up and drain (tens of ms) against a run of the length of a real data set
            out["long_run_first_pass"] = run([], long_images)
            out["long_run"] = run([], long_images)
        # two contexts on this one GPU, each with its own readers (2 x 8), batches dealt alternately: what the HOST side of a
        # multi-GPU node has to sustain per pair of GPUs -- the frame source, the page cache and the memory system are shared by
        # all GPUs of a host, PCIe is not (here both contexts share one link, so the frame rate stays at one GPU's)
        out["two_contexts_one_gpu"] = run(["--devices", "0,0"], max(n_images, long_images))
        out["note"] = ("frames/s = the binary's own last line (timer from just before its workers start to after the last result, "
                       "stream set-up and pinned staging included), as the reference prints it; `wall_frames_per_s` = images / the wall time of "
                       "the whole process as its caller sees it (the service starts one process per request: HIP runtime, code object, context, "
                       "mask upload before the timer, exit after it: `start_up_and_tear_down_s`; tools/cli_startup.sh breaks it down); at most 8 of the threads feed the GPU when it decodes the chunks; "
                       "PCIe floor for 7.5 MB chunks at the 55 GB/s measured on this pool: ~7.3 k frames/s; every frame is a file of its own "
                       "(7.5 GB per 1000), `first_pass_over_fresh_files` = the same run the first time those files are read")
    except Exception as e:  # the bench line must still come out
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        if not keep_dir:
            shutil.rmtree(work, ignore_errors=True)
    return out


# ---------------------------------------------------------------------------------------------------
def bench_sweep(args, dev, local_rank):
    """BASELINE.json configs[4]: 100-frame Eiger-16M fine-phi sweep (800 reflections with a rocking curve,
    min_spot_size 3, min_spot_size_3d 15 -- tests/3d_connected_components.sh:27-37), frames resident in HBM.
    One step = the whole sweep: four 25-frame batches through threshold + 2D components, their strong-pixel
    lists appended to the device-resident 3D stack, then ffs_stack3d_finish (3D union-find, centroids,
    filters, per-signal labels) with the reflections on the host."""
    import torch
    import ffs_amd
    from ffs_amd import synth
    W, H, NZ, B = 4148, 4362, 100, 25
    p = synth.sweep_params(seed=5000, n_frames=NZ, n_spots=800)
    mask = synth.mask_eiger16m()
    ctx = ffs_amd.Context(W, H, np.uint16, max_batch=B, device=local_rank)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=0, min_spot_size=3, min_spot_size_3d=15)
    pitch, fstride = ctx.device_layout()
    d_frames = torch.empty(NZ * fstride, dtype=torch.uint8, device=dev)
    host = np.zeros((B, H, pitch // 2), np.uint16)
    for z0 in range(0, NZ, B):
        host[:, :, :W] = synth.frames(p, range(z0, z0 + B), threads=min(16, os.cpu_count() or 1))
        d_frames[z0 * fstride:(z0 + B) * fstride].copy_(torch.from_numpy(host.view(np.uint8).reshape(-1)))
    del host
    streams = [ctx.stream() for _ in range(4)]   # the whole sweep in flight: a batch's sparse stage runs beside the next batches' threshold kernels
    n_refl, finish_ms = 0, []
    # self-check: every sweep's 3D reflection table (counts and digest) and every frame's strong pixels and boxes against the
    # committed oracle results for this sweep (tests/golden/bench_workloads.npz, make_golden_bench.py --only sweep16m)
    from ffs_amd import fixtures
    expected = fixtures.load_expected_sweep("sweep16m")
    check = {"sweeps": 0, "bad_sweeps": 0, "first_bad": None}

    def sweep():
        nonlocal n_refl
        stack = ffs_amd.Stack3D(ctx)
        inflight = []
        ok = True
        for b in range(NZ // B + len(streams)):
            if len(inflight) == len(streams) or (b >= NZ // B and inflight):
                b0, s = inflight.pop(0)
                s.wait_counts()
                raw = s.last_frame_counts
                if expected is not None:
                    sl = slice(b0 * B, b0 * B + B)
                    ok = ok and np.array_equal(raw["num_strong_pixels"], expected["num_strong_pixels"][sl]) and np.array_equal(raw["n_boxes"], expected["n_boxes"][sl])
                stack.add_batch(s)
            if b < NZ // B:
                s = streams[b % len(streams)]
                s.submit_device(d_frames.data_ptr() + b * B * fstride, pitch, fstride, B, first_frame_id=b * B)
                inflight.append((b, s))
        refl, n_calc, fs, fp = stack.finish()
        finish_ms.append(stack.last_finish_ms())
        n_refl = len(refl)
        if expected is not None:
            got = (len(refl), n_calc, fs, fp)
            want = (expected["n_reflections"], expected["n_calculated"], expected["n_filtered_size"], expected["n_filtered_sep"])
            ok = ok and got == want and fixtures.reflections_digest(refl) == expected["digest"]
            check["sweeps"] += 1
            if not ok:
                check["bad_sweeps"] += 1
                if check["first_bad"] is None:
                    check["first_bad"] = {"got": [int(v) for v in got], "want": [int(v) for v in want]}
        stack.close()

    for _ in range(max(1, args.warmup // 2)):
        sweep()
    torch.cuda.synchronize(dev)
    finish_ms.clear()
    check.update(sweeps=0, bad_sweeps=0, first_bad=None)
    steps = max(1, args.steps // 10)
    t0 = time.perf_counter()
    for _ in range(steps):
        sweep()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    results_checked = None if expected is None else bool(check["sweeps"] > 0 and check["bad_sweeps"] == 0)
    out = {
        "metric": "detector frames/s (Eiger-16M 100-frame sweep, 2D + 3D connected components)", "value": round(steps * NZ / elapsed, 1),
        "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": max(1, args.warmup // 2), "ms_per_step": round(elapsed / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "sweep16m: 100 x 4148x4362 uint16 frames resident in HBM, 800 reflections with a rocking curve, "
                               "min_spot_size 3, min_spot_size_3d 15; one step = whole sweep incl. ffs_stack3d_finish",
                   "frames_per_batch": B, "streams": len(streams), "reflections_3d": n_refl,
                   "stack3d_finish_device_ms": round(float(np.mean(finish_ms)), 3)},
        "results_checked": results_checked,
        "results_check": ({"sweeps_compared_in_the_timed_region": check["sweeps"], "bad_sweeps": check["bad_sweeps"], "first_bad": check["first_bad"],
                           "what": "every frame's strong pixels and boxes, the 3D reflection table's counts (found, calculated, filtered by size / "
                                   "separation) and its digest over every field",
                           "against": "tests/golden/bench_workloads.npz (sweep16m: reference standalone.cc threshold + restated 2D / 3D connected "
                                      "components; tests/golden/make_golden_bench.py --only sweep16m)"}
                          if expected is not None else {"skipped": "no committed oracle results for this sweep"}),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_sweep(p, mask, NZ)
    print(json.dumps(out), flush=True)
    return 0 if results_checked is not False else 6


def cpu_baseline_sweep(p, mask, NZ, n_sample=32):
    """The sweep's CPU path on this box's host cores, on a bounded sample: the first `n_sample` frames of the sweep through the
    reference's standalone.cc threshold (oracle/_ref; else the restatement) + the oracle's 2D components, one frame per thread on
    16 threads (the reference's threading model, spotfinder/spotfinder.cc:725-752), then the oracle's 3D labelling of those slices
    on one thread (the reference labels after all frames, serially: connected_components.cc:270-470)."""
    from concurrent.futures import ThreadPoolExecutor
    from ffs_amd import synth
    from oracle import oracle as O
    H, W = mask.shape
    hi = host_info()
    cores = max(1, min(hi["affinity_cores"], int(os.environ.get("FFS_BENCH_CPU_THREADS", "16"))))
    frames = synth.frames(p, range(n_sample), threads=cores)
    kind = "reference" if O.have_ref() else "port"

    def worker(idx_list):   # (one spot-finder object per thread, as cpu_baseline's: its tables are allocated once)
        sf = O.RefSpotfinder(W, H) if kind == "reference" else O.PortSpotfinder(W, H)
        dst = np.empty((H, W), np.uint8)
        out = []
        for i in idx_list:
            sf.run_f64(frames[i].astype(np.float64), mask, dst)      # the reference converts too (spotfinder.cc:1024)
            cc = O.cc2d(dst, frames[i], 3)
            out.append((i, cc.k, cc.intensity))
        return out
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(worker, [list(range(c, n_sample, cores)) for c in range(cores)]))
    slices = [(k, it) for _, k, it in sorted((r for part in parts for r in part), key=lambda r: r[0])]
    t1 = time.perf_counter()
    want = O.cc3d(slices, W, H, 15, 2.0)
    t2 = time.perf_counter()
    return {"value": round(n_sample / (t2 - t0), 3), "unit": "frames/s", "cores": cores, "kind": kind, "nproc": hi["affinity_cores"],
            "cpu_model": hi["cpu_model"], "host_cores": hi["nproc"],
            "frames_2d_s": round(t1 - t0, 3), "labelling_3d_s": round(t2 - t1, 3), "reflections_3d_in_sample": int(len(want.reflections)),
            "sample": f"the first {n_sample} of the sweep's {NZ} frames: threshold = " + ("reference baseline/spotfinder/standalone.cc (oracle/_ref)" if kind == "reference" else "oracle port")
                      + f" + oracle 2D components, one frame per thread on {cores} threads, then the oracle's 3D labelling of the {n_sample} slices on one thread; "
                      + f"{t2 - t0:.1f} s wall (the host shows {hi['nproc']} cores)"}


# ---------------------------------------------------------------------------------------------------
def bench_single_process(args):
    """--single-process --gpus N: ONE process, one context + `streams` ffs_streams + one host thread per GPU,
    contexts registered with ffs_multi_init (the model of `spotfinder --gpus N`, host/spotfinder.cc).  Each thread
    runs the submit/wait pipeline natively (ffs_bench_pipeline: no interpreter lock in the timed region).  No data-path
    collective: frames are independent; value = frames of all GPUs / wall time from the common start to the last
    GPU's finish."""
    import torch
    import ffs_amd
    n = args.gpus
    have = torch.cuda.device_count()
    # --devices 0,0: a rehearsal with several contexts on one GPU (the one-GPU boxes of the pool)
    devs = [int(t) for t in args.devices.split(",")] if args.devices else list(range(n))
    if len(devs) != n or max(devs) >= have:
        print(f"bench.py --single-process: --gpus {n} asked for, {have} GPU(s) visible (devices {devs})", file=sys.stderr)
        return 3
    W, H, dt, bytes_per_px = WORKLOADS[args.workload]
    B = args.batch
    transport = ffs_amd.api.multi_init(devs)
    gpus = []
    for d in range(n):
        frames, mask = make_inputs(args.workload, B, d)
        ctx = ffs_amd.Context(W, H, dt, max_batch=B, device=devs[d])
        ctx.set_mask(mask)
        ctx.set_params(want_reflections=1, algorithm=1 if args.algorithm == "dispersion_extended" else 0)
        pitch, fstride = ctx.device_layout()
        host = np.zeros((B, H, pitch // np.dtype(dt).itemsize), dt)
        host[:, :, :W] = frames
        d_frames = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(torch.device("cuda", devs[d]))
        del host, frames
        gpus.append({"ctx": ctx, "streams": [ctx.stream() for _ in range(max(1, args.streams))], "buf": d_frames,
                     "pitch": pitch, "fstride": fstride})

    def region(steps):
        start = threading.Barrier(n + 1)
        res = [None] * n

        def work(d):
            g = gpus[d]
            start.wait()
            res[d] = ffs_amd.api.bench_pipeline(g["streams"], g["buf"].data_ptr(), g["pitch"], g["fstride"], B, steps,
                                                first_frame_id=d * steps * B)
        th = [threading.Thread(target=work, args=(d,)) for d in range(n)]
        for t in th:
            t.start()
        for d in set(devs):
            torch.cuda.synchronize(d)
        start.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        for d in set(devs):
            torch.cuda.synchronize(d)
        return time.perf_counter() - t0, res

    region(args.warmup)
    times, last = [], None
    for _ in range(max(1, args.reps)):
        el, last = region(args.steps)
        times.append(el)
    el2, _ = region(2 * args.steps)
    # self-check: each context's totals over its last timed region against the committed oracle results for its frames
    from ffs_amd import fixtures
    results_checked = True
    for d in range(n):
        exp = fixtures.load_expected(args.workload, args.algorithm, d, B)
        if exp is None:
            results_checked = None
            break
        want = (int(exp["n_boxes"].sum()) * args.steps, int(exp["num_strong_pixels"].sum()) * args.steps)
        if tuple(int(v) for v in last[d]) != want:
            results_checked = False
    med = statistics.median(times)
    steady = max(0.0, (el2 - med) / args.steps)
    # The gather of one step's spot rows (every context's last batch), two ways: read where ffs_wait left them (host memory, one
    # memcpy per context: what `spotfinder --gpus N` does) against the north star's collective on the library's own communicators
    # (ffs_multi_gather_rows: counts by ncclAllGather, rows by ncclSend / ncclRecv to the root device, one D2H).  Untimed A/B.
    gather_ab = None
    try:
        last_streams = [g["streams"][(args.steps - 1) % len(g["streams"])] for g in gpus]
        cap = 1 << 18
        scratch = np.empty((cap + 1, 4), np.float32)
        t_host, t_rccl, n_host, n_rccl = [], [], 0, 0
        for rep in range(6):
            t0 = time.perf_counter()
            n_host = 0
            for s in last_streams:
                n_host += s.pack_spot_centres(scratch[n_host:], cap - n_host)
            t_host.append(time.perf_counter() - t0)
            if transport == "rccl":
                t0 = time.perf_counter()
                rows = ffs_amd.api.multi_gather_rows(last_streams, root=0, cap=cap)
                t_rccl.append(time.perf_counter() - t0)
                n_rccl = len(rows)
                same = n_rccl == n_host and np.array_equal(rows.view(np.uint32), scratch[:n_host].view(np.uint32))
            else:
                same = None
        gather_ab = {"rows_per_step": int(n_host), "host_read_ms": round(min(t_host[1:]) * 1e3, 4),
                     "rccl_gather_ms": (round(min(t_rccl[1:]) * 1e3, 4) if t_rccl else None), "rows_identical": same,
                     "what": "one step's spot-centre rows of all contexts into one host array: memcpy from the library's host arrays against "
                             "ffs_multi_gather_rows (H2D of each context's rows, ncclAllGather of counts, ncclSend / ncclRecv to the root device, D2H)"}
    except Exception as e:  # the line must still come out
        gather_ab = {"error": f"{type(e).__name__}: {e}"}
    out = {
        "metric": "detector frames/s (Eiger-16M 4362x4148 uint16)" if args.workload == "eiger16m" else f"detector frames/s ({args.workload})",
        "value": round(n * args.steps * B / med, 2), "unit": "frames/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(med / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16" if dt == np.uint16 else "u32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {W}x{H} {np.dtype(dt).name}, 7x7 dispersion window, {B} frames/step/GPU resident "
                               "in HBM, spots+centroids returned to host", "frames_per_step_per_gpu": B, "streams": args.streams,
                   "parallelism": f"single process, {n} contexts on devices {devs} (ffs_multi_init, transport for rotation lists: "
                                  f"{transport}), one host thread per context, no data-path collective",
                   "spots_per_frame": round(sum(r[0] for r in last) / max(1, n * args.steps * B), 1),
                   "strong_pixels_per_frame": round(sum(r[1] for r in last) / max(1, n * args.steps * B), 1)},
        "repetitions": {"n": len(times), "ms_per_step": [round(t / args.steps * 1e3, 4) for t in times], "value_from": "median"},
        "steady_ms_per_step": round(steady * 1e3, 4), "drain_ms": round(max(0.0, med - steady * args.steps) * 1e3, 4),
        "n_contexts_seen": n,
        "results_checked": results_checked,
        "gather_ab": gather_ab,
    }
    print(json.dumps(out), flush=True)
    return 0 if results_checked is not False else 6


# ---------------------------------------------------------------------------------------------------
def dry_run(args):
    """Rehearsal of the launcher and of the N-rank plumbing on CPU (`--dry-run`, tests): gloo process group, fake
    spot rows through the same pack -> all_gather -> count path, no GPU touched, value 0."""
    import torch
    import torch.distributed as dist
    from ffs_amd import dist as D
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("FFS_BENCH_DRYRUN_FAIL_RANK") == str(rank):     # (tests: a rank that dies before the rendezvous)
        return 7
    if world > 1:
        dist.init_process_group("gloo")
    cap = 64
    rows = np.zeros((cap + 1, 4), np.float32)
    n_mine = 0 if (world > 2 and rank == 1) else 3 + rank          # unequal counts, and an empty rank when there are three or more
    rows[:n_mine, 0] = D._ids_as_float_lanes([rank * 10 + 1])[0]
    rows[:n_mine, 1:] = rank + 0.5
    rows[cap].view(np.uint32)[:3] = (n_mine, n_mine, rank + 1)
    t = torch.from_numpy(rows)
    if args.gather == "rows" and world > 1:
        got, counts, reqs = D.gather_rows_to_root(t, n_mine, tag=rank + 1, root=0)
        for r in reqs:
            r.wait()
        seen = len({int(v) for v in counts[:, 1] if v})
        if rank == 0:
            want = sum(0 if (world > 2 and r == 1) else 3 + r for r in range(world))
            assert got.shape[0] == want == int(counts[:, 0].sum()), (got.shape, want)
            assert sorted(D.rows_by_frame(got.numpy())) == sorted(r * 10 + 1 for r in range(world) if not (world > 2 and r == 1))
    else:
        g = D.all_gather_fixed(t).numpy() if world > 1 else rows
        seen = ranks_seen(g.reshape(world, cap + 1, 4), cap)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU)", "value": 0.0, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "n_ranks_seen": seen}), flush=True)
    return 0


def ranks_seen(blocks, cap):
    """Distinct rank tags (third word of every block's count row, rank + 1) in a gathered buffer."""
    tags = {int(b[cap].view(np.uint32)[2]) for b in blocks}
    tags.discard(0)
    return len(tags)


# ---------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-settle", action="store_true", help="no untimed settling regions between the warm-up steps and the timed repetitions")
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the timed `steps`-step region; `value` comes from the median one")
    ap.add_argument("--batch", type=int, default=32, help="frames per step and per GPU")
    ap.add_argument("--workload", default="eiger16m", choices=sorted(WORKLOADS) + ["sweep16m"],
                    help="sweep16m = BASELINE.json configs[4]: a 100-frame Eiger-16M rotation sweep through the 2D path, "
                         "the device-resident 3D stack and its finish (one step = one whole sweep)")
    ap.add_argument("--streams", type=int, default=4, help="batches in flight per GPU (measured: 2 -> 52.8 k, 3 -> 59.9 k, 4 -> 61.1 k, 6 -> 64.1 k frames/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algorithm", default="dispersion", choices=["dispersion", "dispersion_extended"],
                    help="dispersion = the headline metric; dispersion_extended = the second algorithm of the "
                         "same CLI flag (SURVEY 8f), reported as its own metric")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N>1: batches whose spot lists are gathered by one RCCL collective (8: the count exchange of gather k has "
                         "2.6 ms -- eight steps -- to get through the GPU's queues before gather k + 1 needs it; with 4 the submit "
                         "thread waited for it 0.4 ms per step, profiles/r04h_gather_ab.txt)")
    ap.add_argument("--gather", default="rows", choices=["rows", "padded"],
                    help="N>1: rows = all_gather of the ranks' row counts, then exactly the written (frame_id, x, y, z) rows point to "
                         "point to rank 0 (north_star's gather); padded = one all_gather_into_tensor of fixed-size blocks to every rank (A/B)")
    ap.add_argument("--no-streamed", action="store_true",
                    help="skip the two short PCIe-inclusive legs (pinned host frames -> ffs_submit, and bitshuffle-LZ4 "
                         "chunks -> ffs_submit_compressed); they are reported as streamed_frames_per_s / "
                         "streamed_compressed, never as `value`")
    ap.add_argument("--streamed", action="store_true", help="(kept for old command lines: the streamed legs are on by default)")
    ap.add_argument("--no-cli-e2e", action="store_true",
                    help="skip the end-to-end leg of the drop-in binary (bin/spotfinder on a 1000-frame Eiger-stream directory)")
    ap.add_argument("--cli-images", type=int, default=1000)
    ap.add_argument("--cli-long-images", type=int, default=4096,
                    help="images of the long end-to-end run (every image a 7.5 MB file in /dev/shm: 30 GB of tmpfs for 4096); 0 skips it; "
                         "shrunk by itself when the file system or the memory cgroup has not the room")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N driven from ONE process: a context and a host thread per GPU behind ffs_multi_init "
                         "(the C++ driver's model) instead of one rank per GPU")
    ap.add_argument("--devices", default="", help="--single-process: device index of every context, e.g. 0,0 to rehearse two contexts on one GPU")
    ap.add_argument("--dry-run", action="store_true", help="rehearse launcher + gather plumbing on CPU (gloo), no GPU")
    ap.add_argument("--launch-timeout", type=float, default=1500.0)
    ap.add_argument("--tune", default="", help="ffs_ctx_set_tuning pairs for A/B runs, 'key=value,key=value' (results are the same)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched and not args.single_process:
        sys.exit(launch_ranks(args))        # this process never touches the GPU
    if args.dry_run:
        sys.exit(dry_run(args))
    if args.single_process and args.gpus > 1:
        sys.exit(bench_single_process(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # The drop-in binary end to end, before this process initialises the GPU (its children are exec'ed)
    e2e = None
    if (world == 1 and rank == 0 and not args.no_cli_e2e and args.workload == "eiger16m"
            and args.algorithm == "dispersion"):
        e2e = cli_e2e(n_images=args.cli_images, long_images=args.cli_long_images)

    import torch
    if world > 1 and torch.cuda.device_count() <= local_rank:
        print(f"bench.py rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible: not started",
              file=sys.stderr)
        sys.exit(3)
    dist = None
    if world > 1 or os.environ.get("FFS_BENCH_FORCE_DIST"):  # the env var rehearses the RCCL path on one GPU
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout carries the one JSON line
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    dev = torch.device("cuda", local_rank)

    import ffs_amd
    if args.workload == "sweep16m":
        return bench_sweep(args, dev, local_rank)
    W, H, dt, bytes_per_px = WORKLOADS[args.workload]
    B = args.batch
    n_unique = B
    frames, mask = make_inputs(args.workload, n_unique, rank)

    ctx = ffs_amd.Context(W, H, dt, max_batch=B, device=local_rank)
    if args.tune:
        ctx.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",") if kv})
    ctx.set_mask(mask)
    ext = args.algorithm == "dispersion_extended"
    ctx.set_params(want_reflections=1, algorithm=1 if ext else 0)
    pitch, fstride = ctx.device_layout()
    # inputs resident in HBM, in the library's pitched layout
    host = np.zeros((B, H, pitch // np.dtype(dt).itemsize), dt)
    host[:, :, :W] = frames
    d_frames = torch.from_numpy(host.view(np.uint8).reshape(-1)).to(dev)
    del host
    ptr = d_frames.data_ptr()
    streams = [ctx.stream() for _ in range(max(1, args.streams))]

    # ---- self-check (the reference's driver has one too, --validate: spotfinder/spotfinder.cc:1012-1053): every batch waited for
    # inside the timed regions has its per-frame (n_boxes, num_strong_pixels) compared with the committed oracle results for these
    # frames (tests/golden/bench_workloads.npz, generated by tests/golden/make_golden_bench.py), and after the timing one batch
    # per stream has every box and reflection compared through their digests.  A mismatch fails the run.
    from ffs_amd import fixtures
    expected = fixtures.load_expected(args.workload, args.algorithm, rank, B)
    check = {"batches": 0, "bad_batches": 0, "first_bad": None}

    def check_counts(st):
        if expected is None:
            return
        raw = st.last_frame_counts
        check["batches"] += 1
        if not (np.array_equal(raw["n_boxes"], expected["n_boxes"]) and np.array_equal(raw["num_strong_pixels"], expected["num_strong_pixels"])):
            check["bad_batches"] += 1
            if check["first_bad"] is None:
                f = int(np.flatnonzero((raw["n_boxes"] != expected["n_boxes"]) | (raw["num_strong_pixels"] != expected["num_strong_pixels"]))[0])
                check["first_bad"] = {"frame_in_batch": f, "got": [int(raw["n_boxes"][f]), int(raw["num_strong_pixels"][f])],
                                      "want": [int(expected["n_boxes"][f]), int(expected["num_strong_pixels"][f])]}

    # ---- N>1: gather of the per-frame spot lists -------------------------------------------------
    # One RCCL collective per `gather_every` batches ("one small collective per batch of
    # frames, not per frame"): every rank contributes a fixed-size block of (frame_id, x, y, z) rows.
    use_dist = dist is not None
    G = max(1, args.gather_every)
    spot_cap = 2048 * B
    # Two sets of buffers: a group's pinned block must not be overwritten while its H2D copy is in flight.
    rows_mode = args.gather == "rows"
    if use_dist and rows_mode:
        # rows of a group's G batches end to end (+ one scratch row ffs_stream_spot_centres puts its counts in)
        pin = (lambda t: t.pin_memory()) if os.environ.get("FFS_BENCH_PACK_PINNED", "1") != "0" else (lambda t: t)
        pack_host = [pin(torch.empty((G * spot_cap + 1, 4), dtype=torch.float32)) for _ in range(2)]
        pack_dev = [torch.empty((G * spot_cap + 1, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        recv_buf = [torch.empty((world * G * spot_cap if rank == 0 else 1, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        buf_free = [None, None]
        pack_np = [t.numpy() for t in pack_host]
    elif use_dist:
        pack_host = [torch.empty((G, spot_cap + 1, 4), dtype=torch.float32).pin_memory() for _ in range(2)]
        pack_dev = [torch.empty((G, spot_cap + 1, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        gather_buf = [torch.empty((world * G, spot_cap + 1, 4), dtype=torch.float32, device=dev) for _ in range(2)]
        buf_free = [None, None]            # event after which pack_host[b] may be overwritten
        pack_np = [t.numpy() for t in pack_host]
    cur, pending = 0, 0
    rows_at = 0             # rows mode: rows packed into pack_host[cur] so far
    gather_s = [0.0, 0.0]   # host seconds: packing, collectives
    gather_t = {"h2d": 0.0, "begin": 0.0, "finish": 0.0, "buf_sync": 0.0}   # where a flush's host time goes
    last_gather = [None]    # index of the gather buffer the newest collective filled
    gathered_rows = [0, 0]  # rows mode: rows landed on rank 0, collectives
    tags_seen = set()

    # The (frame_id, x, y, z) rows come out of the library with one memcpy per batch (ffs_stream_spot_centres: the rows are laid
    # down while ffs_wait assembles the reflections; walking the 72-byte records again was 0.3 ms per batch, as long as a step),
    # and one gather is always in flight: a flush begins gather k (counts all_gather, nothing waited for) and finishes gather
    # k - 1 (its counts are on the host by then: sends / receives posted).  (A helper thread for the packing was tried: handing
    # a job to a Python thread costs 0.5-0.7 ms of interpreter-lock hand-over, more than the work.)
    in_flight = [None]      # rows mode: (handle, buffer index) of the gather begun and not yet finished
    gather_scratch = None
    # (the gather's copies and collectives stay on torch's current stream: a stream of their own -- one more beside the library's
    # four -- made every collective's launch wait longer for a hardware queue: 70 k frames/s against 83 k in the one-rank rehearsal)
    if use_dist and rows_mode:
        from ffs_amd import dist as D
        gather_scratch = D.RowGatherScratch(dev, slots=3)

    def pack_job(stream, b, slot):
        nonlocal rows_at
        if rows_mode:
            room = G * spot_cap - rows_at
            rows_at += stream.pack_spot_centres(pack_np[b][rows_at:], room)   # (FFS_ERR_OVERFLOW when a batch does not fit)
        else:
            row = pack_np[b][slot]
            stream.pack_spot_centres(row, spot_cap)
            row[spot_cap].view(np.uint32)[2] = rank + 1          # who packed this block (n_ranks_seen)

    def finish_rows_gather():
        from ffs_amd import dist as D
        if in_flight[0] is None:
            return
        h, b = in_flight[0]
        in_flight[0] = None
        t0 = time.perf_counter()
        got, counts, reqs = D.gather_rows_finish(h, pack_dev[b], root=0, recv_buf=recv_buf[b])
        for r in reqs:
            r.wait()                        # (nccl: orders the current stream behind the transfers, no host wait)
        gather_t["finish"] += time.perf_counter() - t0
        gathered_rows[0] += int(counts[:, 0].sum())
        gathered_rows[1] += 1
        tags_seen.update(int(t) for t in counts[:, 1] if t)
        ev = torch.cuda.Event()
        ev.record()
        buf_free[b] = ev

    def flush_gather(last=False):
        if not use_dist:
            return
        _flush_gather(last)

    def _flush_gather(last):
        nonlocal cur, pending, rows_at
        tg = time.perf_counter()
        if pending:
            if rows_mode:
                from ffs_amd import dist as D
                n = rows_at
                rows_at = 0
                t0 = time.perf_counter()
                if n:
                    pack_dev[cur][:n].copy_(pack_host[cur][:n], non_blocking=True)
                t1 = time.perf_counter()
                h = D.gather_rows_begin(n, rank + 1, dev, scratch=gather_scratch)
                gather_t["h2d"] += t1 - t0
                gather_t["begin"] += time.perf_counter() - t1
                finish_rows_gather()                 # the one before this
                in_flight[0] = (h, cur)
            else:
                for g in range(pending, G):      # unused slots of a partial group: zero spots
                    pack_host[cur][g, spot_cap] = 0          # (rows written, rows wanted) = (0, 0)
                pack_dev[cur].copy_(pack_host[cur], non_blocking=True)
                dist.all_gather_into_tensor(gather_buf[cur].view(-1), pack_dev[cur].view(-1))
                ev = torch.cuda.Event()
                ev.record()
                buf_free[cur] = ev
            last_gather[0] = cur
            cur, pending = cur ^ 1, 0
        if last and rows_mode:
            finish_rows_gather()                     # every frame's spots are on rank 0 before the clock stops
        gather_s[1] += time.perf_counter() - tg

    def gather(results, stream):
        nonlocal pending
        if not use_dist:
            return
        tg = time.perf_counter()
        if pending == 0:
            if rows_mode and in_flight[0] is not None and in_flight[0][1] == cur:
                finish_rows_gather()                 # (this buffer's gather is still in flight: only with fewer than two flushes between)
            if buf_free[cur] is not None:
                t0 = time.perf_counter()
                buf_free[cur].synchronize()
                gather_t["buf_sync"] += time.perf_counter() - t0
        pack_job(stream, cur, pending)
        pending += 1
        gather_s[0] += time.perf_counter() - tg
        if pending == G:
            flush_gather()

    strong_px = 0
    # Duration of every batch's threshold stage inside the timed regions, from the HIP events that ride on ITS dispatch (the stream's
    # own start / stop events of the streaming kernel: ffs_stream_timings): the kernel as it runs in the pipeline, beside the other
    # batches' sparse launches -- what `rocprofv3 --kernel-trace --stats` of this command averages too.
    thr_ms = []

    def reap(done):
        nonlocal strong_px
        _, nbx, nst = done.wait_counts()      # (every frame's boxes and centroids are in the library's host arrays)
        check_counts(done)
        thr_ms.append(done.timings()["threshold"])
        gather(None, done)
        strong_px += nst
        return nbx

    def run_steps(k):
        """k steps, `streams` batches in flight"""
        inflight = []
        spots = 0
        for step in range(k + len(streams)):
            if step < k:
                s = streams[step % len(streams)]
                if len(inflight) == len(streams):
                    spots += reap(inflight.pop(0))
                s.submit_device(ptr, pitch, fstride, B, first_frame_id=(rank * k + step) * B)
                inflight.append(s)
            elif inflight:
                spots += reap(inflight.pop(0))
        flush_gather(last=True)          # every frame's spots are gathered before the clock stops
        return spots

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(k):
        """EXACTLY k steps between barrier + synchronize on both sides; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        sp = run_steps(k)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, sp

    # The untimed kernel-alone legs run FIRST (round 5; they followed the timed regions before): the memory ceiling of this box
    # beside the nominal 8 TB/s, the threshold kernel launched alone (`ms_per_launch_alone`) and the same
    # with the dense byte mask as an output (the reference kernel's result_strong, 1 B/px: the hot path's own strong mask is the
    # bit plane, the byte mask is written only on request).  Nothing is added to the run -- but a GPU that
    # sat idle while Python loaded fixtures comes up through its clock states over ~20 ms of work, and with the timed regions
    # first the early repetitions (0.36, 0.34, 0.33 ms per step against 0.32 from the fourth on) paid for it.
    # (40 iterations of the ceiling probe, ~25 ms: tools/rep_probe.py -- the streaming kernel takes 0.325 ms in the first 20 steps after
    # half a second of idleness and 0.292 ms from the fourth repetition on, and the ceiling itself is under-read on cold clocks)
    peak_read, peak_mix = streams[0].bench_hbm(iters=40)
    ms_cand, ms_exact = streams[0].bench_threshold(ptr, pitch, fstride, B, iters=10)
    ms_dense = None
    if not ext:
        ctx.set_params(want_reflections=1, algorithm=0, want_strong_mask=1)
        ms_dense, _ = streams[0].bench_threshold(ptr, pitch, fstride, B, iters=10)
        ctx.set_params(want_reflections=1, algorithm=0, want_strong_mask=0)
    run_steps(args.warmup)
    # Untimed, after the W warm-up steps the driver asked for: regions of the same K steps until two in a row agree within 1 % (at most
    # eight).  The GPU's clocks follow its load over tens of milliseconds; W = 5 steps are 1.5 ms, and without this the repetitions below
    # read 0.322, 0.311, 0.303, 0.295, 0.297 ms per step (profiles/r06q_bench_lines_steps20.jsonl) -- a ramp, not the pipeline.  Every
    # one of these regions is listed in the line (`settling_ms_per_step`); the timed repetitions that follow are all listed too.
    settling = []
    if not args.no_settle:
        for _ in range(8):
            el, _ = timed(args.steps)
            settling.append(el)
            if len(settling) >= 2 and abs(settling[-1] - settling[-2]) < 0.01 * settling[-2]:
                break
    thr_ms.clear()
    # `reps` repetitions of the same `steps`-step region: the timed region is a few ms, one slow dispatch moves a single
    # repetition by several per cent -- `value` is the median repetition
    times = []
    spots = 0
    for _ in range(max(1, args.reps)):
        strong_px = 0
        el, spots = timed(args.steps)
        times.append(el)
    strong_steps = strong_px
    elapsed = statistics.median(times)
    # the fixed tail of a run (the last batches' sparse launches drain after the last streaming kernel), made visible:
    # steady = (T(2K) - T(K)) / K, drain = T(K) - K steady
    thr_timed = list(thr_ms)          # (the batches of the `reps` timed regions)
    el2, _ = timed(2 * args.steps)
    steady = max(0.0, (el2 - elapsed) / args.steps)
    drain = max(0.0, elapsed - steady * args.steps)
    digests_ok = None
    if expected is not None:          # untimed: every box and reflection of one more batch per stream, through their digests
        digests_ok = True
        for st in streams:
            st.submit_device(ptr, pitch, fstride, B, first_frame_id=0)
        for st in streams:
            for f, fr in enumerate(st.wait(copy=False)):
                if fixtures.frame_digest(fr.boxes, fr.reflections) != expected["digest"][f].tobytes():
                    digests_ok = False
    results_checked = None if expected is None else bool(check["bad_batches"] == 0 and check["batches"] > 0 and digests_ok)
    if dist is not None and expected is not None:     # every rank checks its own shard; the line reports all of them
        t = torch.tensor([0 if results_checked else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        results_checked = bool(int(t.item()) == 0)
    n_ranks_seen = 1
    if use_dist and rows_mode:
        n_ranks_seen = len(tags_seen)
    elif use_dist and last_gather[0] is not None:
        torch.cuda.synchronize(dev)
        g = gather_buf[last_gather[0]].cpu().numpy()
        n_ranks_seen = ranks_seen(g, spot_cap)

    alg_bytes = float(W) * H * bytes_per_px * B
    # The kernel's duration for the roofline: standard algorithm = the average over every batch of the timed regions, each from the
    # events on its own dispatch (the contract: "average launch duration, measured live with HIP events over the timed region");
    # `ms_alone` = the same kernel launched alone in a loop (ffs_bench_threshold: each launch behind the fills that reset the
    # stream's planes, i.e. on cold mask tables).  Extended algorithm: its first pass alone (the stage's events span three kernels).
    ms_alone = ms_cand
    if not ext and thr_timed:
        ms_cand = float(sum(thr_timed) / len(thr_timed))
    achieved = alg_bytes / (ms_cand * 1e-3) / 1e9
    tm = streams[0].timings()
    traffic, traffic_src = pmc_traffic(args.workload + ("_extended" if ext else ""), B)
    traffic_dense, traffic_dense_src = pmc_traffic(args.workload + "_dense", B)

    streamed = None
    if not args.no_streamed:
        # host frames in each stream's pinned staging buffer -> H2D + full hot path per batch
        bufs = []
        for st in streams:
            hb = st.host_buffer()
            hb[:B] = frames[:B]
            bufs.append(hb)
        n_st = 24   # short legs: ~0.5 s raw, ~0.1 s compressed, whatever --steps says
        for i, st in enumerate(streams):   # (untimed: the first copy out of a fresh staging area takes ~8 ms, alloc_cost.hip)
            st.submit(bufs[i][:B], first_frame_id=0)
        for st in streams:
            st.wait()
        barrier()
        ts = time.perf_counter()
        inflight = []
        for step in range(n_st + len(streams)):
            if step < n_st:
                i = step % len(streams)
                if len(inflight) == len(streams):
                    inflight.pop(0).wait()
                streams[i].submit(bufs[i][:B], first_frame_id=step * B)
                inflight.append(streams[i])
            elif inflight:
                inflight.pop(0).wait()
        barrier()
        streamed = n_st * B / (time.perf_counter() - ts)

        # the same, with the frames arriving as the detector writes them: bitshuffle-LZ4 chunks placed in
        # the pinned staging buffer, decoded on the GPU (ffs_submit_compressed)
        from ffs_amd import bslz4
        n_cmp = min(4, B)                                  # numpy encoder: a few unique frames, repeated
        uniq = [np.frombuffer(bslz4.compress(frames[i]), np.uint8) for i in range(n_cmp)]
        views = []
        for st in streams:
            hb, at, v = st.host_bytes(), 0, []
            for i in range(B):
                c = uniq[i % n_cmp]
                hb[at:at + c.size] = c
                v.append(hb[at:at + c.size])
                at += (c.size + 63) & ~63
            views.append(v)
        ms_dec, _ = streams[0].decode_only(views[0], iters=5, want_frames=False)
        for i, st in enumerate(streams):   # (untimed warm-up, as above)
            st.submit_compressed(views[i], first_frame_id=0)
        for st in streams:
            st.wait()
        submit_ms = 0.0
        barrier()
        ts = time.perf_counter()
        inflight = []
        for step in range(n_st + len(streams)):
            if step < n_st:
                i = step % len(streams)
                if len(inflight) == len(streams):
                    inflight.pop(0).wait()
                tq = time.perf_counter()
                streams[i].submit_compressed(views[i], first_frame_id=step * B)
                submit_ms += (time.perf_counter() - tq) * 1e3 / n_st
                inflight.append(streams[i])
            elif inflight:
                inflight.pop(0).wait()
        barrier()
        streamed_cmp = n_st * B / (time.perf_counter() - ts)
        chunk_mb = sum(c.size for c in uniq) / n_cmp / 1e6

    out = None
    if rank == 0:
        total_frames = world * args.steps * B
        phys = (traffic if traffic else alg_bytes) / (ms_cand * 1e-3) / 1e9
        out = {
            "metric": ("detector frames/s (Eiger-16M 4362x4148 uint16)" if args.workload == "eiger16m"
                       else f"detector frames/s ({args.workload})") + (", extended dispersion" if ext else ""),
            "value": round(total_frames / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16" if dt == np.uint16 else "u32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H} {np.dtype(dt).name}, "
                                   + ("extended dispersion (7x7 first pass, 5x5 erosion, 11x11 second pass), " if ext
                                      else "7x7 dispersion window, ") +
                                   f"{B} frames/step/GPU resident in HBM, spots+centroids returned to host",
                       "frames_per_step_per_gpu": B, "streams": len(streams),
                       "parallelism": (f"one process per GPU x {world}, frame queue sharded, no data-path collective" if use_dist
                                       else "single GPU"),
                       "spot_gather": ((f"every {G} batches: all_gather of the ranks' row counts, then exactly the written rows by RCCL "
                                        "send/recv (one group) to rank 0" if rows_mode else
                                        f"RCCL all_gather_into_tensor of padded blocks to every rank every {G} batches")
                                       if use_dist else "none (single GPU)"),
                       "spots_per_frame": round(spots / max(1, args.steps * B), 1),
                       "strong_pixels_per_frame": round(strong_steps / max(1, args.steps * B), 1)},
            "repetitions": {"n": len(times), "ms_per_step": [round(t / args.steps * 1e3, 4) for t in times],
                            "value_from": "median repetition of the same steps-step region (barrier + synchronize on both sides of each)"},
            "steady_ms_per_step": round(steady * 1e3, 4),
            "drain_ms": round(drain * 1e3, 4),
            "settling_ms_per_step": [round(t / args.steps * 1e3, 4) for t in settling],
            "settling": "untimed regions of the same K steps after the W warm-up steps, until two in a row agree within 1 % (at most 8): the GPU's "
                        "clocks follow its load over tens of ms; --no-settle switches it off",
            "order_of_legs": "memory-ceiling probe (40 iterations) and the kernel-alone legs, THEN warm-up and the timed regions: a GPU that idled "
                             "while Python loaded the fixtures needs ~30 ms of load to reach its clocks (tools/rep_probe.py); every repetition is listed",
            "n_ranks_seen": n_ranks_seen,
            "results_checked": results_checked,
            "results_check": ({"batches_compared_in_timed_regions": check["batches"], "bad_batches": check["bad_batches"],
                               "first_bad": check["first_bad"], "digests_of_boxes_and_reflections_ok": digests_ok,
                               "against": "tests/golden/bench_workloads.npz (oracle results for this rank's frames: reference standalone.cc "
                                          "threshold + restated connected components; tests/golden/make_golden_bench.py)"}
                              if expected is not None else
                              {"skipped": "no committed oracle results for this workload / rank / batch size"}),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "measured_peak": {"read_only_GBps": round(peak_read, 1), "read_write_2to1_GBps": round(peak_mix, 1),
                                           "probe": "ffs_bench_hbm: linear 16 B/lane reads of the batch's pixel buffer; the same with an "
                                                    "8 B zero store per 16 B read (the kernel's read/write mix)"},
                         "frac_physical": (round(phys / HBM_PEAK_GBS, 4) if traffic else None),
                         "frac_of_measured_read": round(phys / max(peak_read, 1.0), 4),
                         "target": "frac_of_measured_read >= 0.80 (physical bytes / time / read ceiling measured in this run); `frac` is "
                                   "SURVEY 8(d)'s algorithmic fraction, which credits 2 B/px the kernel never moves (it reads 1.0 at 0.289 ms: above that it "
                                   "says nothing); `frac_physical` = PMC bytes / time / nominal 8 TB/s",
                         "kernel": ("k_stream_u16<2,true> (extended first pass)" if ext else
                                    "k_stream_u16 (whole threshold stage)" if dt == np.uint16 else "k_stream_u32 (whole threshold stage)"),
                         "ms_per_launch": round(ms_cand, 4),
                         "ms_per_launch_from": (f"mean over the {len(thr_timed)} batches of the timed regions, each from the HIP events on its own dispatch "
                                                "(in the pipeline, beside the other batches' sparse launches)" if (not ext and thr_timed) else
                                                "ffs_bench_threshold: the kernel launched alone, 10 launches"),
                         "ms_per_launch_alone": round(ms_alone, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "exact_kernel_ms_per_launch": round(ms_exact, 4),
                         "dense_mask": False,
                         "note": "algorithmic bytes = SURVEY 8(d): pixel + 1 mask byte read + 1 strong-mask byte written per pixel; the kernel "
                                 "reads the mask as bit tables and leaves the strong mask bit-packed (1 bit/px, non-zero words only), so its physical "
                                 "traffic is below the algorithmic figure (`traffic`, `physical_GBps`); `with_dense_mask` is the same kernel also "
                                 "writing the byte mask (want_strong_mask=1)",
                         "physical_GBps": (round(traffic / (ms_cand * 1e-3) / 1e9, 1) if traffic else None),
                         "with_dense_mask": ({"ms_per_launch": round(ms_dense, 4),
                                              "achieved": round(alg_bytes / (ms_dense * 1e-3) / 1e9, 1),
                                              "frac": round(alg_bytes / (ms_dense * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "traffic": traffic_dense, "traffic_source": traffic_dense_src,
                                              "frac_of_measured_mix": round((traffic_dense or alg_bytes) / (ms_dense * 1e-3) / 1e9 / max(peak_mix, 1.0), 4)}
                                             if ms_dense else None)},
            "stage_ms_last_batch": {k: round(v, 4) for k, v in tm.items()},
        }
        # the whole threshold stage (streaming kernel + fix-up) against the same algorithmic bytes
        out["roofline"]["threshold_stage_frac"] = round(
            alg_bytes / ((ms_cand + ms_exact) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if use_dist:
            n_timed = args.warmup + (len(times) + 2) * args.steps
            out["config"]["gather_host_ms_per_step"] = {"pack": round(gather_s[0] / n_timed * 1e3, 4),
                                                        "collective": round(gather_s[1] / n_timed * 1e3, 4),
                                                        **{k: round(v / n_timed * 1e3, 4) for k, v in gather_t.items()}}
            # bytes that land per step: on rank 0 only (rows), or on EVERY rank (padded blocks)
            out["config"]["gather_bytes_per_step"] = (int(gathered_rows[0] * 16 / max(1, gathered_rows[1]) / G) if rows_mode
                                                      else int(world * (spot_cap + 1) * 16))
            out["config"]["gather_lands_on"] = "rank 0" if rows_mode else "every rank"
        if streamed is not None:
            out["streamed_frames_per_s"] = round(streamed * world, 1)
            out["streamed_compressed"] = {
                "frames_per_s": round(streamed_cmp * world, 1), "chunk_MB": round(chunk_mb, 2),
                "compression_ratio": round(W * H * np.dtype(dt).itemsize / 1e6 / chunk_mb, 2),
                "decode_ms_per_batch": round(ms_dec, 4),
                "decode_out_GBps": round(W * H * np.dtype(dt).itemsize * B / (ms_dec * 1e-3) / 1e9, 1),
                "submit_call_ms": round(submit_ms, 3)}
        if e2e is not None:
            out["cli_e2e"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, mask, ext)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if results_checked is False:
        print(f"bench.py rank {rank}: results differ from the oracle's ({check})", file=sys.stderr)
        sys.exit(6)


if __name__ == "__main__":
    main()
