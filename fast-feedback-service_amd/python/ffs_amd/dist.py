"""Multi-GPU plumbing: one process per GPU, frames sharded over ranks, spot lists gathered.

The reference has no distributed layer (one process, one GPU: src/ffs/cuda_arg_parser.cc:56-61);
frames are independent units (spotfinder/spotfinder.cc:752), so the data path needs no collective.
The only exchange is the gather of per-frame results:
  * stills: (frame_id, x, y, z) per spot -> every rank (one collective per batch);
  * rotation sweeps: each frame's strong-pixel list (k, intensity) -> the rank that owns the 3D
    stack (ffs_stack3d_add_slice).
Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) with device tensors in bench.py,
"gloo" with CPU tensors in the tests.
"""
from __future__ import annotations

import numpy as np


def frame_shard(n_frames: int, rank: int, world: int) -> list[int]:
    """Frame i belongs to rank i mod world (static, deterministic partition of the frame queue)."""
    return list(range(rank, n_frames, world))


class SpotGatherTruncated(RuntimeError):
    """A rank had more spots than its block of the gather holds."""


def _ids_as_float_lanes(ids) -> np.ndarray:
    """Frame ids travel in a float32 lane of the (frame_id, x, y, z) rows.  As VALUES they would collide
    from 2^24 on, so the lane carries the low 32 bits of the id as a bit pattern."""
    return (np.asarray(ids, np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def pack_spots(results, cap: int) -> np.ndarray:
    """(cap + 1, 4) float32: rows = (frame_id bits, com_x, com_y, com_z); the last row holds, as uint32 bit
    patterns, (rows written, rows wanted) -- wanted > written means the block was too small."""
    out = np.zeros((cap + 1, 4), np.float32)
    n = wanted = 0
    for r in results:
        refl = r.reflections
        wanted += len(refl)
        m = min(len(refl), cap - n)
        out[n:n + m, 0] = _ids_as_float_lanes([r.frame_id])[0]
        out[n:n + m, 1] = refl["com_x"][:m]
        out[n:n + m, 2] = refl["com_y"][:m]
        out[n:n + m, 3] = refl["com_z"][:m]
        n += m
    out[cap].view(np.uint32)[:2] = (n, wanted)
    return out


def pack_spots_batch(results, refl_all: np.ndarray, cap: int, out: np.ndarray | None = None) -> np.ndarray:
    """Same layout as pack_spots, built from the batch-wide reflection array a Stream keeps
    (`stream.last_batch_reflections`: frame i's reflections follow frame i-1's) without a
    per-frame loop over arrays.  `out` may be a reusable (cap + 1, 4) float32 buffer."""
    if out is None:
        out = np.empty((cap + 1, 4), np.float32)
    counts = np.fromiter((len(r.reflections) for r in results), np.int64, len(results))
    ids = _ids_as_float_lanes(np.fromiter((r.frame_id for r in results), np.int64, len(results)))
    wanted = int(counts.sum())
    n = int(min(wanted, cap))
    out[:n, 0] = np.repeat(ids, counts)[:n]
    # com_x, com_y, com_z are three consecutive float32 fields of the record: one strided copy
    off = refl_all.dtype.fields["com_x"][1]
    if (refl_all.dtype.itemsize % 4 == 0 and off % 4 == 0 and refl_all.dtype.fields["com_y"][1] == off + 4
            and refl_all.dtype.fields["com_z"][1] == off + 8 and refl_all.flags.c_contiguous):
        v = refl_all.view(np.float32).reshape(-1, refl_all.dtype.itemsize // 4)
        out[:n, 1:4] = v[:n, off // 4:off // 4 + 3]
    else:
        out[:n, 1] = refl_all["com_x"][:n]
        out[:n, 2] = refl_all["com_y"][:n]
        out[:n, 3] = refl_all["com_z"][:n]
    out[cap] = 0
    out[cap].view(np.uint32)[:2] = (n, wanted)
    return out


def unpack_spots(gathered: np.ndarray, world: int, cap: int, blocks_per_rank: int = 1):
    """-> {frame_id: (n,3) float32 centres} merged over ranks, insertion in frame order.  Frame ids come
    back as the low 32 bits of what was packed.  Raises SpotGatherTruncated if any block was too small."""
    g = np.ascontiguousarray(gathered, np.float32).reshape(world * blocks_per_rank, cap + 1, 4)
    merged = {}
    for r in range(world * blocks_per_rank):
        n, wanted = (int(v) for v in g[r, cap].view(np.uint32)[:2])
        if wanted > n:
            raise SpotGatherTruncated(f"block {r}: {wanted} spots, room for {cap}")
        rows = g[r, :n]
        ids = rows[:, 0].view(np.uint32)
        for fid in np.unique(ids):
            merged[int(fid)] = rows[ids == fid, 1:4].copy()
    return dict(sorted(merged.items()))


def all_gather_fixed(t, group=None):
    """One collective: every rank contributes a tensor of identical shape; returns the
    concatenation along dim 0 (rank-major)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.cat(out, 0)


class _RowGather:
    """A gather in flight between gather_rows_begin and gather_rows_finish."""
    __slots__ = ("counts_dev", "counts_host", "event", "world", "rank", "group")


class RowGatherScratch:
    """The small tensors a GPU row gather needs, allocated ONCE (a pinned allocation and a host-to-device copy out of pageable
    memory per gather cost 1.3 ms of host time per flush -- four steps' worth -- in bench.py's one-rank rehearsal): (count, tag)
    in pinned host memory and on the device, every rank's counts on the device and in pinned host memory; `slots` sets of them,
    used in turn (a set is free again once its gather has been finished)."""

    def __init__(self, device, group=None, slots: int = 2):
        import torch
        import torch.distributed as dist
        self.world = dist.get_world_size(group)
        self.sets = [dict(me_host=torch.zeros(2, dtype=torch.int64).pin_memory(),
                          me_dev=torch.zeros(2, dtype=torch.int64, device=device),
                          counts_dev=torch.zeros(2 * self.world, dtype=torch.int64, device=device),
                          counts_host=torch.zeros(2 * self.world, dtype=torch.int64).pin_memory(),
                          event=torch.cuda.Event()) for _ in range(slots)]
        self.next = 0

    def take(self):
        s = self.sets[self.next]
        self.next = (self.next + 1) % len(self.sets)
        return s


def gather_rows_begin(n: int, tag: int, device, group=None, scratch: RowGatherScratch | None = None) -> _RowGather:
    """First half of the row gather: the all_gather of every rank's (row count, tag).  On a GPU group nothing here waits for the
    device -- the counts travel into pinned host memory behind an event that gather_rows_finish waits for -- so a caller that
    finishes gather k while beginning gather k + 1 never stalls its submit loop on a collective.  `scratch`: preallocated
    tensors (RowGatherScratch) for callers that gather often."""
    import torch
    import torch.distributed as dist
    h = _RowGather()
    h.world, h.rank, h.group = dist.get_world_size(group), dist.get_rank(group), group
    if scratch is not None:
        s = scratch.take()
        s["me_host"][0] = int(n)
        s["me_host"][1] = int(tag)
        s["me_dev"].copy_(s["me_host"], non_blocking=True)
        dist.all_gather_into_tensor(s["counts_dev"], s["me_dev"], group=group)
        s["counts_host"].copy_(s["counts_dev"], non_blocking=True)
        s["event"].record()
        h.counts_dev, h.counts_host, h.event = s["counts_dev"], s["counts_host"], s["event"]
        return h
    me = torch.tensor([int(n), int(tag)], dtype=torch.int64, device=device)
    h.counts_dev = all_gather_fixed(me, group)
    if h.counts_dev.is_cuda:
        h.counts_host = torch.empty(h.counts_dev.shape, dtype=torch.int64).pin_memory()
        h.counts_host.copy_(h.counts_dev, non_blocking=True)
        h.event = torch.cuda.Event()
        h.event.record()
    else:
        h.counts_host, h.event = h.counts_dev, None
    return h


def gather_rows_finish(h: _RowGather, rows, root: int = 0, recv_buf=None):
    """Second half: with the counts on the host, EXACTLY the written rows travel point to point to `root` (one group of sends and
    receives).  Returns (gathered, counts, requests) as gather_rows_to_root does."""
    import torch
    import torch.distributed as dist
    if h.event is not None:
        h.event.synchronize()
    counts = h.counts_host.numpy().reshape(h.world, 2).copy()
    n = int(counts[h.rank, 0])
    ops, gathered = [], None
    if h.rank == root:
        total = int(counts[:, 0].sum())
        if recv_buf is None:
            recv_buf = torch.empty((max(total, 1), 4), dtype=rows.dtype, device=rows.device)
        if recv_buf.shape[0] < total:
            raise SpotGatherTruncated(f"{total} rows gathered, room for {recv_buf.shape[0]}")
        off = 0
        for r in range(h.world):
            c = int(counts[r, 0])
            if r == h.rank:
                if c:
                    recv_buf[off:off + c].copy_(rows[:c])
            elif c:
                ops.append(dist.P2POp(dist.irecv, recv_buf[off:off + c], r, h.group))
            off += c
        gathered = recv_buf[:total]
    elif n:
        ops.append(dist.P2POp(dist.isend, rows[:n], root, h.group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    return gathered, counts, reqs


def gather_rows_to_root(rows, n: int, tag: int = 0, root: int = 0, group=None, recv_buf=None):
    """The gather `north_star` words ("a simple RCCL-over-xGMI gather of the per-frame spot lists"): one tiny
    all_gather of every rank's row count, then EXACTLY the written rows travel, point to point, to `root` -- no padding, and
    nothing lands on ranks that do not read it.  `rows`: (cap, 4) float32 tensor whose first `n` rows are valid (device
    tensor under "nccl" = RCCL: the sends and receives go out as one group; CPU tensor under "gloo").  `tag` travels with the
    count (bench.py: rank + 1).  Returns (gathered, counts, requests): on `root` `gathered` is the (sum of counts, 4) tensor
    of all ranks' rows in rank order (a view of `recv_buf` when given, which must hold them), None elsewhere; `counts` is the
    (world, 2) int64 array of (rows, tag) per rank; wait on `requests` before reading `gathered` / reusing `rows`.
    (= gather_rows_begin + gather_rows_finish back to back; bench.py keeps one gather in flight between the two.)"""
    return gather_rows_finish(gather_rows_begin(n, tag, rows.device, group), rows, root, recv_buf)


def rows_by_frame(rows: np.ndarray) -> dict:
    """{frame_id: (n, 3) float32 centres} from gathered (frame_id bits, x, y, z) rows, in frame order."""
    rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 4)
    ids = rows[:, 0].view(np.uint32)
    return {int(fid): rows[ids == fid, 1:4].copy() for fid in np.unique(ids)}


def gather_strong_lists(slices: dict, device=None, group=None) -> dict:
    """Variable-length gather of {frame_id: (k uint32, intensity uint32)} to every rank:
    all_gather of counts, then one padded all_gather of the payload."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    ids = sorted(slices)
    n_entries = sum(len(slices[i][0]) for i in ids)
    meta = torch.tensor([len(ids), n_entries], dtype=torch.int64, device=device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_ids = max(int(m[0]) for m in metas)
    max_entries = max(int(m[1]) for m in metas)
    # header: (frame_id, count) per slice; payload: k, intensity concatenated
    head = np.zeros((max(max_ids, 1), 2), np.int64)
    k = np.zeros(max(max_entries, 1), np.int64)
    it = np.zeros(max(max_entries, 1), np.int64)
    at = 0
    for j, i in enumerate(ids):
        kk, ii = slices[i]
        head[j] = (i, len(kk))
        k[at:at + len(kk)] = kk
        it[at:at + len(kk)] = ii
        at += len(kk)
    payload = torch.from_numpy(np.concatenate([head.reshape(-1), k, it]))
    if device is not None:
        payload = payload.to(device)
    outs = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(outs, payload, group=group)
    merged = {}
    hl = head.size
    for r in range(world):
        o = outs[r].cpu().numpy()
        h = o[:hl].reshape(-1, 2)
        kk = o[hl:hl + len(k)]
        ii = o[hl + len(k):]
        at = 0
        for j in range(int(metas[r][0])):
            fid, cnt = int(h[j, 0]), int(h[j, 1])
            merged[fid] = (kk[at:at + cnt].astype(np.uint32), ii[at:at + cnt].astype(np.uint32))
            at += cnt
    return dict(sorted(merged.items()))
