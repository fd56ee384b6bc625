"""CPU, world_size 2, gloo: the N>1 path -- frame sharding and the spot-list gathers that
bench.py runs over RCCL.  Strong pixels here come from the oracle (this is a test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Fr:
    def __init__(self, fid, refl):
        self.frame_id, self.reflections = fid, refl


def _sweep():
    from ffs_amd import synth
    W, H, NZ = 160, 120, 8
    p = synth.sweep_params(seed=31, n_frames=NZ, n_spots=25, width=W, height=H)
    return W, H, NZ, synth.frames(p, range(NZ), threads=2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffs_amd import dist as D
        from oracle import oracle as O
        W, H, NZ, frames = _sweep()
        mask = np.ones((H, W), np.uint8)
        mine = D.frame_shard(NZ, rank, world)
        assert mine == list(range(rank, NZ, world))
        results, slices = [], {}
        for f in mine:
            strong = O.dispersion(frames[f], mask)
            cc = O.cc2d(strong, frames[f], 3)
            refl = O.cc2d_reflections(cc.k, cc.intensity, W, H, 3, 2.0).reflections
            results.append(_Fr(f, refl))
            slices[f] = (cc.k.astype(np.uint32), cc.intensity)
        cap = 256
        packed = torch.from_numpy(D.pack_spots(results, cap))
        gathered = D.all_gather_fixed(packed).numpy()
        spots = D.unpack_spots(gathered, world, cap)
        merged = D.gather_strong_lists(slices)
        if rank == 0:
            q.put((spots, merged))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    from oracle import oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    spots, merged = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    W, H, NZ, frames = _sweep()
    mask = np.ones((H, W), np.uint8)
    assert list(spots) == [f for f in range(NZ) if f in spots] and list(merged) == list(range(NZ))
    single = []
    n_spots = 0
    for f in range(NZ):
        strong = O.dispersion(frames[f], mask)
        cc = O.cc2d(strong, frames[f], 3)
        np.testing.assert_array_equal(merged[f][0], cc.k.astype(np.uint32))
        np.testing.assert_array_equal(merged[f][1], cc.intensity)
        refl = O.cc2d_reflections(cc.k, cc.intensity, W, H, 3, 2.0).reflections
        if len(refl):
            want = np.stack([refl["com_x"], refl["com_y"], refl["com_z"]], 1)
            np.testing.assert_array_equal(spots[f], want)
            n_spots += len(refl)
        single.append((cc.k, cc.intensity))
    assert n_spots > 10
    # the gathered lists drive the same 3D labelling as a single process would
    a = O.cc3d([merged[f] for f in range(NZ)], W, H, 3, 2.0)
    b = O.cc3d(single, W, H, 3, 2.0)
    assert a.n_calculated == b.n_calculated and len(a.reflections) == len(b.reflections) > 0
    np.testing.assert_array_equal(a.reflections, b.reflections)


def test_frame_shard_partitions():
    from ffs_amd import dist as D
    for world in (1, 2, 3, 8):
        got = sorted(sum((D.frame_shard(37, r, world) for r in range(world)), []))
        assert got == list(range(37))


def test_pack_spots_batch_equals_pack_spots():
    from ffs_amd import dist as D
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    results, parts = [], []
    for fid in (5, 6, 9):
        n = int(rng.integers(0, 7))
        refl = np.zeros(n, O.REFL_DT)
        refl["com_x"], refl["com_y"], refl["com_z"] = rng.random(n), rng.random(n), 0.5
        results.append(_Fr(fid, refl))
        parts.append(refl)
    allr = np.concatenate(parts)
    a, b = D.pack_spots(results, 32), D.pack_spots_batch(results, allr, 32)
    n = int(a[32].view(np.uint32)[0])
    assert n == len(allr) == int(b[32].view(np.uint32)[0]) == int(b[32].view(np.uint32)[1])
    np.testing.assert_array_equal(a[:n], b[:n])      # rows beyond the count are don't-care


def test_spot_rows_keep_large_frame_ids_and_flag_truncation():
    from ffs_amd import dist as D
    from oracle import oracle as O
    refl = np.zeros(3, O.REFL_DT)
    refl["com_x"], refl["com_y"], refl["com_z"] = (1.5, 2.5, 3.5), (4.5, 5.5, 6.5), 0.5
    ids = [(1 << 24) + 1, (1 << 24) + 2, (1 << 31) + 7]     # collide or lose bits as float32 values
    results = [_Fr(i, refl) for i in ids]
    got = D.unpack_spots(D.pack_spots(results, 16), 1, 16)
    assert list(got) == ids and all(len(v) == 3 for v in got.values())
    allr = np.concatenate([refl] * 3)
    got = D.unpack_spots(D.pack_spots_batch(results, allr, 16), 1, 16)
    assert list(got) == ids
    with pytest.raises(D.SpotGatherTruncated):
        D.unpack_spots(D.pack_spots(results, 8), 1, 8)


def _rows_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffs_amd import dist as D
        cap = 40
        counts = [7, 0, 19]                      # unequal, and rank 1 has nothing
        rows = np.zeros((cap, 4), np.float32)
        n = counts[rank]
        ids = np.repeat(np.arange(rank * 100, rank * 100 + 4), 5)[:n] + (1 << 24)      # ids past 2^24: bit patterns, not values
        rows[:n, 0] = D._ids_as_float_lanes(ids)
        rows[:n, 1] = np.arange(n) + 0.25 * rank
        rows[:n, 2] = rank
        rows[:n, 3] = 0.5
        for rep in range(2):                     # twice: the receive buffer is reused
            recv = torch.full((sum(counts) + 3, 4), -1.0) if rank == 0 else None
            got, cts, reqs = D.gather_rows_to_root(torch.from_numpy(rows), n, tag=rank + 1, root=0, recv_buf=recv)
            for r in reqs:
                r.wait()
            assert cts[:, 0].tolist() == counts and cts[:, 1].tolist() == [1, 2, 3]
            assert (got is None) == (rank != 0)
        if rank == 0:
            q.put(got.numpy().copy())
        # a receive buffer that is too small is an error on the root, not a silent cut -- raised after the counts have
        # been exchanged (so no rank is left waiting in a collective): here every rank sends nothing
        if rank == 0:
            with pytest.raises(D.SpotGatherTruncated):
                D.gather_rows_to_root(torch.from_numpy(rows), 5, root=0, recv_buf=torch.empty((2, 4)))
        else:
            D.gather_rows_to_root(torch.from_numpy(rows), 0, root=0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_three_rank_row_gather_with_unequal_counts_and_an_empty_rank():
    """north_star's gather: all_gather of the ranks' row counts, then exactly the written rows point to point to
    rank 0 (`ffs_amd.dist.gather_rows_to_root`, what bench.py --gpus N runs over RCCL by default)."""
    from ffs_amd import dist as D
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got.shape == (26, 4)
    by_frame = D.rows_by_frame(got)
    assert list(by_frame) == [(1 << 24) + 0, (1 << 24) + 1, (1 << 24) + 200, (1 << 24) + 201, (1 << 24) + 202, (1 << 24) + 203]
    assert [len(v) for v in by_frame.values()] == [5, 2, 5, 5, 5, 4]
    np.testing.assert_array_equal(got[:7, 1], np.arange(7, dtype=np.float32))            # rank 0's rows first, in order
    np.testing.assert_array_equal(got[7:, 1], np.arange(19, dtype=np.float32) + 0.5)     # then rank 2's
    assert (got[7:, 2] == 2).all()
