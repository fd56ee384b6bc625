"""The RCCL leg of the multi-GPU path on ONE GPU: a 1-rank `nccl` process group gathers spot centres
that the HIP path produced (ffs_stream_spot_centres -> device tensor -> all_gather_into_tensor), exactly the
calls bench.py makes with N > 1 ranks, and the strong-pixel lists of a small sweep (dist.gather_strong_lists)
that feed the 3D stack.  World size > 1 is covered on CPU by tests/test_distributed_gloo.py."""
import socket

import numpy as np
import pytest

from util import make_frame

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_nccl_gather_of_hip_spots(ffs):
    import torch
    import torch.distributed as dist
    from ffs_amd import dist as D
    from oracle import oracle as O
    W, H, B = 517, 389, 6
    frames, mask = [], None
    for i in range(B):
        img, mask = make_frame(W=W, H=H, seed=100 + i, n_spots=30)
        frames.append(img)
    ctx = ffs.Context(W, H, np.uint16, max_batch=B)
    ctx.set_mask(mask)
    ctx.set_params(want_reflections=1, want_strong_list=1)
    st = ctx.stream()
    first_id = (1 << 24) + 5            # ids beyond float32's integer range must survive the float lanes
    res = st.process(np.stack(frames), first_frame_id=first_id)

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        cap = 64 * B
        host = torch.empty((cap + 1, 4), dtype=torch.float32).pin_memory()
        n = st.pack_spot_centres(host.numpy(), cap)
        assert n == sum(len(r.reflections) for r in res) > 20
        block = host.to(dev, non_blocking=True)
        out = torch.empty((1, cap + 1, 4), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out.view(-1), block.view(-1))
        spots = D.unpack_spots(out.cpu().numpy(), 1, cap)
        for i, (r, img) in enumerate(zip(res, frames)):
            cc = O.cc2d(O.dispersion(img, mask), img, 3)
            refl = O.cc2d_reflections(cc.k, cc.intensity, W, H, 3, 2.0).reflections
            if len(refl) == 0:
                assert first_id + i not in spots
                continue
            want = np.stack([refl["com_x"], refl["com_y"], refl["com_z"]], 1)
            np.testing.assert_array_equal(spots[first_id + i], want)      # HIP -> RCCL -> here == oracle
        # the default of bench.py --gpus N: counts, then exactly the written rows to rank 0 (gather_rows_to_root; with one rank the
        # root's own rows are copied on the device -- the count exchange is the RCCL call that runs here)
        rows_dev = host[:n].to(dev, non_blocking=True) if n else torch.empty((1, 4), dtype=torch.float32, device=dev)
        recv = torch.empty((n + 2, 4), dtype=torch.float32, device=dev)
        got, counts, reqs = D.gather_rows_to_root(rows_dev, n, tag=1, root=0, recv_buf=recv)
        for q in reqs:
            q.wait()
        assert counts.tolist() == [[n, 1]] and got.shape == (n, 4)
        by_frame = D.rows_by_frame(got.cpu().numpy())
        assert list(by_frame) == list(spots)
        for fid in spots:
            np.testing.assert_array_equal(by_frame[fid], spots[fid])
        # too small a block is reported, not dropped silently
        with pytest.raises(ffs.FfsError):
            st.pack_spot_centres(np.empty((9, 4), np.float32), 8)
        # the rotation-sweep exchange: per-frame strong lists to the rank that owns the 3D stack
        slices = {first_id + i: (r.strong_k.copy(), r.strong_intensity.copy()) for i, r in enumerate(res)}
        merged = D.gather_strong_lists(slices, device=dev)
        assert list(merged) == sorted(slices)
        for fid, (k, inten) in slices.items():
            np.testing.assert_array_equal(merged[fid][0], k)
            np.testing.assert_array_equal(merged[fid][1], inten)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", [None, "rccl"])
def test_native_gather_of_spot_rows(ffs, transport):
    """ffs_multi_gather_rows: the library's own RCCL gather of 2D spot rows (counts by ncclAllGather, rows by ncclSend / ncclRecv to the
    root device, one copy to the host) with two contexts on this one GPU -- a one-rank communicator.  transport None: the streams
    of the root's GPU hand their rows over by device copies (the count exchange is the collective that runs); "rccl": forced, every
    stream sends to its own rank.  An empty batch on one side, row order = stream order, too small a receiver reported."""
    from ffs_amd import api
    W, H, B = 517, 389, 4
    imgs, mask = [], None
    for i in range(2 * B):
        img, mask = make_frame(W=W, H=H, seed=300 + i, n_spots=25)
        imgs.append(img)
    got_t = api.multi_init([0, 0], transport)
    if got_t != "rccl":
        pytest.skip("no RCCL on this machine")
    ctxs = [ffs.Context(W, H, np.uint16, max_batch=B) for _ in range(2)]
    streams = []
    for c in ctxs:
        c.set_mask(mask)
        c.set_params(want_reflections=1)
        streams.append(c.stream())
    res = [streams[0].process(np.stack(imgs[:B]), first_frame_id=(1 << 24) + 1), streams[1].process(np.stack(imgs[B:]), first_frame_id=77)]
    want = []
    for s in streams:
        host = np.empty((4096 + 1, 4), np.float32)
        n = s.pack_spot_centres(host, 4096)
        want.append(host[:n].copy())
    assert len(want[0]) > 20 and len(want[1]) > 20
    for root in (0, 1):
        rows = api.multi_gather_rows(streams, root=root, cap=8192)
        np.testing.assert_array_equal(rows.view(np.uint32), np.concatenate(want).view(np.uint32))
    rows = api.multi_gather_rows(streams[::-1], root=0, cap=8192)          # stream order is row order
    np.testing.assert_array_equal(rows.view(np.uint32), np.concatenate(want[::-1]).view(np.uint32))
    with pytest.raises(ffs.FfsError):
        api.multi_gather_rows(streams, root=0, cap=len(want[0]) + 3)
    # one side with nothing to send
    streams[1].process(np.zeros((B, H, W), np.uint16), first_frame_id=500)
    rows = api.multi_gather_rows(streams, root=1, cap=8192)
    np.testing.assert_array_equal(rows.view(np.uint32), want[0].view(np.uint32))
    # while a batch is in flight the stream is refused
    streams[0].submit(np.stack(imgs[:B]), first_frame_id=0)
    with pytest.raises(ffs.FfsError):
        api.multi_gather_rows(streams, root=0, cap=8192)
    streams[0].wait()
