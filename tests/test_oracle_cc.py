"""CPU: the connected-components restatement.  The reference's arithmetic for this stage sits on
Boost.Graph (not under /root/reference) and its numeric pins need external datasets, so this
stage is pinned by hand-checkable known answers and scipy.ndimage.label property tests
(SURVEY.md section 8c)."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import oracle as O


def _frame(H, W, pts):
    img = np.zeros((H, W), np.uint16)
    res = np.zeros((H, W), np.uint8)
    for (y, x, v) in pts:
        img[y, x] = v
        res[y, x] = 1
    return res, img


def test_single_pixel():
    res, img = _frame(10, 12, [(4, 7, 33)])
    cc = O.cc2d(res, img, 1)
    assert cc.num_strong_pixels == 1 and len(cc.boxes) == 1
    b = cc.boxes[0]
    assert (b["l"], b["t"], b["r"], b["b"], b["num_pixels"]) == (7, 4, 7, 4, 1)
    r = O.cc2d_reflections(cc.k, cc.intensity, 12, 10, 1, 2.0).reflections[0]
    assert (r["com_x"], r["com_y"], r["com_z"]) == (7.5, 4.5, 0.5)       # pixel centres, hpp:86-88
    assert r["peak_centroid_distance"] == 0.0


def test_l_shape_and_min_size_filter():
    res, img = _frame(10, 10, [(2, 2, 10), (3, 2, 10), (4, 2, 10), (4, 3, 30), (8, 8, 5)])
    cc = O.cc2d(res, img, 3)
    assert cc.n_unfiltered_boxes == 2 and len(cc.boxes) == 1
    assert cc.num_strong_pixels == 5 and cc.num_strong_pixels_filtered == 4
    b = cc.boxes[0]
    assert (b["l"], b["t"], b["r"], b["b"], b["num_pixels"]) == (2, 2, 3, 4, 4)
    rf = O.cc2d_reflections(cc.k, cc.intensity, 10, 10, 3, 2.0)
    assert rf.n_calculated == 2 and rf.n_filtered_size == 1
    r = rf.reflections[0]
    # COM: x = (2.5*10*3 + 3.5*30)/60, y = (2.5*10 + 3.5*10 + 4.5*40)/60
    assert r["com_x"] == np.float32((2.5 * 30 + 3.5 * 30) / 60)
    assert r["com_y"] == np.float32((2.5 * 10 + 3.5 * 10 + 4.5 * 40) / 60)
    assert (r["peak_x"], r["peak_y"], r["peak_intensity"]) == (3, 4, 30)


def test_diagonal_pixels_are_two_spots():
    res, img = _frame(8, 8, [(2, 2, 9), (3, 3, 9)])
    assert O.cc2d(res, img, 1).n_unfiltered_boxes == 2        # 4-connectivity only


def test_row_wrap_quirk():
    """k+1 is linked with no row-end check (connected_components.cc:62-70)."""
    W = 16
    res, img = _frame(6, W, [(2, W - 1, 7), (3, 0, 7)])
    cc = O.cc2d(res, img, 1)
    assert cc.n_unfiltered_boxes == 1
    b = cc.boxes[0]
    assert (b["l"], b["r"], b["t"], b["b"]) == (0, W - 1, 2, 3)


def test_label_order_is_min_linear_index():
    # component A has the smaller minimum index although most of it lies lower
    res, img = _frame(12, 12, [(1, 10, 5), (2, 10, 5), (3, 10, 5), (2, 2, 5), (2, 3, 5)])
    cc = O.cc2d(res, img, 1)
    assert [int(b["l"]) for b in cc.boxes] == [10, 2]


def test_peak_tie_prefers_smallest_zyx():
    res, img = _frame(8, 8, [(3, 3, 50), (3, 4, 50), (4, 3, 50)])
    cc = O.cc2d(res, img, 1)
    r = O.cc2d_reflections(cc.k, cc.intensity, 8, 8, 1, 10.0).reflections[0]
    assert (r["peak_x"], r["peak_y"]) == (3, 3)


def test_peak_centroid_filter():
    # one very bright pixel far from the bulk of a long streak -> distance > 2
    pts = [(5, x, 10) for x in range(2, 14)] + [(5, 14, 200)]
    res, img = _frame(12, 20, pts)
    cc = O.cc2d(res, img, 3)
    rf = O.cc2d_reflections(cc.k, cc.intensity, 20, 12, 3, 2.0)
    assert rf.n_calculated == 1 and rf.n_filtered_sep == 1 and len(rf.reflections) == 0
    keep = O.cc2d_reflections(cc.k, cc.intensity, 20, 12, 3, 0.0)   # 0 disables the filter (cc:224)
    assert len(keep.reflections) == 1


def test_3d_blob_persists_over_z():
    W, H = 20, 20
    mk = lambda pts: (np.array(sorted(y * W + x for y, x in pts), np.uint64), np.full(len(pts), 10, np.uint32))
    s0 = mk([(5, 5), (5, 6)])
    s1 = mk([(5, 6), (9, 9)])
    s2 = mk([(9, 9), (5, 6)])
    s3 = mk([(15, 15)])
    r = O.cc3d([s0, s1, s2, s3], W, H, 1, 0.0)
    assert r.n_calculated == 3
    a, b, c = r.reflections
    assert (a["z_min"], a["z_max"], a["num_pixels"]) == (0, 2, 4)      # (5,5),(5,6) chain through z
    assert (b["z_min"], b["z_max"], b["num_pixels"]) == (1, 2, 2)      # (9,9) in slices 1,2
    assert (c["z_min"], c["z_max"], c["num_pixels"]) == (3, 3, 1)
    assert a["com_z"] == np.float32((0.5 * 20 + 1.5 * 10 + 2.5 * 10) / 40)


@pytest.mark.parametrize("seed", range(8))
def test_partition_matches_scipy_label_2d(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(5, 60)), int(rng.integers(5, 60))
    res = (rng.random((H, W)) < rng.uniform(0.05, 0.5)).astype(np.uint8)
    res[:, -1] = 0                     # keep the row-wrap quirk out of the comparison
    img = rng.integers(1, 1000, (H, W)).astype(np.uint16)
    cc = O.cc2d(res, img, 1)
    lab, n = ndimage.label(res, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    assert cc.n_unfiltered_boxes == n
    objs = ndimage.find_objects(lab)
    # scipy labels in raster order of first pixel = ascending minimum linear index
    for b, sl, i in zip(cc.boxes, objs, range(1, n + 1)):
        assert (b["t"], b["b"] + 1, b["l"], b["r"] + 1) == (sl[0].start, sl[0].stop, sl[1].start, sl[1].stop)
        assert b["num_pixels"] == int((lab == i).sum())
    rf = O.cc2d_reflections(cc.k, cc.intensity, W, H, 1, 0.0).reflections
    coms = ndimage.center_of_mass(img.astype(np.float64), lab, range(1, n + 1))
    for r, (cy, cx) in zip(rf, coms):
        assert abs(r["com_x"] - (cx + 0.5)) < 1e-4 and abs(r["com_y"] - (cy + 0.5)) < 1e-4


@pytest.mark.parametrize("seed", range(4))
def test_partition_matches_scipy_label_3d(seed):
    rng = np.random.default_rng(100 + seed)
    Z, H, W = 6, 20, 24
    vol = (rng.random((Z, H, W)) < 0.15).astype(np.uint8)
    vol[:, :, -1] = 0
    slices = []
    for z in range(Z):
        k = np.flatnonzero(vol[z]).astype(np.uint64)
        slices.append((k, np.ones(len(k), np.uint32)))
    r = O.cc3d(slices, W, H, 1, 0.0)
    st = ndimage.generate_binary_structure(3, 1)       # 6-connectivity
    lab, n = ndimage.label(vol, structure=st)
    assert r.n_calculated == n
    sizes = sorted(int((lab == i).sum()) for i in range(1, n + 1))
    assert sorted(int(x) for x in r.reflections["num_pixels"]) == sizes


def _helix_components(vol):
    """Independent formulation of the reference's graph INCLUDING its row-wrap edge: strong pixels of each slice
    as a 1D sequence (linear index k = y W + x), edges k -- k+1 (no row-end check, connected_components.cc:62-70),
    k -- k+W, and the same k in the next slice (:352-370); components by scipy's sparse-graph labelling; numbered
    in order of their smallest (z, k) vertex (Boost's DFS discovery order over ascending vertex ids)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    Z, H, W = vol.shape
    flat = vol.reshape(Z, H * W) != 0
    ids = -np.ones((Z, H * W), np.int64)
    n = 0
    for z in range(Z):
        on = np.flatnonzero(flat[z])
        ids[z, on] = np.arange(n, n + len(on))
        n += len(on)
    src, dst = [], []
    for z in range(Z):
        a = ids[z]
        for step in (1, W):                        # k+1 joins (W-1, y) with (0, y+1) too
            both = (a[:-step] >= 0) & (a[step:] >= 0)
            src.append(a[:-step][both]); dst.append(a[step:][both])
        if z + 1 < Z:
            both = (a >= 0) & (ids[z + 1] >= 0)
            src.append(a[both]); dst.append(ids[z + 1][both])
    src, dst = np.concatenate(src), np.concatenate(dst)
    g = coo_matrix((np.ones(len(src), np.int8), (src, dst)), shape=(n, n))
    _, lab = connected_components(g, directed=False)
    first = np.full(lab.max() + 1 if n else 0, n, np.int64)
    np.minimum.at(first, lab, np.arange(n))
    order = np.argsort(first)                      # label order = order of the smallest vertex
    rank = np.empty_like(order); rank[order] = np.arange(len(order))
    return ids, rank[lab] if n else lab


@pytest.mark.parametrize("seed", range(10))
def test_row_wrap_edge_and_label_order_against_a_graph_formulation_2d(seed):
    """The row-wrap quirk and the label order, which the scipy.ndimage comparison above has to leave out, against
    an independent sparse-graph formulation."""
    rng = np.random.default_rng(200 + seed)
    H, W = int(rng.integers(4, 40)), int(rng.integers(3, 40))
    res = (rng.random((H, W)) < rng.uniform(0.2, 0.6)).astype(np.uint8)
    res[:, -1] |= (rng.random(H) < 0.7)            # plenty of pixels in the last column ...
    res[:, 0] |= (rng.random(H) < 0.7)             # ... and the first: many wrap edges
    img = rng.integers(1, 500, (H, W)).astype(np.uint16)
    cc = O.cc2d(res, img, 1)
    ids, lab = _helix_components(res[None])
    n = int(lab.max()) + 1 if len(lab) else 0
    assert cc.n_unfiltered_boxes == n == len(cc.boxes)
    k_on = np.flatnonzero(res.reshape(-1))
    ys, xs = np.divmod(k_on, W)
    wrapped = 0
    for i, b in enumerate(cc.boxes):               # boxes come in label order
        sel = lab == i
        assert b["num_pixels"] == int(sel.sum())
        assert (b["l"], b["r"], b["t"], b["b"]) == (xs[sel].min(), xs[sel].max(), ys[sel].min(), ys[sel].max())
        wrapped += int(b["l"] == 0 and b["r"] == W - 1)
    # the same frame WITHOUT the wrap edges has more components whenever a wrap edge joins two of them
    lab4, n4 = ndimage.label(res, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    assert n <= n4
    if seed == 0:
        assert n < n4 and wrapped > 0


@pytest.mark.parametrize("seed", range(6))
def test_row_wrap_edge_and_label_order_against_a_graph_formulation_3d(seed):
    rng = np.random.default_rng(300 + seed)
    Z, H, W = int(rng.integers(2, 7)), int(rng.integers(4, 20)), int(rng.integers(3, 20))
    vol = (rng.random((Z, H, W)) < 0.3).astype(np.uint8)
    vol[:, :, -1] |= (rng.random((Z, H)) < 0.5)
    vol[:, :, 0] |= (rng.random((Z, H)) < 0.5)
    slices = []
    for z in range(Z):
        k = np.flatnonzero(vol[z]).astype(np.uint64)
        slices.append((k, (1 + (k % 7)).astype(np.uint32)))
    r = O.cc3d(slices, W, H, 1, 0.0)
    ids, lab = _helix_components(vol)
    n = int(lab.max()) + 1
    assert r.n_calculated == n == len(r.reflections)
    zz = np.concatenate([np.full(len(k), z) for z, (k, _) in enumerate(slices)])
    kk = np.concatenate([k for k, _ in slices]).astype(np.int64)
    inten = np.concatenate([i for _, i in slices]).astype(np.float64)
    for i, rf in enumerate(r.reflections):          # reflections come in label order
        sel = lab == i
        assert rf["num_pixels"] == int(sel.sum())
        assert (rf["z_min"], rf["z_max"]) == (zz[sel].min(), zz[sel].max())
        assert (rf["x_min"], rf["x_max"]) == ((kk[sel] % W).min(), (kk[sel] % W).max())
        w = inten[sel]
        assert abs(rf["com_z"] - ((zz[sel] + 0.5) * w).sum() / w.sum()) < 1e-4
    assert np.array_equal(O.cc3d_signals(slices, W, H, 1, 0.0), lab)   # every signal's reflection
