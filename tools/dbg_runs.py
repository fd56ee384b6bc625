#!/usr/bin/env python3
"""Debug aid for the run-based sparse stage: which frames / boxes differ from the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-feedback-service_amd", "python")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffs_amd
from util import oracle_frame, make_frame
src = open(os.path.join(ROOT, "tests", "test_gpu_edge_paths.py")).read()
ns = {}
exec("import numpy as np\n" + src[src.index("def _blob_frame"):src.index("@pytest.mark.parametrize(\"chain_runs\"")], ns)
_blob_frame = ns["_blob_frame"]
W, H = 1000, 700
a = _blob_frame(W, H, 1, 500)
b = _blob_frame(W, H, 2, 420)
b[300, :] = 900; b[301, 0:40] = 900; b[400:440, 31:33] = 700; b[500, 64:96] = 1234; b[502, 63:97] = 1234
b[0, 0:5] = 800; b[H - 1, W - 5:W] = 800; b[0, W - 3:W] = 800; b[1, 0:3] = 800
c = _blob_frame(W, H, 3, 480); c[c > 150] = 2000
sparse, _ = make_frame(W=W, H=H, seed=33, n_spots=30)
frames = np.stack([a, b, sparse, np.zeros((H, W), np.uint16), c])
ones = np.ones((H, W), np.uint8)
ctx = ffs_amd.Context(W, H, np.uint16, max_batch=5, max_strong_per_frame=90000)
ctx.set_params(want_strong_mask=1, want_strong_list=1, min_spot_size=1, max_peak_centroid_separation=3.0)
st = ctx.stream()
for rep in range(3):
    res = st.process(frames)
    for f, (fr, img) in enumerate(zip(res, frames)):
        strong, cc, refl = oracle_frame(img, ones, 1, 3.0, None)
        print(f"rep {rep} frame {f}: strong {fr.num_strong_pixels}/{cc.num_strong_pixels} comps {fr.n_components}/{cc.n_unfiltered_boxes} "
              f"filtered px {fr.num_strong_pixels_filtered}/{cc.num_strong_pixels_filtered} boxes {len(fr.boxes)}/{len(cc.boxes)}")
        n = min(len(fr.boxes), len(cc.boxes))
        bad = [i for i in range(n) if any(fr.boxes[k][i] != cc.boxes[k][i] for k in ("l", "t", "r", "b", "num_pixels"))]
        print("   differing boxes:", len(bad), [(i, tuple(int(fr.boxes[k][i]) for k in ("l", "t", "r", "b", "num_pixels")), tuple(int(cc.boxes[k][i]) for k in ("l", "t", "r", "b", "num_pixels"))) for i in bad[:6]])
mask = (np.random.default_rng(9).random((H, W)) > 0.002).astype(np.uint8)
mask[:, 500:504] = 0
ctx.set_mask(mask)
ctx.set_params(want_strong_list=1)
for rep in range(2):
    res = st.process(frames)
    for f, (fr, img) in enumerate(zip(res, frames)):
        strong, cc, refl = oracle_frame(img, mask, 3, 2.0, None)
        print(f"masked rep {rep} frame {f}: strong {fr.num_strong_pixels}/{cc.num_strong_pixels} comps {fr.n_components}/{cc.n_unfiltered_boxes} "
              f"filtered px {fr.num_strong_pixels_filtered}/{cc.num_strong_pixels_filtered} boxes {len(fr.boxes)}/{len(cc.boxes)}")
        n = min(len(fr.boxes), len(cc.boxes))
        bad = [i for i in range(n) if any(fr.boxes[k][i] != cc.boxes[k][i] for k in ("l", "t", "r", "b", "num_pixels"))]
        print("   differing boxes:", len(bad), [(i, tuple(int(fr.boxes[k][i]) for k in ("l", "t", "r", "b", "num_pixels")), tuple(int(cc.boxes[k][i]) for k in ("l", "t", "r", "b", "num_pixels"))) for i in bad[:6]])
        same_k = np.array_equal(fr.strong_k.astype(np.uint64), cc.k)
        print("   list equal:", same_k)
