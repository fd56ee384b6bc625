import sys, os
ROOT='/root/repo' if os.path.exists('/root/repo/tests') else os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, ROOT+'/fast-feedback-service_amd/python'); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import numpy as np, ffs_amd as ffs
from oracle import oracle as O
rng = np.random.default_rng(hash("single_photons") % 1000)
H, W = 300, 1300
img = (rng.random((H, W)) < 0.03).astype(np.uint16)
mask = (rng.random((H, W)) > 0.01).astype(np.uint8)
ctx = ffs.Context(W, H, np.uint16, max_strong_per_frame=W*H)
ctx.set_mask(mask); ctx.set_params(want_strong_mask=1, want_strong_list=1)
fr = ctx.stream().process(img)[0]
want = O.dispersion(img, mask)
d = np.argwhere(fr.strong_mask != want)
print(len(d), "mismatches; oracle strong", int(want.sum()), "gpu", int(fr.strong_mask.sum()))
for y,x in d[:12]:
    w = (img*mask)[max(y-3,0):y+4, max(x-3,0):x+4]
    m = mask[max(y-3,0):y+4, max(x-3,0):x+4]
    # the 8th pixel the gather would load
    j = x % 8; jm3 = j-3; bx = x-3 if jm3%2==0 else x-4
    extra = bx+7 if jm3%2==0 else bx
    col = img[max(y-3,0):y+4, extra] if 0<=extra<W else None
    mcol = mask[max(y-3,0):y+4, extra] if 0<=extra<W else None
    print((y,x), "j",j, "m",int(m.sum()), "x",int(w.sum()), "gpu",fr.strong_mask[y,x], "want",want[y,x], "extra col", extra, None if col is None else col.tolist(), None if mcol is None else mcol.tolist(), "raw window sum", int(img[max(y-3,0):y+4, max(x-3,0):x+4].sum()))
