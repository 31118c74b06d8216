"""Does RtUploadOptions.leaf_collapse change a frame? (it must not) usage: python3 scripts/gpu_collapse_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rta
p = rta.load(); A = p._abi
hs = p.HostScene("book1", 1)
ctx = p.Context(0)
cam = hs.camera(1.5)
W, H, SPP = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (300, 200, 16)
prm = p.make_params(W, H, SPP, seed=3)
for env in ({},):
    for k in ("RT_FIRST_IN_SHADE", "RT_BIG_SPHERES_FIRST"):
        os.environ.pop(k, None)
    os.environ.update(env)
    base, _ = ctx.render(ctx.upload(hs.desc), cam, prm)
    for lc in (2, 4):
        img, st = ctx.render(ctx.upload(hs.desc, leaf_collapse=lc), cam, prm)
        d = np.abs(img - base).max(axis=2)
        print(env, "leaf_collapse", lc, "pixels that differ:", int((d > 0).sum()), "of", W * H, "max diff of a pixel sum", float(d.max()), "segments", st["segments"])
