"""Is the device BIASED on the dense sphere cluster of the book-2 final scene (1000 r = 10 spheres under RotateY + Translate), or do its
paths merely diverge from the f64 oracle's (a cluster of convex mirrors-of-normals amplifies an f32 rounding by ~dist/r per bounce)?
Several seeds; per seed segments/sample and mean radiance of device and oracle; then the mean and standard error of the differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import rta
from oracle import binding as orc
p = rta.load(); A = p._abi
ctx = p.Context(0)
W = H = 96; SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 64; SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
def scene(xf):
    rng = np.random.default_rng(5)
    b = p.SceneBuilder(background=(0.6, 0.6, 0.6))
    white = b.lambertian((0.73, 0.73, 0.73))
    sp = [b.sphere(tuple(rng.uniform(0, 165, 3)), 10, white) for _ in range(1000)]
    node = b.bvh(sp, 0, 1)
    if xf: node = b.translate(b.rotate_y(node, 15), (-100, 270, 395))
    return b, b.desc(b.hittable_list([node]))
for xf in (True, False):
    b, desc = scene(xf)
    cam = p.camera_new((478, 278, -600), (-20 if xf else 82, 350 if xf else 82, 480 if xf else 82), (0, 1, 0), 22, 1.0, 0.0, 10.0, 0, 1) if xf else \
          p.camera_new((600, 300, -600), (82, 82, 82), (0, 1, 0), 22, 1.0, 0.0, 10.0, 0, 1)
    sc = ctx.upload(desc)
    dseg, dmean, segs = [], [], []
    for seed in range(1, SEEDS + 1):
        prm = p.make_params(W, H, SPP, max_depth=50, seed=seed)
        img, st = ctx.render(sc, cam, prm)
        ref, ost = orc.render(desc, cam, prm, precision=64, n_threads=16, count=True)
        dseg.append(st["segments"] / ost["segments"] - 1); dmean.append(img.mean() / ref.mean() - 1); segs.append(ost["segments"] / ost["samples"])
        d = np.abs(img - ref) / SPP
        print(f"xform {xf} seed {seed} seg/sample orc {segs[-1]:.4f} dseg {dseg[-1]:+.5f} dmean {dmean[-1]:+.5f} mean|d| {d.mean():.2e} frac pixels differing {float((d.max(axis=2) > 1e-6).mean()):.3f}", flush=True)
    dseg, dmean = np.array(dseg), np.array(dmean)
    print(f"xform {xf}: dseg mean {dseg.mean():+.5f} +- {dseg.std(ddof=1)/np.sqrt(len(dseg)):.5f}   dmean {dmean.mean():+.5f} +- {dmean.std(ddof=1)/np.sqrt(len(dmean)):.5f}", flush=True)
