"""Which object of the book-2 final scene makes the device trace 0.27 % more segments than the oracle in the glass_and_fog crop?
Variants of the scene's objects under a constant grey background, device against the f64 oracle: segments per sample and mean."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import rta
from oracle import binding as orc
p = rta.load(); A = p._abi
ctx = p.Context(0)
W = H = 192; SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cam = p.camera_new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
def variant(which):
    b = p.SceneBuilder(background=(0.6, 0.6, 0.6))
    ids = [b.xz_rect(-1000, 1000, -1000, 1000, 0, b.lambertian((0.48, 0.83, 0.53)))]
    if "glass" in which: ids.append(b.sphere((260, 150, 45), 50, b.dielectric(1.5)))
    if "sub" in which or "subglass" in which:
        bd = b.sphere((360, 150, 145), 70, b.dielectric(1.5))
        if "subglass" in which or "sub" in which and "nomedium" not in which: pass
        ids.append(bd)
        if "nomedium" not in which: ids.append(b.constant_medium(bd, 0.2, (0.2, 0.4, 0.9)))
    if "medonly" in which:
        bd = b.sphere((360, 150, 145), 70, b.dielectric(1.5))
        ids.append(b.constant_medium(bd, 0.2, (0.2, 0.4, 0.9)))
    if "fog" in which:
        ids.append(b.constant_medium(b.sphere((0, 0, 0), 5000, b.dielectric(1.5)), 0.0001, (1, 1, 1)))
    if "metal" in which: ids.append(b.sphere((0, 150, 145), 50, b.metal((0.8, 0.8, 0.9), 1.0)))
    return b, b.desc(b.hittable_list(ids))
for which in ["glass", "sub", "sub_nomedium", "medonly", "fog", "metal"]:
    b, desc = variant(which)
    prm = p.make_params(W, H, SPP, max_depth=50, seed=3)
    img, st = ctx.render(ctx.upload(desc), cam, prm)
    ref, ost = orc.render(desc, cam, prm, precision=64, n_threads=16, count=True)
    d = np.abs(img - ref) / SPP
    print(f"{which:14s} seg/sample gpu {st['segments']/st['samples']:.5f} orc {ost['segments']/ost['samples']:.5f} ({st['segments']/ost['segments']-1:+.5f})  mean gpu {img.mean()/SPP:.6f} orc {ref.mean()/SPP:.6f} ({img.mean()/ref.mean()-1:+.5f})  mean|d| {d.mean():.2e}", flush=True)
