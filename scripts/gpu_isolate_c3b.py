"""Ablation of the book-2 final scene (lit twin) on the glass_and_fog crop: which member makes the device trace more segments than the oracle?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import rta, crops as K
from oracle import binding as orc
p = rta.load(); A = p._abi
ctx = p.Context(0)
SPP = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = H = 800
cam = p.camera_new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
rect = (384, 256, 448, 320)
ti = (256 // 64) * ((W + 63) // 64) + 384 // 64; nt = ((W + 63) // 64) * ((H + 63) // 64)
def scene(drop):
    rng = np.random.default_rng(5)
    b = p.SceneBuilder(background=(0, 0, 0))
    ground = b.lambertian((0.48, 0.83, 0.53))
    boxes1 = []
    for i in range(20):
        for j in range(20):
            x0, z0 = -1000.0 + i * 100, -1000.0 + j * 100
            boxes1.append(b.box((x0, 0, z0), (x0 + 100, float(rng.uniform(1, 101)), z0 + 100), ground))
    ids = []
    if "boxes1" not in drop: ids.append(b.bvh(boxes1, 0, 1))
    ids.append(b.flip_face(b.xz_rect(123, 423, 147, 412, 554, b.diffuse_light((7, 7, 7)))))
    if "moving" not in drop: ids.append(b.moving_sphere((400, 400, 200), (430, 400, 200), 0, 1, 50, b.lambertian((0.7, 0.3, 0.1))))
    if "glass" not in drop: ids.append(b.sphere((260, 150, 45), 50, b.dielectric(1.5)))
    if "metal" not in drop: ids.append(b.sphere((0, 150, 145), 50, b.metal((0.8, 0.8, 0.9), 1.0)))
    if "sub" not in drop:
        bd = b.sphere((360, 150, 145), 70, b.dielectric(1.5))
        ids.append(bd)
        if "submedium" not in drop: ids.append(b.constant_medium(bd, 0.2, (0.2, 0.4, 0.9)))
    if "fog" not in drop: ids.append(b.constant_medium(b.sphere((0, 0, 0), 5000, b.dielectric(1.5)), 0.0001, (1, 1, 1)))
    if "earth" not in drop: ids.append(b.sphere((400, 200, 400), 100, b.lambertian((0.3, 0.4, 0.8))))
    if "perlin" not in drop: ids.append(b.sphere((220, 280, 300), 80, b.lambertian(texture=b.noise(0.1, rng))))
    if "boxes2" not in drop:
        white = b.lambertian((0.73, 0.73, 0.73))
        sp = [b.sphere(tuple(rng.uniform(0, 165, 3)), 10, white) for _ in range(1000)]
        ids.append(b.translate(b.rotate_y(b.bvh(sp, 0, 1), 15), (-100, 270, 395)))
    return b, b.desc(b.hittable_list(ids))
for drop in [(), ("fog",), ("submedium",), ("sub",), ("glass",), ("boxes2",), ("boxes1",), ("moving",), ("perlin", "earth"), ("metal",)]:
    b, desc = scene(drop)
    prm = p.make_params(W, H, SPP, max_depth=50, seed=1, flags=A.RT_FLAG_COUNTERS, tile_size=64, shard_index=ti, shard_count=nt)
    buf, st = ctx.render(ctx.upload(desc), cam, prm)
    img = buf.reshape(64, 64, 3)
    ref, ost = orc.render(desc, cam, p.make_params(W, H, SPP, max_depth=50, seed=1), precision=64, n_threads=16, rect=rect, count=True)
    ref = np.asarray(ref)[rect[1]:rect[3], rect[0]:rect[2]]
    d = np.abs(img - ref) / SPP
    print(f"drop {str(drop):22s} seg/sample gpu {st['segments']/st['samples']:.5f} orc {ost['segments']/ost['samples']:.5f} ({st['segments']/ost['segments']-1:+.5f})  mean gpu {img.mean()/SPP:.6f} orc {ref.mean()/SPP:.6f} ({img.mean()/ref.mean()-1:+.5f})  mean|d| {d.mean():.2e} bad {float((d.max(axis=2)>2e-3).mean()):.3f}", flush=True)
