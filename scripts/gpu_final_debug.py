import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
from oracle import binding as orc
ctx = p.Context(0)
rng = np.random.default_rng(1)

def boxes(b, n=20):
    g = b.lambertian((0.48, 0.83, 0.53)); ids = []
    for i in range(n):
        for j in range(n):
            w = 100.0; x0 = -1000 + i * w; z0 = -1000 + j * w
            ids.append(b.box((x0, 0, z0), (x0 + w, float(rng.uniform(1, 101)), z0 + w), g))
    return b.bvh(ids, 0, 1)

def scene(parts):
    b = p.SceneBuilder(background=(0.1, 0.1, 0.1), bvh_seed=3)
    objs = []
    if "boxes" in parts: objs.append(boxes(b))
    if "light" in parts: objs.append(b.xz_rect(123, 423, 147, 412, 554, b.diffuse_light((7, 7, 7))))
    if "moving" in parts: objs.append(b.moving_sphere((400, 400, 200), (430, 400, 200), 0, 1, 50, b.lambertian((0.7, 0.3, 0.1))))
    if "glass" in parts: objs.append(b.sphere((260, 150, 45), 50, b.dielectric(1.5)))
    if "metal" in parts: objs.append(b.sphere((0, 150, 145), 50, b.metal((0.8, 0.8, 0.9), 1.0)))
    if "medium" in parts:
        bd = b.sphere((360, 150, 145), 70, b.dielectric(1.5)); objs.append(bd); objs.append(b.constant_medium(bd, 0.2, (0.2, 0.4, 0.9)))
    if "fog" in parts: objs.append(b.constant_medium(b.sphere((0, 0, 0), 5000, b.dielectric(1.5)), 0.0001, (1, 1, 1)))
    if "cloud" in parts:
        white = b.lambertian((0.73, 0.73, 0.73))
        ids = [b.sphere(rng.uniform(0, 165, 3), 10, white) for _ in range(1000)]
        objs.append(b.translate(b.rotate_y(b.bvh(ids, 0, 1), 15), (-100, 270, 395)))
    return b, b.desc(b.hittable_list(objs))

cam = p.camera_new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
for parts in (["boxes"], ["boxes", "light"], ["cloud"], ["boxes", "cloud"], ["medium"], ["fog", "boxes"], ["boxes", "light", "moving", "glass", "metal"], ["boxes", "light", "moving", "glass", "metal", "medium", "fog", "cloud"]):
    for depth in (1, 50):
        rng = np.random.default_rng(1)
        b, desc = scene(parts)
        prm = p.make_params(64, 64, 4, max_depth=depth, flags=1)
        img, st = ctx.render(ctx.upload(desc), cam, prm)
        ref, ost = orc.render(desc, cam, prm, precision=64, n_threads=16, count=True)
        d = np.abs(img - ref) / 4
        print(parts, "depth", depth, f"mean|d| {d.mean():.2e} bad {(d.max(axis=2) > 2e-3).mean():.4f} seg {st['segments']}/{ost['segments']} node {st['node_tests']}/{ost['node_tests']} prims {st['prim_tests']}/{ost['prim_tests']}", flush=True)
