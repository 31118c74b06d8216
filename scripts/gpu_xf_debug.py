import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
from oracle import binding as orc
ctx = p.Context(0)
cam = p.camera_new((0, 0, 40), (0, 0, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
def run(name, build, depth=50):
    b = p.SceneBuilder(background=(0.7, 0.8, 1.0))
    world = build(b)
    desc = b.desc(world)
    prm = p.make_params(64, 64, 4, max_depth=depth, flags=1)
    img, st = ctx.render(ctx.upload(desc), cam, prm)
    ref, ost = orc.render(desc, cam, prm, precision=64, n_threads=8, count=True)
    d = np.abs(img - ref) / 4
    print(f"{name:28s} mean|d| {d.mean():.2e} bad {(d.max(axis=2) > 2e-3).mean():.4f} seg {st['segments']}/{ost['segments']} prims {st['prim_tests'][0]}/{ost['prim_tests'][0]} info {p.compile_info(desc)['n_nodes']}", flush=True)
W = lambda b: b.lambertian((0.73, 0.73, 0.73))
rng = np.random.default_rng(0)
cs = [rng.uniform(-8, 8, 3) for _ in range(40)]
run("1 sphere", lambda b: b.hittable_list([b.sphere((0, 0, 0), 5, W(b))]))
run("1 sphere translate", lambda b: b.hittable_list([b.translate(b.sphere((0, 0, 0), 5, W(b)), (3, 1, 2))]))
run("1 sphere rotate", lambda b: b.hittable_list([b.rotate_y(b.sphere((4, 0, 0), 5, W(b)), 30)]))
run("1 sphere both", lambda b: b.hittable_list([b.translate(b.rotate_y(b.sphere((4, 0, 0), 5, W(b)), 30), (3, 1, 2))]))
run("40 list both", lambda b: b.hittable_list([b.translate(b.rotate_y(b.hittable_list([b.sphere(c, 1.5, W(b)) for c in cs]), 30), (3, 1, 2))]))
run("40 bvh plain", lambda b: b.hittable_list([b.bvh([b.sphere(c, 1.5, W(b)) for c in cs])]))
run("40 bvh both", lambda b: b.hittable_list([b.translate(b.rotate_y(b.bvh([b.sphere(c, 1.5, W(b)) for c in cs]), 30), (3, 1, 2))]))
run("40 bvh both + sphere after", lambda b: b.hittable_list([b.translate(b.rotate_y(b.bvh([b.sphere(c, 1.5, W(b)) for c in cs]), 30), (3, 1, 2)), b.sphere((0, -12, 0), 3, W(b))]))
