import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
from oracle import binding as orc
ctx = p.Context(0)
def run(name, n, r, ext, wrap, kind, camdist, seed=1, depth=50):
    rng = np.random.default_rng(seed)
    cs = [rng.uniform(0, ext, 3) for _ in range(n)]
    b = p.SceneBuilder(background=(0.7, 0.8, 1.0), bvh_seed=3)
    w = b.lambertian((0.73,) * 3)
    ids = [b.sphere(c, r, w) for c in cs]
    inner = b.bvh(ids, 0, 1) if kind == 'bvh' else b.hittable_list(ids)
    off = (-100, 270, 395)
    if wrap == 'both': inner = b.translate(b.rotate_y(inner, 15), off)
    elif wrap == 'tr': inner = b.translate(inner, off)
    desc = b.desc(b.hittable_list([inner]))
    ctr = np.array(off if wrap != 'none' else (0, 0, 0)) + ext / 2
    cam = p.camera_new(tuple(ctr + np.array([0.3, 0.1, -1.0]) * camdist), tuple(ctr), (0, 1, 0), 30, 1.0, 0.0, 10.0, 0, 1)
    prm = p.make_params(64, 64, 4, max_depth=depth, flags=1)
    img, st = ctx.render(ctx.upload(desc), cam, prm)
    ref, ost = orc.render(desc, cam, prm, precision=64, n_threads=16, count=True)
    d = np.abs(img - ref) / 4
    print(f"{name:34s} mean|d| {d.mean():.2e} bad {(d.max(axis=2) > 2e-3).mean():.4f} seg {st['segments']}/{ost['segments']} sph {st['prim_tests'][0]}/{ost['prim_tests'][0]} lds {st['bvh_in_lds']}", flush=True)
run("1000 r10 none bvh", 1000, 10, 165, 'none', 'bvh', 600)
run("1000 r10 tr bvh", 1000, 10, 165, 'tr', 'bvh', 600)
run("1000 r10 both bvh", 1000, 10, 165, 'both', 'bvh', 600)
run("1000 r10 both list", 1000, 10, 165, 'both', 'list', 600)
run("200 r10 both bvh", 200, 10, 165, 'both', 'bvh', 600)
run("200 r4 both bvh", 200, 4, 165, 'both', 'bvh', 600)
run("1000 r10 none bvh depth2", 1000, 10, 165, 'none', 'bvh', 600, depth=2)
run("1000 r10 none list", 1000, 10, 165, 'none', 'list', 600)
