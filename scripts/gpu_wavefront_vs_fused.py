import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load(); A = p._abi
ctx = p.Context(0)
def build(n, seed):
    rng = np.random.default_rng(seed)
    b = p.SceneBuilder(background=(0.7, 0.8, 1.0), background_mode=A.RT_BG_SKY_GRADIENT)
    lam = b.lambertian((0.5, 0.5, 0.5)); glass = b.dielectric(1.5)
    ids = []
    for i in range(n):
        c = np.array([rng.uniform(-2, 2), 0.3, rng.uniform(-2, 2)])
        if i % 2: ids.append(b.moving_sphere(c, c + (0, rng.uniform(0.1, 0.5), 0), 0.0, 1.0, 0.3, lam))
        else: ids.append(b.sphere(c, 0.3, glass))
    return b, b.desc(b.bvh(ids, 0.0, 1.0))
cam = p.camera_new((5, 2, 2), (0, 0.3, 0), (0, 1, 0), 40, 1.5, 0.0, 10.0, 0.0, 0.0)
for n in (2, 3, 4, 5, 6, 8, 12, 20):
    for seed in range(6):
        b, desc = build(n, seed)
        sc = ctx.upload(desc)
        os.environ["RT_DRAIN_AT"] = "0"
        w, sw = ctx.render(sc, cam, p.make_params(96, 64, 4, seed=3, max_depth=2, flags=A.RT_FLAG_COUNTERS))
        f, sf = ctx.render(sc, cam, p.make_params(96, 64, 4, seed=3, max_depth=2, flags=A.RT_FLAG_COUNTERS | A.RT_FLAG_FUSED))
        if 0: print("NaN pixels: wavefront", int(np.isnan(w).any(axis=2).sum()), "fused", int(np.isnan(f).any(axis=2).sum()), "segments", sw["segments"], sf["segments"], "samples", sw["samples"], sf["samples"], "iters", sw["iterations"], sf["iterations"])
        nd = int((~np.isclose(w, f, equal_nan=True)).any(axis=2).sum())
        if nd:
            info = p.compile_info(desc)
            nodes, sph, meta = p.compile_dump(desc)
            print("n", n, "seed", seed, "differing", nd, "nodes", sw["node_tests"], sf["node_tests"], "prims", sw["prim_tests"][:2], sf["prim_tests"][:2], "n_nodes", info["n_nodes"])
            ys, xs = np.nonzero(np.abs(w - f).max(axis=2))
            for y, x in list(zip(ys, xs))[:3]:
                print("   px", x, y, w[y, x], f[y, x])
            continue
print("no mismatch found")
