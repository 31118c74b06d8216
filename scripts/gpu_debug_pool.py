import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
hs = p.HostScene('book1', 1)
ctx = p.Context(0)
scene = ctx.upload(hs.desc)
W, H = 150, 90
cam = hs.camera(W / H)
for SPP, depth in [(20, 50), (20, 1), (16, 50), (4, 50), (20, 2)]:
    a, _ = ctx.render(scene, cam, p.make_params(W, H, SPP, seed=5, max_depth=depth))
    for pool in (4096, 8192, 30720, 1024):
        c, st = ctx.render(scene, cam, p.make_params(W, H, SPP, seed=5, max_depth=depth, pool_slots=pool))
        c2, _ = ctx.render(scene, cam, p.make_params(W, H, SPP, seed=5, max_depth=depth, pool_slots=pool))
        d = np.abs(a - c)
        ys, xs = np.nonzero(d.max(axis=2))
        print(f"spp {SPP} depth {depth} pool {pool}: differing pixels {len(ys)} max {d.max():.3g} self-consistent {np.array_equal(c, c2)} iters {st['iterations']} segs {st['segments']}",
              list(zip(ys[:6].tolist(), xs[:6].tolist())), flush=True)
