import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
ctx = p.Context(0)
rng = np.random.default_rng(1)
cs = [rng.uniform(0, 165, 3) for _ in range(1000)]
b = p.SceneBuilder(background=(0.7, 0.8, 1.0), bvh_seed=3)
w = b.lambertian((0.73,) * 3)
ids = [b.sphere(c, 10, w) for c in cs]
desc = b.desc(b.hittable_list([b.hittable_list(ids)]))
ctr = np.array([82.5] * 3)
cam = p.camera_new(tuple(ctr + np.array([0.3, 0.1, -1.0]) * 600), tuple(ctr), (0, 1, 0), 30, 1.0, 0.0, 10.0, 0, 1)
for depth in (2, 3):
    prm = p.make_params(64, 64, 1, max_depth=depth)
    img, st = ctx.render(ctx.upload(desc), cam, prm)
    np.save(f"gpurun_out/dbg_gpu_d{depth}.npy", img)
print("ok")
