import numpy as np
def build(p):
    rng = np.random.default_rng(1)
    b = p.SceneBuilder(background=(0.1, 0.1, 0.1), bvh_seed=3)
    g = b.lambertian((0.48, 0.83, 0.53)); ids = []
    for i in range(20):
        for j in range(20):
            w = 100.0; x0 = -1000 + i * w; z0 = -1000 + j * w
            ids.append(b.box((x0, 0, z0), (x0 + w, float(rng.uniform(1, 101)), z0 + w), g))
    objs = [b.bvh(ids, 0, 1), b.constant_medium(b.sphere((0, 0, 0), 5000, b.dielectric(1.5)), 0.0001, (1, 1, 1))]
    desc = b.desc(b.hittable_list(objs))
    cam = p.camera_new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
    return b, desc, cam
