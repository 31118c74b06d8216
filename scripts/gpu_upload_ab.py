"""A/B of an upload-time diagnostic switch (an environment variable rt_scene_upload reads) on one scene: the scene is uploaded once per value
and rendered a few times; the frames must be equal bit for bit. usage: gpu_upload_ab.py scene W H spp VAR value [value ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import rta
p = rta.load(); A = p._abi
name, W, H, spp, VAR = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
values = sys.argv[6:]
image = None
if name.startswith("final"):
    from PIL import Image
    image = np.asarray(Image.open(os.path.join(ROOT, "tests/golden/earthmap_rgb.png")).convert("RGB"))
hs = p.HostScene(name, 1, image=image) if image is not None else p.HostScene(name, 1)
ctx = p.Context(0)
cam = hs.camera(W / H)
frames = []
for rep in range(2):
    for v in values:
        os.environ[VAR] = v
        sc = ctx.upload(hs.desc)
        prm = p.make_params(W, H, spp, max_depth=50, seed=1, flags=A.RT_FLAG_TIMING)
        ctx.render(sc, cam, prm)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); img, st = ctx.render(sc, cam, prm); ts.append(time.perf_counter() - t)
        dt = min(ts)
        frames.append(img)
        print(f"{name} {VAR}={v}: {dt*1e3:7.2f} ms  {W*H*spp/dt/1e6:8.1f} Msamples/s   extend {st['extend_ms']:.1f} shade {st['shade_ms']:.1f} drain {st['drain_ms']:.1f} other {st['other_ms']:.1f}", flush=True)
        del sc
print("frames equal:", all(bool(np.array_equal(frames[0], f)) for f in frames[1:]))
