"""Render one scene a few times and nothing else (for rocprofv3 --pmc / --kernel-trace passes).
usage: python3 scripts/gpu_render_once.py <scene> <W> <H> <spp> [reps] [flags]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
image = None
if name.startswith("final"):
    from PIL import Image
    image = np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))
hs = p.HostScene(name, 1, image=image) if image is not None else p.HostScene(name, 1)
ctx = p.Context(0)
scene = ctx.upload(hs.desc)
cam = hs.camera(W / H)
for r in range(reps):
    t = time.time()
    img, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=flags))
    dt = time.time() - t
    print(name, W, H, spp, "%.1f ms" % (dt * 1e3), "%.1f Msamples/s" % (W * H * spp / dt / 1e6), "segments", st["segments"], "iters", st["iterations"], "geom", st["debug"][6:8],
          "extend %.1f shade %.1f other %.1f ms" % (st["extend_ms"], st["shade_ms"], st["other_ms"]), flush=True)
