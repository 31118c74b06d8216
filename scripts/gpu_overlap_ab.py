"""A/B of the two-lane render loop (RT_OVERLAP=1: the pool's halves on two streams, one half shaded while the other is walked) on the
bench workload; the frames must be equal bit for bit. A fifth argument names another render-time diagnostic switch to A/B instead
(RT_TILE_SEARCH: item -> tile by search in tile_prefix instead of by arithmetic). usage: gpu_overlap_ab.py [scene W H spp [VAR]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import rta
p = rta.load(); A = p._abi
name = sys.argv[1] if len(sys.argv) > 1 else "book1"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1200, 800, 500)
VAR = sys.argv[5] if len(sys.argv) > 5 else "RT_OVERLAP"
image = None
if name.startswith("final"):
    from PIL import Image
    image = np.asarray(Image.open(os.path.join(ROOT, "tests/golden/earthmap_rgb.png")).convert("RGB"))
hs = p.HostScene(name, 1, image=image) if image is not None else p.HostScene(name, 1)
ctx = p.Context(0)
sc = ctx.upload(hs.desc)
cam = hs.camera(W / H)
frames = {}
for mode in ("0", "1", "0", "1"):
    os.environ[VAR] = mode
    prm = p.make_params(W, H, spp, max_depth=50, seed=1, flags=A.RT_FLAG_TIMING)
    ctx.render(sc, cam, prm)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); img, st = ctx.render(sc, cam, prm); ts.append(time.perf_counter() - t)
    dt = min(ts)
    frames[mode] = img
    print(f"{name} {VAR}={mode}: {dt*1e3:7.2f} ms  {W*H*spp/dt/1e6:8.1f} Msamples/s   sum of kernel times: extend {st['extend_ms']:.1f} shade {st['shade_ms']:.1f} drain {st['drain_ms']:.1f} other {st['other_ms']:.1f} launches {st['extend_launches']}", flush=True)
print("frames equal:", bool(np.array_equal(frames["0"], frames["1"])))
