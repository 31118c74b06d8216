"""First GPU bring-up: book-1 on the HIP path vs the f64 oracle, same seed."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
from oracle import binding as orc

os.makedirs("gpurun_out", exist_ok=True)
hs = p.HostScene('book1', 1)
ctx = p.Context(0)
scene = ctx.upload(hs.desc)
W, H, SPP = 240, 160, 16
cam = hs.camera(W / H)
prm = p.make_params(W, H, SPP, flags=3)
t = time.time(); img, st = ctx.render(scene, cam, prm); dt = time.time() - t
print("gpu", dt, st, flush=True)
ref, ost = orc.render(hs.desc, cam, prm, precision=64, n_threads=16, count=True)
print("oracle", ost, flush=True)
d = np.abs(img.astype(np.float64) - ref) / SPP
print("mean abs diff", d.mean(), "max", d.max(), "frac>1e-3", (d.max(axis=2) > 1e-3).mean(), "frac>1e-2", (d.max(axis=2) > 1e-2).mean())
print("gpu mean", img.mean(axis=(0, 1)) / SPP, "oracle mean", ref.mean(axis=(0, 1)) / SPP)
print("Vn/seg gpu", st['node_tests'] / st['segments'], "oracle", ost['node_tests'] / ost['segments'])
print("Vp/seg gpu", st['prim_tests'][0] / st['segments'], "oracle", ost['prim_tests'][0] / ost['segments'])
p.write_png("gpurun_out/book1_gpu.png", p.tonemap(img, SPP))
p.write_png("gpurun_out/book1_orc.png", p.tonemap(ref.astype(np.float32), SPP))
# throughput
for (w, h, spp) in [(600, 400, 50), (1200, 800, 100)]:
    cam = hs.camera(w / h)
    prm = p.make_params(w, h, spp, flags=2)
    ctx.render(scene, cam, prm)
    t = time.time(); img, st = ctx.render(scene, cam, prm); dt = time.time() - t
    print(w, h, spp, "Msamples/s", w * h * spp / dt / 1e6, {k: st[k] for k in ('render_ms', 'extend_ms', 'shade_ms', 'other_ms', 'iterations', 'segments', 'pool_slots', 'bvh_in_lds')}, flush=True)
