import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
from oracle import binding as orc
name = sys.argv[1] if len(sys.argv) > 1 else "big"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
t = time.time(); hs = p.HostScene(name, 5, n, 512); print("host build", time.time() - t, flush=True)
ctx = p.Context(0)
t = time.time(); scene = ctx.upload(hs.desc); print("upload+compile", time.time() - t, flush=True)
cam = hs.camera(1.0)
# parity on a small frame
W = H = 96; SPP = 4
prm = p.make_params(W, H, SPP, flags=1)
img, st = ctx.render(scene, cam, prm)
t = time.time(); ref, ost = orc.render(hs.desc, cam, prm, precision=64, n_threads=16, count=True); print("oracle", time.time() - t)
d = np.abs(img - ref) / SPP
print(f"parity mean|d| {d.mean():.2e} bad {(d.max(axis=2) > 2e-3).mean():.4f} seg {st['segments']}/{ost['segments']} node {st['node_tests']}/{ost['node_tests']} prims {st['prim_tests']}/{ost['prim_tests']}", flush=True)
p.write_png("gpurun_out/big_gpu.png", p.tonemap(img, SPP))
for (w, h, spp) in [(1024, 1024, 64)]:
    prm = p.make_params(w, h, spp, flags=2)
    ctx.render(scene, cam, p.make_params(w, h, 8))
    t = time.time(); img, st = ctx.render(scene, cam, prm); dt = time.time() - t
    print(name, w, h, spp, f"{w*h*spp/dt/1e6:.1f} Msamples/s", {k: st[k] for k in ('render_ms', 'extend_ms', 'shade_ms', 'iterations', 'segments', 'pool_slots', 'bvh_in_lds')}, flush=True)
    img2, st2 = ctx.render(scene, cam, p.make_params(512, 512, 16, flags=1)); print("  node tests/seg", st2["node_tests"]/st2["segments"], "prim tests/seg", sum(st2["prim_tests"])/st2["segments"], flush=True)
p.write_png("gpurun_out/big_gpu_2k.png", p.tonemap(img[::4, ::4], 64))
