"""GPU vs f64 oracle on the other reference scenes (Cornell MIX, cornell_smoke, book-2 final, book1_ref)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
from oracle import binding as orc
from PIL import Image
os.makedirs("gpurun_out", exist_ok=True)
earth = np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))
ctx = p.Context(0)
names = sys.argv[1:] or ["cornell", "cornell_smoke", "book1_ref", "final"]
for name in names:
    hs = p.HostScene(name, 1, image=earth if name == "final" else None)
    aspect = 1.5 if name.startswith("book1") else 1.0
    W, H, SPP = (96, 64, 16) if aspect == 1.5 else (80, 80, 16)
    cam = hs.camera(aspect)
    try:
        scene = ctx.upload(hs.desc)
        prm = p.make_params(W, H, SPP, flags=1)
        t = time.time(); img, st = ctx.render(scene, cam, prm); dt = time.time() - t
    except Exception as e:
        print(name, "GPU FAILED:", e, flush=True); continue
    ref, ost = orc.render(hs.desc, cam, prm, precision=64, n_threads=16, count=True)
    d = np.abs(img.astype(np.float64) - ref) / SPP
    print(f"{name:14s} gpu mean {img.mean()/SPP:.5f} oracle {ref.mean()/SPP:.5f}  mean|d| {d.mean():.3e} frac>2e-3 {(d.max(axis=2)>2e-3).mean():.4f} "
          f"nonfinite_orc {ost['nonfinite_samples']} seg gpu/orc {st['segments']}/{ost['segments']} node {st['node_tests']}/{ost['node_tests']} prims {st['prim_tests']} / {ost['prim_tests']} {dt:.2f}s", flush=True)
    p.write_png(f"gpurun_out/{name}_gpu.png", p.tonemap(img, SPP))
    p.write_png(f"gpurun_out/{name}_orc.png", p.tonemap(ref.astype(np.float32), SPP))
