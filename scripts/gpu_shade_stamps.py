"""In-kernel s_memtime stamps of k_shade (-DRT_SHADE_STAMPS build: RT_HIP_LIB=.../librt_hip_sstamps.so): where a wave's life goes — waiting
for the queue size, for its path records, in shade_segment (primitive -> material -> texture -> scatter), in the compaction
(barriers + one returning atomic per workgroup) and for its stores. Every stamp drains the wave's memory counters, so the phases do not
overlap as they do in the product build: shares only. usage: python3 scripts/gpu_shade_stamps.py [book1|final|cornell]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
which = sys.argv[1] if len(sys.argv) > 1 else "book1"
ctx = p.Context(0)
if which == "final":
    from PIL import Image
    hs = p.HostScene("final", 1, image=np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))); W, H, spp = 800, 800, 200
elif which == "cornell":
    hs = p.HostScene("cornell", 0); W, H, spp = 600, 600, 500
else:
    hs = p.HostScene("book1", 1); W, H, spp = 1200, 800, 500
scene = ctx.upload(hs.desc)
cam = hs.camera(W / H)
prm = p.make_params(W, H, spp, flags=2, tail_paths=1)
ctx.render(scene, cam, prm)
img, st = ctx.render(scene, cam, prm)
waves = st["node_tests"]; ph = st["prim_tests"][:5]; tot = sum(ph)
names = ["queue size", "path records", "shade_segment", "compaction", "stores"]
print(which, "waves", waves, "cycles/wave (100 MHz) %.0f = %.2f us" % (tot / waves, tot / waves / 100.0), "shade_ms %.1f" % st["shade_ms"])
print("  " + "; ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, ph)))
d = st["debug"]
print("  end of sample + next camera ray: %.0f cycles a wave (inside the shade_segment share)" % (d[3] / waves))
if d[2]:
    print("  of shade_segment (waves whose first lane hit a primitive: %d): primitive record %.0f cycles, material record %.0f cycles, whole phase %.0f" % (d[2], d[0] / d[2], d[1] / d[2], ph[2] / waves))
