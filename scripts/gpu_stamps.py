"""In-kernel s_memtime stamps of k_extend (-DRT_STAMPS build: RT_HIP_LIB=.../librt_hip_stamps.so): share of wave time in refill / node pass / primitive pass.
usage: python3 scripts/gpu_stamps.py [book1|c5|final|cornell] (the stamps themselves cost about as much as the work between them: ratios only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
which = sys.argv[1] if len(sys.argv) > 1 else "book1"
ctx = p.Context(0)
if which == "c5":
    hs = p.HostScene("big_sah", 5, 1000000, 512); W, H, spp = 2048, 2048, 16
elif which == "final":
    from PIL import Image
    hs = p.HostScene("final", 1, image=np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))); W, H, spp = 400, 400, 100
elif which == "cornell":
    hs = p.HostScene("cornell", 0); W, H, spp = 600, 600, 500
else:
    hs = p.HostScene("book1", 1); W, H, spp = 1200, 800, 500
scene = ctx.upload(hs.desc)
cam = hs.camera(W / H)
os.environ["RT_DRAIN_AT"] = "0"
prm = p.make_params(W, H, spp, flags=2)
ctx.render(scene, cam, prm)
img, st = ctx.render(scene, cam, prm)
d = st['debug']
tot = d[3]
print(which, "waves", d[4], "refill %.1f%% node %.1f%% prim %.1f%% other %.1f%%" % (100*d[0]/tot, 100*d[1]/tot, 100*d[2]/tot, 100*(tot-d[0]-d[1]-d[2])/tot), "cycles/wave %.0f" % (tot/d[4]),
      "extend_ms %.1f" % st['extend_ms'], "shade_ms %.1f" % st['shade_ms'], "iters", st['iterations'], "geom", st['debug'][6:8])
if d[5]:
    print("  node steps per wave %.0f; lanes on a tree record when a step begins: %.1f of 64 (the rest are parked at a leaf, idle or done)" % (d[5] / d[4], st["prim_tests"][5] / d[5]))
if not (prm.flags & 1):
    names = ["sphere", "moving", "rect", "tri", "medium"]
    print("  prim passes/wave %.1f;" % (st["node_tests"] / d[4]), " ".join("%s: %.1f passes/wave, %.1f lanes/pass;" % (names[k], (v >> 40) / d[4], (v & ((1 << 40) - 1)) / max(1, v >> 40)) for k, v in enumerate(st["prim_tests"][:5]) if v), "segments/wave %.0f" % (st["segments"] / d[4]))
