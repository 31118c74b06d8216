"""Scene statistics as rt_scene_upload sees them: python3 scripts/gpu_scene_info.py scene[,scene...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
ctx = p.Context(0)
for which in sys.argv[1].split(","):
    if which == "final":
        from PIL import Image
        hs = p.HostScene("final", 1, image=np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB")))
    else:
        hs = p.HostScene(which, 1 if which == "book1" else 0)
    scene = ctx.upload(hs.desc)
    img, st = ctx.render(scene, hs.camera(1.0), p.make_params(64, 64, 4, flags=2))
    print(which, {k: st[k] for k in st if k not in ("debug",)}, flush=True)
    img, st = ctx.render(scene, hs.camera(1.0), p.make_params(400, 400, 64, flags=1))
    seg = st["segments"]
    print(which, "per segment: node tests %.2f" % (st["node_tests"] / seg), "prim tests [sphere moving rect tri medium enter]", [round(x / seg, 3) for x in st["prim_tests"]], "segments/sample %.2f" % (seg / st["samples"]), flush=True)
