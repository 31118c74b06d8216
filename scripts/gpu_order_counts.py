"""Visit counts of a small scene walked from HBM (RT_LDS_SCENE=0) with one record order and with the near-first order per direction
octant: what near-first ordering would buy the LDS walk. usage: python3 scripts/gpu_order_counts.py [book1|book1_sah|final|cornell]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
ctx = p.Context(0)
os.environ["RT_LDS_SCENE"] = "0"
for which in (sys.argv[1] if len(sys.argv) > 1 else "book1,book1_sah").split(","):
    if which == "final":
        from PIL import Image
        hs = p.HostScene("final", 1, image=np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB")))
    else:
        hs = p.HostScene(which, 1)
    for setting in ("0", "1"):
        os.environ["RT_OCTANT_ORDER"] = setting
        scene = ctx.upload(hs.desc)
        img, st = ctx.render(scene, hs.camera(1.5), p.make_params(600, 400, 32, flags=1))
        seg = st["segments"]
        print(which, "octant order", setting, "in_lds", st["bvh_in_lds"], "node tests/seg %.2f" % (st["node_tests"] / seg), "prim tests/seg", [round(x / seg, 2) for x in st["prim_tests"][:5]], flush=True)
