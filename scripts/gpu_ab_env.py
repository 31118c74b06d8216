"""A/B of an upload-time environment switch on the GPU box: python3 scripts/gpu_ab_env.py ENV_VAR scene[,scene...]
(scenes: book1 | cornell | cornell_smoke | final; the switch is read by rt_scene_upload, so each setting gets its own upload)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
var = sys.argv[1]
ctx = p.Context(0)
for which in sys.argv[2].split(","):
    if which == "final":
        from PIL import Image
        hs = p.HostScene("final", 1, image=np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))); W, H, spp = 800, 800, 200
    elif which in ("cornell", "cornell_smoke"):
        hs = p.HostScene(which, 0); W, H, spp = 600, 600, 500
    else:
        hs = p.HostScene("book1", 1); W, H, spp = 1200, 800, 500
    cam = hs.camera(W / H)
    ref = None
    for setting in ("0", "1", "0", "1"):
        os.environ[var] = setting
        scene = ctx.upload(hs.desc)
        ctx.render(scene, cam, p.make_params(W, H, spp))
        _, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=2))      # per-kernel times (events around every launch)
        img, st0 = ctx.render(scene, cam, p.make_params(W, H, spp))            # the production timing
        same = "" if ref is None else (" identical_to_first=%s" % bool(np.array_equal(ref, img)))
        if ref is None: ref = img
        print(which, var, "=", setting, "extend_ms %.1f shade_ms %.1f drain_ms %.1f render_ms %.1f Msamples/s %.1f%s" % (st['extend_ms'], st['shade_ms'], st['drain_ms'], st0['render_ms'], W * H * spp / st0['render_ms'] / 1e3, same), flush=True)
        scene.close() if hasattr(scene, "close") else None
