"""Odd parameter corners through the C ABI: every case must equal its own pool-limited re-render bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load()
A = p._abi
ctx = p.Context(0)
from PIL import Image
earth = np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))
scenes = {"book1": p.HostScene("book1", 1), "cornell": p.HostScene("cornell", 0), "final": p.HostScene("final", 1, image=earth), "smoke": p.HostScene("cornell_smoke", 0)}
up = {k: ctx.upload(v.desc) for k, v in scenes.items()}
bad = 0
for name, hs in scenes.items():
    for (W, H, spp, depth, flags, tile) in [(2, 2, 1, 50, 0, 0), (7, 5, 3, 50, 0, 8), (7, 5, 3, 50, A.RT_FLAG_SAMPLE_BLOCKS, 8), (33, 17, 17, 1, 0, 16),
                                            (64, 40, 37, 50, A.RT_FLAG_SAMPLE_BLOCKS, 32), (129, 65, 5, 3, 0, 32)]:
        cam = hs.camera(W / H)
        a, sa = ctx.render(up[name], cam, p.make_params(W, H, spp, max_depth=depth, seed=9, flags=flags, tile_size=tile))
        b, sb = ctx.render(up[name], cam, p.make_params(W, H, spp, max_depth=depth, seed=9, flags=flags, tile_size=tile, pool_slots=256))
        ok = np.array_equal(a, b) and np.isfinite(a).all() and sa["samples"] == W * H * spp == sb["samples"] and sa["segments"] == sb["segments"]
        bad += not ok
        print(name, (W, H, spp, depth, flags, tile), "ok" if ok else "MISMATCH", "segments", sa["segments"], "iters", sa["iterations"], sb["iterations"], flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
