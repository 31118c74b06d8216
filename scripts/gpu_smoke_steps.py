"""Tiny renders with a progress line each (run under `timeout`): finds the first configuration that does not come back."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
p = rta.load(); A = p._abi
ctx = p.Context(0)
hs = p.HostScene("book1", 1)
scene = ctx.upload(hs.desc)
def go(tag, W, H, spp, **kw):
    print("start", tag, flush=True)
    t = time.time()
    img, st = ctx.render(scene, hs.camera(W / H), p.make_params(W, H, spp, **kw))
    print("done ", tag, "%.1f ms" % ((time.time() - t) * 1e3), "samples", st["samples"], "segments", st["segments"], "iters", st["iterations"], "drain", st["drain_paths"], "pool", st["pool_slots"], flush=True)
os.environ["RT_DRAIN_AT"] = "0"
go("wavefront 96x64x4", 96, 64, 4)
go("wavefront 400x225x16", 400, 225, 16)
os.environ.pop("RT_DRAIN_AT")
go("default 96x64x4 (drain)", 96, 64, 4)
go("default 400x225x16", 400, 225, 16)
go("fused pool 4096", 96, 64, 4, flags=A.RT_FLAG_FUSED, pool_slots=4096)
go("default 1200x800x20", 1200, 800, 20)
