"""BASELINE config 5 stand-in (1M spheres + 262K triangles, SAH) at 2048x2048 — one line per run, then the segments of all runs
(what the PMC passes divide by)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rta
p = rta.load()
ctx = p.Context(0)
hs = p.HostScene("big_sah", 5, 1000000, 512)
scene = ctx.upload(hs.desc)
cam = hs.camera(1.0)
W = H = 2048; spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
total = 0
for r in range(2):
    t = time.time(); img, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=2)); dt = time.time() - t
    total += st["segments"]
    print(os.environ.get("RT_HIP_LIB", "default").split("_")[-1], "%.1f ms %.1f Msamples/s extend %.1f shade %.1f drain %.1f iters %d" % (dt * 1e3, W * H * spp / dt / 1e6, st["extend_ms"], st["shade_ms"], st["drain_ms"], st["iterations"]), flush=True)
print("segments", total, "scene_nodes", st["scene_nodes"], "scene_prims", st["scene_prims"])
