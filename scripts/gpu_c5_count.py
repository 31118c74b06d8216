import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rta
p = rta.load()
ctx = p.Context(0)
hs = p.HostScene("big_sah", 5, 1000000, 512)
scene = ctx.upload(hs.desc)
cam = hs.camera(1.0)
W = H = 2048; spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
import struct
def f(u): return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]
for md in (16,):
    t = time.time(); img, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=1, max_depth=md)); dt = time.time() - t
    print(os.environ.get("RT_HIP_LIB", "default").split("_")[-1], "max_depth", md, "%.1f ms" % (dt * 1e3), "segments", st["segments"], "node tests/seg %.2f" % (st["node_tests"] / st["segments"]),
          "prim/seg", [round(x / st["segments"], 3) for x in st["prim_tests"]], flush=True)
    dbg = st["debug"]
    print("long walks:", dbg[0], "o", f(dbg[1] >> 32), f(dbg[1]), f(dbg[2] >> 32), "d", f(dbg[2]), f(dbg[3] >> 32), f(dbg[3]), "tmax", f(dbg[4] >> 32), "from", hex(dbg[4] & 0xFFFFFFFF))
