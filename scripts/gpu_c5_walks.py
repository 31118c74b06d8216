"""The three walks of a scene in HBM on the config-5 stand-in (1 M spheres + 262 K triangles, SAH tree unless 'ref' is given), 2048^2:
binary records in one order, binary records per direction octant (round 2's default), the 8-wide tree (8 lanes per ray)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import rta
p = rta.load(); A = p._abi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
which = sys.argv[2] if len(sys.argv) > 2 else "big_sah"
ctx = p.Context(0)
hs = p.HostScene(which, 5, 1000000, 512)
W = H = 2048
cam = hs.camera(1.0)
ref = None
for name, flags in (("binary, one order", A.RT_LAYOUT_CHILD_ORDER_AS_REFERENCE), ("binary, per octant", 0), ("8-wide", A.RT_LAYOUT_WIDE_NODES)):
    t0 = time.time(); sc = ctx.upload(hs.desc, flags); up = time.time() - t0
    for count in (True, False):
        prm = p.make_params(W, H, spp, max_depth=50, seed=1, flags=A.RT_FLAG_TIMING | (A.RT_FLAG_COUNTERS if count else 0))
        img, st = ctx.render(sc, cam, prm)
        if count:
            seg = st["segments"]
            print(f"{name:20s} upload {up:5.1f} s  box tests/segment {st['node_tests']/seg:7.2f}  prim tests/segment {sum(st['prim_tests'][:5])/seg:6.2f}  segments {seg}", flush=True)
        else:
            print(f"{name:20s} {W*H*spp/st['render_ms']/1e3:8.1f} Msamples/s  k_extend {st['extend_ms']+st['drain_ms']:8.2f} ms  k_shade {st['shade_ms']:7.2f} ms  launches {st['extend_launches']}", flush=True)
    if ref is None: ref = img
    else: print(f"{'':20s} pixels that differ from the first walk's: {float((np.abs(img - ref).max(axis=2) > 0).mean()):.2e}", flush=True)
    sc.close()
