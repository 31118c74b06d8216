"""Config-5 scene (1 M spheres + 262 K triangles) at 2048x2048: the three walks of a scene that does not fit LDS — 16-byte compressed
records (default), 32-byte records with the top of the tree in LDS (RT_NODE16=0, RT_TOP_NODES=n), 32-byte records in HBM only
(RT_TOP_NODES=0) — on the SAH and the reference-shaped tree, plus the 16-byte records in ONE order (RT_OCTANT_ORDER=0). One line per
setting. Every walk finds the same closest hits up to hits that tie within rounding — a million overlapping spheres have many pairs of
surfaces a few ulps apart, and which of two such hits survives depends on which boxes were entered with which t_max; the share of
differing pixels is printed (about 2e-4)."""
import numpy as np
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rta
p = rta.load()
ctx = p.Context(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
W = H = 2048
for name in ("big_sah", "big"):
    hs = p.HostScene(name, 5, 1000000, 512)
    cam = hs.camera(1.0)
    ref = None
    for n16, top, octs in (("1", 0, "0"), ("1", 0, "1"), ("0", 0, "1"), ("0", 1024, "1"), ("0", 4096, "1")):
        os.environ["RT_NODE16"] = n16; os.environ["RT_TOP_NODES"] = str(top); os.environ["RT_OCTANT_ORDER"] = octs
        scene = ctx.upload(hs.desc)
        best = None
        for r in range(2):
            t = time.time(); img, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=2)); dt = time.time() - t
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        _, c = ctx.render(scene, cam, p.make_params(W, H, max(1, spp // 8), flags=1))
        same = "ref" if ref is None else ("same" if (img == ref).all() else "%.2e of the pixels differ" % float((np.abs(img - ref).max(axis=2) > 0).mean()))
        if ref is None:
            ref = img
        print(f"{name:8s} node16 {n16} octants {octs} top {top:5d} -> lds_top {st['lds_top_nodes']:5d} geom {st['debug'][6:8]} {dt*1e3:8.1f} ms {W*H*spp/dt/1e6:7.1f} Msamples/s extend {st['extend_ms']:7.1f} "
              f"shade {st['shade_ms']:6.1f} drain {st['drain_ms']:5.1f} node/seg {c['node_tests']/c['segments']:6.1f} prim/seg {sum(c['prim_tests'][:5])/c['segments']:5.2f} frame {same}", flush=True)
        scene.close()
