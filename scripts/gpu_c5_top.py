"""Config-5 scene (1 M spheres + 262 K triangles) at 2048x2048: k_extend with the top of the tree in LDS (RT_TOP_NODES records) against
the HBM-only walk (RT_TOP_NODES=0), SAH and reference-shaped trees. One line per setting; counters from a 1/4-spp counting pass."""
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
    for top in (0, 512, 1024, 2048, 4096):
        os.environ["RT_TOP_NODES"] = str(top)
        scene = ctx.upload(hs.desc)
        best = None
        for r in range(2):
            t = time.time(); img, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=2)); dt = time.time() - t
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        same = "ref" if ref is None else ("same" if (img == ref).all() else "DIFFERENT")
        if ref is None:
            ref = img
        print(f"{name:8s} top {top:5d} -> lds_top_nodes {st['lds_top_nodes']:5d} geom {st['debug'][6:8]} {dt*1e3:8.1f} ms {W*H*spp/dt/1e6:7.1f} Msamples/s extend {st['extend_ms']:7.1f} shade {st['shade_ms']:6.1f} iters {st['iterations']} frame {same}", flush=True)
        scene.close()
