"""Config 5 with and without its ground rect tested before the walk (RT_BIG_SPHERES_FIRST = 0 / 1): the one-GPU stand-in (SAH tree, 2048^2 x 16)
and the parity scene (reference-shaped tree, 1 M spheres + a torus mesh, 2048^2 x 8); the frames differ where a ray meets a sphere AT its point of contact with the ground (every sphere of this scene rests on the rect: equal t,
the later test wins) — as they do between the SAH and the reference-shaped tree of this scene. usage: python3 scripts/gpu_c5_first.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rta
p = rta.load()
ctx = p.Context(0)
for name, args, spp in (("big_sah", (5, 1000000, 512), 16), ("big", (5, 1000000, 256), 8)):
    hs = p.HostScene(name, *args)
    cam = hs.camera(1.0)
    frames = []
    for mode in ("0", "1", "0", "1"):
        os.environ["RT_BIG_SPHERES_FIRST"] = mode
        t = time.time(); scene = ctx.upload(hs.desc); up = time.time() - t
        prm = p.make_params(2048, 2048, spp, flags=2)
        ctx.render(scene, cam, prm)
        t = time.time(); img, st = ctx.render(scene, cam, prm); dt = time.time() - t
        frames.append(img)
        print(name, "RT_BIG_SPHERES_FIRST=" + mode, "%.1f ms %.1f Msamples/s extend %.1f shade %.1f drain %.1f upload %.1f s" % (dt * 1e3, 2048 * 2048 * spp / dt / 1e6, st["extend_ms"], st["shade_ms"], st["drain_ms"], up), flush=True)
        del scene
    d01 = np.abs(frames[0] - frames[1]).max(axis=2); d02 = np.abs(frames[0] - frames[2]).max(axis=2)
    print(name, "same layout twice: pixels that differ", int((d02 > 0).sum()), "| ground in the tree vs tested first: pixels that differ", int((d01 > 0).sum()), "of", d01.size,
          "(%.2e), mean |diff| of a pixel sum %.3e" % ((d01 > 0).mean(), d01.mean()), flush=True)
