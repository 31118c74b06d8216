import sys; sys.path.insert(0,".")
import numpy as np, rta
p = rta.load(); ctx = p.Context(0)
hs = p.HostScene("big_sah", 5, 200000, 256); rs = p.HostScene("big", 5, 200000, 256)
cam = hs.camera(1.0); prm = p.make_params(96, 96, 16, flags=1)
a, sa = ctx.render(ctx.upload(hs.desc), cam, prm); b, sb = ctx.render(ctx.upload(rs.desc), cam, prm)
d = np.abs(a-b).max(axis=2); ys, xs = np.nonzero(d)
print("differing pixels", len(ys), "max", d.max(), "segments", sa["segments"], sb["segments"])
for n in (0, 256):
    h2 = p.HostScene("big_sah", 5, 200000, n); r2 = p.HostScene("big", 5, 200000, n)
    a, _ = ctx.render(ctx.upload(h2.desc), cam, prm); b, _ = ctx.render(ctx.upload(r2.desc), cam, prm)
    print("mesh subdiv", n, "differing", (np.abs(a-b).max(axis=2) > 0).sum())
