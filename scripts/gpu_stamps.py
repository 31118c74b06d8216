import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rta
p = rta.load()
hs = p.HostScene('book1', 1)
ctx = p.Context(0)
scene = ctx.upload(hs.desc)
cam = hs.camera(1.5)
prm = p.make_params(1200, 800, 500, flags=2)
ctx.render(scene, cam, prm)
img, st = ctx.render(scene, cam, prm)
d = st['debug']
tot = d[3]
print("waves", d[4], "refill %.1f%% node %.1f%% prim %.1f%% other %.1f%%" % (100*d[0]/tot, 100*d[1]/tot, 100*d[2]/tot, 100*(tot-d[0]-d[1]-d[2])/tot), "cycles/wave", tot/d[4], "extend_ms", st['extend_ms'], "iters", st['iterations'])
