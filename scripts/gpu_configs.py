"""Throughput of every BASELINE config on one MI355X (C5 at reduced size), into gpurun_out/configs.json."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rta
from PIL import Image
p = rta.load()
ctx = p.Context(0)
earth = np.asarray(Image.open("tests/golden/earthmap_rgb.png").convert("RGB"))
out = {}
def run(tag, hs, W, H, spp, aspect=None, png=True):
    t = time.time(); scene = ctx.upload(hs.desc); t_up = time.time() - t
    cam = hs.camera(aspect or W / H)
    ctx.render(scene, cam, p.make_params(W, H, spp))
    t = time.time(); img, st = ctx.render(scene, cam, p.make_params(W, H, spp, flags=2)); dt = time.time() - t
    _, c = ctx.render(scene, cam, p.make_params(W, H, max(1, spp // 50), flags=1))
    out[tag] = dict(width=W, height=H, spp=spp, seconds=round(dt, 3), msamples_per_s=round(W * H * spp / dt / 1e6, 1), upload_s=round(t_up, 2),
                    extend_ms=round(st["extend_ms"], 1), shade_ms=round(st["shade_ms"], 1), segments_per_sample=round(st["segments"] / st["samples"], 3),
                    node_tests_per_segment=round(c["node_tests"] / c["segments"], 2), prim_tests_per_segment=[round(x / c["segments"], 2) for x in c["prim_tests"]],
                    bvh_in_lds=st["bvh_in_lds"], scene_nodes=st["scene_nodes"], scene_prims=st["scene_prims"], iterations=st["iterations"])
    print(tag, out[tag], flush=True)
    if png:
        rgb = p.tonemap(img, spp)
        p.write_png(f"gpurun_out/{tag}.png", rgb[::max(1, H // 400), ::max(1, W // 400)])
run("C2_book1_1200x800x500", p.HostScene("book1", 1), 1200, 800, 500)
run("C2_book1_sah", p.HostScene("book1_sah", 1), 1200, 800, 500, png=False)
run("C3_final_800x800x1000", p.HostScene("final", 1, image=earth), 800, 800, 1000)
run("C4_cornell_600x600x1000", p.HostScene("cornell", 0), 600, 600, 1000)
run("cornell_smoke_600x600x200", p.HostScene("cornell_smoke", 0), 600, 600, 200)
run("C5_1M_sah_2048x2048x64", p.HostScene("big_sah", 5, 1000000, 512), 2048, 2048, 64)
json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)
