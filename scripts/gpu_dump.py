"""usage: gpu_dump.py <module in scripts/> <depth> ... : renders 64x64x1spp on the GPU, saves gpurun_out/dump_<mod>_d<depth>.npy"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import rta
p = rta.load()
mod = importlib.import_module(sys.argv[1])
b, desc, cam = mod.build(p)
ctx = p.Context(0)
scene = ctx.upload(desc)
for depth in map(int, sys.argv[2:]):
    img, st = ctx.render(scene, cam, p.make_params(64, 64, 1, max_depth=depth))
    np.save(f"gpurun_out/dump_{sys.argv[1]}_d{depth}.npy", img)
    print(depth, st['segments'])
