"""Per-kind traversal counters of the device against the oracle's, crop by crop (reference-shaped layout: what the oracle counts).
usage: python scripts/gpu_kind_counters.py C3 [C3lit ...]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (first: one HIP runtime in the process)
import rta, crops as K
p = rta.load(); A = p._abi
from PIL import Image
earth = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "earthmap_rgb.png")).convert("RGB"))
ctx = p.Context(0)
KINDS = ["sphere", "moving", "rect", "tri", "medium", "instance"]
with tempfile.TemporaryDirectory() as tmp:
    for name in sys.argv[1:] or ["C3"]:
        cfg = K.CONFIGS[name]
        hs = K.host_scene(p, name, tmp, earth=earth)
        sc = ctx.upload(hs.desc, A.RT_LAYOUT_REFERENCE_COUNTERS)
        cam = hs.camera(cfg["width"] / cfg["height"])
        g = K.load_golden(name)
        for crop in cfg["crops"]:
            ti, nt = K.tile_index(name, crop)
            _, st = ctx.render(sc, cam, p.make_params(cfg["width"], cfg["height"], cfg["spp"], max_depth=50, seed=cfg["seed"], flags=A.RT_FLAG_COUNTERS,
                                                     tile_size=K.TILE, shard_index=ti, shard_count=nt))
            c = [int(v) for v in g[crop + "__counters"]]
            line = f"{name:8s} {crop:22s} seg {st['segments']/c[1]-1:+.4f} node {st['node_tests']/max(1,c[2])-1:+.4f}"
            for k, kn in enumerate(KINDS):
                if c[3 + k] or st['prim_tests'][k]:
                    line += f" | {kn} gpu {st['prim_tests'][k]} orc {c[3+k]} ({st['prim_tests'][k]/max(1,c[3+k])-1:+.4f})"
            print(line, flush=True)
