"""Soak: many renders of changing scenes, sizes, shard counts and flags in ONE process and context — every frame of a repeated (scene, params)
must be the same bits as the first time, device memory must not creep, nothing may fail. usage: python3 scripts/gpu_soak.py [rounds]"""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rta
p = rta.load(); A = p._abi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
from PIL import Image
earth = np.asarray(Image.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/earthmap_rgb.png")).convert("RGB"))
ctx = p.Context(0)
scenes = {}
for name, kw in (("book1", {}), ("book1_sah", {}), ("cornell", {}), ("cornell_smoke", {}), ("final", {"image": earth})):
    hs = p.HostScene(name, 1, **kw)
    scenes[name] = (hs, ctx.upload(hs.desc), ctx.upload(hs.desc, A.RT_LAYOUT_REFERENCE_COUNTERS))
rng = np.random.default_rng(7)
seen = {}
free0 = None
t0 = time.time(); n = 0; samples = 0
for r in range(rounds):
    for name, (hs, sc_def, sc_ref) in scenes.items():
        W, H = [(96, 64), (150, 90), (257, 131), (640, 400)][int(rng.integers(4))]
        spp = int(rng.choice([3, 16, 40]))
        flags = int(rng.choice([0, A.RT_FLAG_COUNTERS, A.RT_FLAG_TIMING, A.RT_FLAG_SAMPLE_BLOCKS]))
        world = int(rng.choice([1, 1, 2, 3]))
        shard = int(rng.integers(world))
        layout = int(rng.integers(2))
        tail = int(rng.choice([0, 1]))
        prm = p.make_params(W, H, spp, seed=int(rng.integers(2)), flags=flags, tile_size=int(rng.choice([0, 16, 32])), shard_index=shard, shard_count=world, tail_paths=tail)
        img, st = ctx.render(sc_ref if layout else sc_def, hs.camera(W / H), prm)
        assert np.isfinite(img).all() or name.startswith("cornell") or name == "final"
        # the same (scene, size, spp, seed, shard) must always give the same bits, whatever the flags, the layout and the tail's hand-over
        key = (name, W, H, spp, prm.seed, prm.tile_size, shard, world, bool(flags & A.RT_FLAG_SAMPLE_BLOCKS))
        crc = zlib.crc32(np.ascontiguousarray(img).tobytes())
        if key in seen:
            assert seen[key] == crc, ("frame changed", key)
        seen[key] = crc
        n += 1; samples += st["samples"]
    free, total = torch.cuda.mem_get_info()
    if r == 20:
        free0 = free
    if r % 10 == 0 or r == rounds - 1:
        print("round %3d: %5d renders, %.2f Gsamples, %.0f s, device memory in use %.1f MiB%s" % (r, n, samples / 1e9, time.time() - t0, (total - free) / 2**20,
              "" if free0 is None else " (round 20: %.1f)" % ((total - free0) / 2**20)), flush=True)
assert free0 is not None and abs(free - free0) < 64 * 2**20, "device memory crept"
print("soak ok: %d renders, %d distinct (scene, params), repeats bit-identical, no creep" % (n, len(seen)))
