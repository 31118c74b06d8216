"""GPU parity AT THE BENCHMARKED SIZES (-m gpu, through the C ABI): the full frames of BASELINE configs C2..C5 are rendered
on the device and four 64x64 crops of each are compared with the f64 oracle's output for the same pixels and seed
(tests/golden/crops_<config>.npz, written by tests/golden/make_crops.py; tests/test_full_frame_fixtures.py re-derives them
on the CPU).

Per crop: mean |diff| of the linear per-pixel mean, fraction of pixels whose mean is off by more than 2e-3, 8-bit agreement,
and — by rendering exactly that tile as a shard of its own with RT_FLAG_COUNTERS — segment, AABB-test and primitive-test
counts against the oracle's counters for the crop; the one-tile shard must equal the full frame's pixels bit for bit.

Tolerances are about twice what was measured on MI355X (TOL below; the measured values are in DESIGN.md section 2). Float
path: the reference computes in f64, the device in f32, so single samples take other branches at rejection tests / Schlick
draws / grazing hits; at 500-1000 spp every pixel holds a few such samples and the statistic is the size of their sum."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import crops as K   # noqa: E402

pytestmark = pytest.mark.gpu

# per crop, worst crop of the config: mean|diff| of the per-pixel mean, pixels off by > 2e-3, 8-bit channels within +-2, then the counters of the
# crop rendered as a one-tile shard against the oracle's: segments (relative), AABB tests and primitive tests (relative excess)
TOL = {
    "C2": dict(mean=4e-5, bad=1e-3, bit8=0.999, seg=6e-5, node=1e-3, prim=1e-3, zmean=0.015, se4=1e-3),       # measured 1.9e-5, 0, 1.0, 2.4e-5, 2.8e-4, 1.4e-4, z 0.0068, 0
    # the book-2 frame is nearly black under the reference's DiffuseLight (front face only): mean radiance ~2e-3, so sqrt gamma turns
    # tiny linear differences into 8-bit steps; the r = 5000 fog sphere makes the scene extent (hence the box padding) large
    "C3": dict(mean=1.5e-4, bad=0.027, bit8=0.93, seg=5e-3, node=0.015, prim=0.015),     # measured 7.3e-5, 0.014, 0.965, 2.4e-3, 7.0e-3, 6.6e-3
    # the lit twins (tests/crops.py): FlipFace around the light, everything else as the literal scenes. At radiance 0.2-1.0 the absolute
    # "pixels off by > 2e-3" says nothing (bad = 1.0 switches it off for C3lit); the scale there is the pixel's standard error (zmean, se4).
    # C3lit's worst crop is the one over the 1000-sphere cluster, where 2 % of the paths diverge from the f64 oracle's
    "C3lit": dict(mean=6e-3, bad=1.0, bit8=0.94, seg=6e-3, node=0.015, prim=0.015, zmean=0.15, se4=1e-3),   # measured 2.8e-3, -, 0.9696 (other crops >= 0.9997), 2.7e-3, 6.5e-3, 6.9e-3, z 0.068, 0
    "SMOKElit": dict(mean=6e-5, bad=0.025, bit8=0.999, seg=7e-5, node=1e-9, prim=7e-5, zmean=0.005, se4=1e-3),   # measured 3.0e-5, 0.011, 0.9998, 3.3e-5, 0, 3.3e-5, z 0.0012, 0
    "C4": dict(mean=2.6e-4, bad=0.014, bit8=0.997, seg=2e-5, node=1e-9, prim=2e-5, zmean=0.01, se4=1e-3),    # measured 1.3e-4, 6.8e-3, 0.9987, 7e-6, 0 (no BVH), 5e-6, z 0.0032, 0
    # C5 walks 16-byte compressed records (corners on a u16 grid over the scene): boxes a grid step looser, so more tests — culling only
    "C5": dict(mean=4e-4, bad=0.025, bit8=0.99, seg=7e-4, node=0.05, prim=0.16),       # measured 2.0e-4, 0.012, 0.996, 3.4e-4, 2.4e-2, 8.1e-2
}
METRICS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_metrics.jsonl")


def record(**kw):
    try:
        os.makedirs(os.path.dirname(METRICS), exist_ok=True)
        with open(METRICS, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def crop_metrics(pkg, a, ref, spp, se=None):
    d = np.abs(a.astype(np.float64) - ref) / spp
    a8, b8 = pkg.tonemap(np.ascontiguousarray(a), spp).astype(int), pkg.tonemap(np.ascontiguousarray(ref.astype(np.float32)), spp).astype(int)
    m = dict(mean=float(d.mean()), bad=float((d.max(axis=2) > 2e-3).mean()), bit8=float(np.mean(np.abs(a8 - b8) <= 2)),
             rel_mean=float(abs(a.mean() - ref.mean()) / max(ref.mean(), 1e-30)))
    if se is not None:
        # SURVEY 8(d)'s converged-mean test: |difference of the per-pixel means| against the pixel's standard error (the oracle's own
        # per-sample spread / sqrt(spp), in the fixture). z = |d| / SE: an independent re-render would sit at ~1.1 on average; the same
        # paths followed in f32 must sit far below it. se4 = share of pixel channels beyond 4 SE.
        z = d / np.maximum(se.astype(np.float64), 1e-9)
        lit = se > 1e-9                 # a channel no sample ever lit has SE 0: there the values must agree outright
        m["zmean"] = float(z[lit].mean()) if lit.any() else 0.0
        m["se4"] = float((z[lit] > 4.0).mean()) if lit.any() else 0.0
    return m


def check_config(pkg, gpu, name, tmp_path, earth, sah=False):
    A = pkg._abi
    cfg, tol = K.CONFIGS[name], TOL[name]
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    hs = K.host_scene(pkg, name, tmp_path, sah=sah, earth=earth)
    scene = gpu.upload(hs.desc)
    scene_counts = gpu.upload(hs.desc, A.RT_LAYOUT_REFERENCE_COUNTERS)          # for the counters: the layout the reference walks (include/rt_hip.h)
    cam = hs.camera(W / H)
    img, st = gpu.render(scene, cam, pkg.make_params(W, H, spp, max_depth=50, seed=cfg["seed"]))
    assert np.isfinite(img).all() and st["samples"] == W * H * spp
    g = K.load_golden(name)
    results = []
    for crop, (x0, y0, x1, y1) in cfg["crops"].items():
        a, ref, ctr = img[y0:y1, x0:x1], g[crop], [int(v) for v in g[crop + "__counters"]]
        m = crop_metrics(pkg, a, ref, spp, g.get(crop + "__se"))
        # the same tile as a one-tile shard, with the device counters on
        ti, n_tiles = K.tile_index(name, crop)
        buf, ts = gpu.render(scene_counts, cam, pkg.make_params(W, H, spp, max_depth=50, seed=cfg["seed"], flags=A.RT_FLAG_COUNTERS, tile_size=K.TILE,
                                                         shard_index=ti, shard_count=n_tiles))
        if name == "C5":
            # the reference's child order and the near-first orders (one record array per direction octant) find the same closest hits up
            # to hits that tie within rounding (as for the SAH tree below); the tile of the production layout is the full frame's, bit for bit
            differ = float((np.abs(buf.reshape(K.TILE, K.TILE, 3) - a).max(axis=2) > 0).mean())
            assert differ < 2e-3, (name, crop, differ)
            own, _ = gpu.render(scene, cam, pkg.make_params(W, H, spp, max_depth=50, seed=cfg["seed"], tile_size=K.TILE, shard_index=ti, shard_count=n_tiles))
            assert np.array_equal(own.reshape(K.TILE, K.TILE, 3), a), (name, crop, "a tile rendered alone differs from the full frame")
        else:
            assert np.array_equal(buf.reshape(K.TILE, K.TILE, 3), a), (name, crop, "a tile rendered alone differs from the full frame")
        m["seg"] = abs(ts["segments"] - ctr[1]) / ctr[1]
        m["node"] = (ts["node_tests"] - ctr[2]) / max(1, ctr[2])
        prim_gpu, prim_orc = sum(ts["prim_tests"][:5]), sum(ctr[3:8])
        m["prim"] = (prim_gpu - prim_orc) / max(1, prim_orc)
        record(config=name + ("_sah" if sah else ""), crop=crop, **m)
        assert ts["samples"] == ctr[0]
        results.append((crop, m))
    for crop, m in results:          # every crop is measured (and recorded) before the first assert
        assert m["mean"] <= tol["mean"] and m["bad"] <= tol["bad"] and m["bit8"] >= tol["bit8"], (name, crop, m)
        if "zmean" in m and "zmean" in tol:
            # (not for the literal C3: a frame whose pixels are mostly a few rare bright samples has no usable per-pixel standard error)
            assert m["zmean"] <= tol["zmean"] and m["se4"] <= tol["se4"] and m["rel_mean"] <= 5e-3, (name, crop, m)
        assert m["seg"] <= tol["seg"], (name, crop, m)
        if not sah:
            # same tree, same order: the device's boxes are a hair looser (they absorb the slab test's rounding), never tighter
            assert -1e-3 <= m["node"] <= tol.get("node", 2e-2) and abs(m["prim"]) <= tol.get("prim", 3e-2), (name, crop, m)
    return hs, scene, cam, img


def test_c2_book1_1200x800x500(pkg, gpu, tmp_path):
    check_config(pkg, gpu, "C2", tmp_path, None)


def test_c3_book2_final_800x800x1000(pkg, gpu, tmp_path, earth):
    check_config(pkg, gpu, "C3", tmp_path, earth)


def test_c3_lit_twin_800x800x1000(pkg, gpu, tmp_path, earth):
    """Config 3 with light in it: final_scene with its light wrapped in FlipFace (main.rs:359-361 does that for Cornell). The literal frame
    above is nearly black under HEAD's DiffuseLight (material.rs:184-190), so THIS frame is where the earth image, the Perlin sphere, the
    fog and the subsurface sphere are compared at values that carry signal."""
    check_config(pkg, gpu, "C3lit", tmp_path, earth)


def test_cornell_smoke_lit_twin_600x600x1000(pkg, gpu, tmp_path):
    check_config(pkg, gpu, "SMOKElit", tmp_path, None)


def test_c4_cornell_600x600x1000(pkg, gpu, tmp_path):
    check_config(pkg, gpu, "C4", tmp_path, None)


def test_c5_million_spheres_and_obj_mesh_4096(pkg, gpu, tmp_path):
    """BASELINE config 5 at its stated scene and frame size: 1 M spheres + a 131 072-triangle mesh imported from an OBJ file,
    4096x4096 (4 spp instead of 2048: 67 M camera paths), BVH in HBM. Then the 8-GPU partition on one GPU: all 8 shards rendered one
    after the other and put together by rt_untile = the unsharded frame, bit for bit; and the library's SAH tree gives the same
    picture up to hits that tie within rounding."""
    from importlib import import_module
    D = import_module("ray_tracer_archive_amd.distributed")
    cfg = K.CONFIGS["C5"]
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    hs, scene, cam, img = check_config(pkg, gpu, "C5", tmp_path, None)
    info = pkg.compile_info(hs.desc)
    assert info["n_spheres"] == 1000000 and info["n_tris"] == 131072 and info["fits_lds"] == 0
    base = pkg.make_params(W, H, spp, max_depth=50, seed=cfg["seed"])
    world = 8
    n = D.shard_floats(base, world)
    gathered = np.zeros((world, n), dtype=np.float32)
    samples = 0
    for r in range(world):
        buf, st = gpu.render(scene, cam, D.shard_params(base, r, world))
        gathered[r, :len(buf)] = buf
        samples += st["samples"]
    assert samples == W * H * spp
    assert np.array_equal(D.assemble(base, gathered, world), img)
    del gathered
    # SAH tree over the same primitives
    hs2 = K.host_scene(pkg, "C5", tmp_path, sah=True)
    img2, st2 = gpu.render(gpu.upload(hs2.desc), cam, base)
    differ = float((np.abs(img - img2).max(axis=2) > 0).mean())
    record(config="C5", crop="sah_vs_reference_tree_pixels_differ", value=differ)
    assert differ < 2e-3
