"""GPU parity on the other rows of SURVEY §8(a): rects / Box / Translate / RotateY / FlipFace / DiffuseLight /
MixturePdf (Cornell), ConstantMedium + Isotropic (cornell_smoke, final), MovingSphere + textures (book1_ref,
final), Triangle. Same seed, GPU (f32) against the f64 oracle. Tolerances (defaults of check()): mean |diff| <= 1.2e-4, pixels off by
more than 2e-3 <= 1.5 %, segments within 0.2 % — about twice the worst scene measured on MI355X (book1_ref: 4.7e-5, 0.72 %, 1.3e-4;
book-2 final: segments 1.0e-3); the one outlier scene states its own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check(pkg, orc, gpu, desc, cam, W, H, SPP, mean_tol=1.2e-4, bad_tol=0.015, seg_tol=2e-3, **kw):
    prm = pkg.make_params(W, H, SPP, flags=pkg._abi.RT_FLAG_COUNTERS, **kw)
    img, st = gpu.render(gpu.upload(desc), cam, prm)
    ref, ost = orc.render(desc, cam, prm, precision=64, n_threads=8, count=True)
    d = np.abs(img.astype(np.float64) - ref) / SPP
    bad = float((d.max(axis=2) > 2e-3).mean())
    import inspect
    from conftest import record_metric
    record_metric(config="scenes", crop=inspect.stack()[1].function, mean=float(d.mean()), bad=bad, seg=abs(st["segments"] - ost["segments"]) / ost["segments"], spp=SPP)
    assert np.isfinite(img).all()
    assert d.mean() <= mean_tol and bad <= bad_tol, (d.mean(), bad)
    assert abs(st["segments"] - ost["segments"]) <= max(8, seg_tol * ost["segments"]), (st["segments"], ost["segments"])
    return img, ref, st, ost


def test_cornell_box_mixture_pdf(pkg, orc, gpu):
    hs = pkg.HostScene("cornell", 0)
    img, ref, st, ost = check(pkg, orc, gpu, hs.desc, hs.camera(1.0), 80, 80, 16)
    # brute-force list as in the reference: every segment tests the 12 rects + 1 sphere, plus the two lights'
    # pdf_value hits per diffuse bounce (~14 primitive tests per segment)
    per_seg = (st["prim_tests"][0] + st["prim_tests"][2]) / st["segments"]
    assert 13.0 <= per_seg <= 15.5
    assert st["node_tests"] == 0 and ost["nonfinite_samples"] == 0
    assert abs(img.mean() - ref.mean()) / ref.mean() < 2e-3


def test_cornell_converged_level(pkg, gpu):
    # SURVEY 8(d) sanity value: mean linear radiance of the Cornell image ~ (0.19, 0.17, 0.16)
    hs = pkg.HostScene("cornell", 0)
    img, _ = gpu.render(gpu.upload(hs.desc), hs.camera(1.0), pkg.make_params(120, 120, 64))
    m = img.reshape(-1, 3).mean(0) / 64
    assert np.allclose(m, (0.19, 0.17, 0.16), atol=0.03)


def test_cornell_smoke_media_in_transformed_boxes(pkg, orc, gpu):
    hs = pkg.HostScene("cornell_smoke", 0)
    check(pkg, orc, gpu, hs.desc, hs.camera(1.0), 80, 80, 16)


def test_book1_reference_variant_moving_spheres_checker(pkg, orc, gpu):
    hs = pkg.HostScene("book1_ref", 1)
    check(pkg, orc, gpu, hs.desc, hs.camera(1.5), 96, 64, 16)


def test_book2_final_scene(pkg, orc, gpu, earth):
    hs = pkg.HostScene("final", 1, image=earth)
    img, ref, st, ost = check(pkg, orc, gpu, hs.desc, hs.camera(1.0), 80, 80, 16)
    assert abs(st["prim_tests"][1] - st["segments"]) <= 8       # the one moving sphere is a list member: every segment probes it (when its walk begins)
    assert st["segments"] - 8 <= st["prim_tests"][4] < 1.9 * st["segments"]   # so does the fog; the other medium only where a ray meets its box
    # the list as the reference walks it: every member in front of every ray — the same picture, bit for bit, and the reference's counts
    prm = pkg.make_params(80, 80, 16, flags=pkg._abi.RT_FLAG_COUNTERS)
    scene = gpu.upload(hs.desc, pkg._abi.RT_LAYOUT_REFERENCE_COUNTERS)
    img_r, sr = gpu.render(scene, hs.camera(1.0), prm)
    assert np.array_equal(img, img_r) and sr["segments"] == st["segments"]
    assert abs(sr["prim_tests"][4] - 2 * sr["segments"]) <= 8 and abs(sr["prim_tests"][1] - sr["segments"]) <= 8
    assert sum(st["prim_tests"][:5]) < 0.8 * sum(sr["prim_tests"][:5])
    # same trees, same order (the tolerances of test_gpu_full_frames.py's C3: looser device boxes, leaves of several primitives)
    assert -1e-3 <= (sr["node_tests"] - ost["node_tests"]) / ost["node_tests"] <= 0.10
    assert abs(sum(sr["prim_tests"][:5]) - sum(ost["prim_tests"][:5])) <= 0.27 * sum(ost["prim_tests"][:5])


def test_textures_noise_image_checker(pkg, orc, gpu, earth):
    rng = np.random.default_rng(3)
    b = pkg.SceneBuilder(background=(0.8, 0.8, 0.8))
    ids = [b.sphere((-2.2, 0, 0), 1.0, b.lambertian(texture=b.noise(4.0, rng))),
           b.sphere((0, 0, 0), 1.0, b.lambertian(texture=b.image(earth[::8, ::8]))),
           b.sphere((2.2, 0, 0), 1.0, b.lambertian(texture=b.checker((0.2, 0.3, 0.1), (0.9, 0.9, 0.9)))),
           b.sphere((0, 2.2, 0), 1.0, b.lambertian(texture=b.image(None))),          # empty image -> cyan (texture.rs:118-120)
           b.sphere((0, -1001, 0), 1000, b.lambertian(texture=b.noise(0.5, rng)))]
    cam = pkg.camera_new((0, 1, 9), (0, 0.3, 0), (0, 1, 0), 40, 1.5, 0.0, 9.0, 0, 0)
    check(pkg, orc, gpu, b.desc(b.bvh(ids)), cam, 96, 64, 8, max_depth=6)


def test_nested_checker_textures(pkg, orc, gpu, earth):
    """CheckerTexture::value recurses into `odd`/`even` (texture.rs:60-69): a checker whose children are a checker, an image and
    (two levels down) a noise texture. The sign of `sines` is a function of p alone, so every level takes the same branch."""
    rng = np.random.default_rng(7)
    b = pkg.SceneBuilder(background=(0.8, 0.8, 0.8))
    inner2 = b.checker_textures(b.noise(2.0, rng), b.solid_color((0.9, 0.1, 0.1)))
    inner1 = b.checker_textures(inner2, b.solid_color((0.1, 0.1, 0.9)))
    outer = b.checker_textures(inner1, b.image(earth[::8, ::8]))
    ids = [b.sphere((0, 0, 0), 1.5, b.lambertian(texture=outer)), b.sphere((3.2, 0, 0), 1.5, b.lambertian(texture=inner1)),
           b.sphere((0, -1001.5, 0), 1000, b.lambertian(texture=b.checker_textures(b.solid_color((0.3, 0.3, 0.3)), outer)))]
    cam = pkg.camera_new((1.5, 1.5, 9), (1.5, 0, 0), (0, 1, 0), 40, 1.5, 0.0, 9.0, 0, 0)
    check(pkg, orc, gpu, b.desc(b.bvh(ids)), cam, 96, 64, 8, max_depth=6)


@pytest.mark.parametrize("wrap", ["translate", "rotate", "both", "flip_both", "double_rotate"])
def test_instance_wrappers(pkg, orc, gpu, wrap):
    """Translate / RotateY / FlipFace chains, including the reference's RotateY quirk (hittable.rs:173 tests the
    child-space ray against the parent-space normal, which can flip the normal of a lone RotateY)."""
    b = pkg.SceneBuilder(background=(0.7, 0.8, 1.0))
    g, w, glass = b.lambertian((0.8, 0.3, 0.3)), b.lambertian((0.73,) * 3), b.dielectric(1.5)
    inner = b.hittable_list([b.box((-1, -1, -1), (1, 1.5, 1), w), b.sphere((2.5, 0, 0), 1.0, glass), b.sphere((-2.5, 0, 0.5), 0.9, g)])
    obj = {"translate": lambda: b.translate(inner, (1, 0.5, -2)),
           "rotate": lambda: b.rotate_y(inner, 30),
           "both": lambda: b.translate(b.rotate_y(inner, 30), (1, 0.5, -2)),
           "flip_both": lambda: b.flip_face(b.translate(b.rotate_y(inner, -40), (0, 0.5, -1))),
           "double_rotate": lambda: b.rotate_y(b.translate(b.rotate_y(inner, 20), (1, 0, 0)), 25)}[wrap]()
    world = b.hittable_list([obj, b.xz_rect(-20, 20, -20, 20, -1.5, g)])
    cam = pkg.camera_new((0, 3, 12), (0, 0, -1), (0, 1, 0), 40, 1.5, 0.0, 10.0, 0, 0)
    check(pkg, orc, gpu, b.desc(world), cam, 96, 64, 8)


def test_bvh_inside_an_instance(pkg, orc, gpu):
    # main.rs:640-646: Translate(RotateY(BVH(1000 overlapping spheres))) at coordinates ~500 — the case that
    # needs the "ray starts on this primitive" rule (an f32 false self-hit would trap the path inside a sphere)
    rng = np.random.default_rng(1)
    b = pkg.SceneBuilder(background=(0.7, 0.8, 1.0), bvh_seed=3)
    w = b.lambertian((0.73,) * 3)
    ids = [b.sphere(rng.uniform(0, 165, 3), 10, w) for _ in range(1000)]
    world = b.hittable_list([b.translate(b.rotate_y(b.bvh(ids, 0, 1), 15), (-100, 270, 395))])
    cam = pkg.camera_new((478, 278, -600), (278, 278, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
    # segments within 1.5 %: the f32 round trip through the instance transform still costs a few grazing re-hits of
    # NEIGHBOURING (overlapping) spheres; before the rule the device traced 2.25x the oracle's segments here
    check(pkg, orc, gpu, b.desc(world), cam, 64, 64, 4, mean_tol=4e-4, bad_tol=0.02, seg_tol=1.4e-2)   # measured 1.8e-4, 0.93 %, 6.8e-3


def test_fog_far_boundary(pkg, orc, gpu):
    # main.rs:590-599: r = 5000 boundary, density 1e-4, rays with |d| <= 1 (t ~ 5000: `rec1.t + 0.0001` vanishes in f32)
    b = pkg.SceneBuilder(background=(0.1, 0.1, 0.1))
    g = b.lambertian((0.48, 0.83, 0.53))
    world = b.hittable_list([b.box((-400, 0, -400), (400, 60, 400), g), b.constant_medium(b.sphere((0, 0, 0), 5000, b.dielectric(1.5)), 0.0001, (1, 1, 1))])
    cam = pkg.camera_new((478, 278, -600), (0, 30, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 1)
    img, ref, st, ost = check(pkg, orc, gpu, b.desc(world), cam, 64, 64, 8)
    assert st["segments"] > 1.3 * 64 * 64 * 8     # the fog does scatter


def test_triangles(pkg, orc, gpu):
    rng = np.random.default_rng(5)
    b = pkg.SceneBuilder(background=(0.7, 0.8, 1.0), bvh_seed=11)
    mats = [b.lambertian(rng.uniform(0.2, 0.9, 3)) for _ in range(5)] + [b.metal((0.8, 0.8, 0.7), 0.05), b.dielectric(1.5)]
    ids = []
    for i in range(120):
        c = rng.uniform(-4, 4, 3)
        v = [c + rng.normal(size=3) * 0.7 for _ in range(3)]
        ids.append(b.triangle(v[0], v[1], v[2], mats[i % len(mats)]))
    ids.append(b.xz_rect(-30, 30, -30, 30, -4.5, mats[0]))
    cam = pkg.camera_new((0, 1, 14), (0, 0, 0), (0, 1, 0), 40, 1.5, 0.0, 10.0, 0, 0)
    img, ref, st, ost = check(pkg, orc, gpu, b.desc(b.bvh(ids)), cam, 96, 64, 8)
    assert st["prim_tests"][3] > 0


def test_nan_policy_reference(pkg, orc, gpu):
    # RT_NAN_REFERENCE sums samples as they are (main.rs:146-155 scrubs only the final sum)
    hs = pkg.HostScene("cornell", 0)
    A = pkg._abi
    prm = pkg.make_params(48, 48, 8, nan_policy=A.RT_NAN_REFERENCE)
    img, _ = gpu.render(gpu.upload(hs.desc), hs.camera(1.0), prm)
    ref, ost = orc.render(hs.desc, hs.camera(1.0), prm, precision=64, n_threads=8)
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(img).all(axis=2)
    assert fin.mean() > 0.99
    assert np.mean(np.abs(img[fin] - ref[fin])) / 8 < 5e-4


def test_sah_builder_gives_the_same_picture(pkg, gpu):
    """RT_BVH_SAH only changes which boxes are tested: per-primitive arithmetic is unchanged,
    so the frame is bit-identical to the reference-shaped tree's while the walk visits far fewer nodes."""
    A = pkg._abi
    ref_s, sah_s = pkg.HostScene("book1", 1), pkg.HostScene("book1_sah", 1)
    cam = ref_s.camera(1.5)
    prm = pkg.make_params(160, 100, 8, flags=A.RT_FLAG_COUNTERS)
    a, sa = gpu.render(gpu.upload(ref_s.desc), cam, prm)
    b, sb = gpu.render(gpu.upload(sah_s.desc), cam, prm)
    assert np.array_equal(a, b)
    # ... and the reference's own tree (the ground sphere IN it; by default it is tested when a walk begins and left out): the same picture again
    c, sc = gpu.render(gpu.upload(ref_s.desc, A.RT_LAYOUT_LISTS_AS_REFERENCE), cam, prm)
    assert np.array_equal(a, c)
    assert sa["segments"] == sb["segments"] == sc["segments"]
    assert sb["node_tests"] < 0.8 * sa["node_tests"] < 0.8 * 0.85 * sc["node_tests"]


def test_config5_million_spheres_and_mesh(pkg, orc, gpu):
    """BASELINE config 5 at reduced size: 200 k spheres + a 65 k-triangle torus, BVH in HBM (not LDS-resident).
    Sub-pixel spheres make single samples chaotic in f32, so parity here is statistical: image means within 1 %
    and the near field (bottom rows, spheres several pixels wide) within the usual per-pixel tolerance."""
    hs = pkg.HostScene("big_sah", 5, 200000, 256)
    ref_scene = pkg.HostScene("big", 5, 200000, 256)
    cam = hs.camera(1.0)
    W = H = 96
    prm = pkg.make_params(W, H, 16, flags=pkg._abi.RT_FLAG_COUNTERS)
    img, st = gpu.render(gpu.upload(hs.desc), cam, prm)
    ref, ost = orc.render(ref_scene.desc, cam, prm, precision=64, n_threads=8, count=True)
    assert st["bvh_in_lds"] == 0 and st["prim_tests"][3] > 0
    assert abs(img.mean() - ref.mean()) / ref.mean() < 1e-2
    assert abs(st["segments"] - ost["segments"]) / ost["segments"] < 2e-2
    a8, b8 = pkg.tonemap(img, 16).astype(int), pkg.tonemap(ref.astype(np.float32), 16).astype(int)
    assert np.mean(np.abs(a8 - b8) <= 2) > 0.80
    # reference-shaped tree: the same frame except where two overlapping spheres' surfaces meet within rounding
    # (the culling test uses the running closest hit, so which of two hits ~1e-5 apart survives can depend on order)
    img2, _ = gpu.render(gpu.upload(ref_scene.desc), cam, prm)
    assert (np.abs(img - img2).max(axis=2) > 0).mean() < 2e-3
    # one record array in the builder's child order against the default, one array per direction octant ordered near-first
    # (rt_api.cpp octant_order): the same picture up to such ties, with far fewer visits
    one_order = gpu.upload(hs.desc, pkg._abi.RT_LAYOUT_REFERENCE_COUNTERS)
    img3, st3 = gpu.render(one_order, cam, prm)
    assert (np.abs(img - img3).max(axis=2) > 0).mean() < 2e-3 and abs(st3["segments"] - st["segments"]) <= 1e-4 * st["segments"]
    assert st["node_tests"] < 0.85 * st3["node_tests"] and sum(st["prim_tests"][:5]) < 0.85 * sum(st3["prim_tests"][:5])
    # the 8-wide tree walked 8 lanes to a ray (RT_LAYOUT_WIDE_NODES, kernels.hip k_extend_wide: one 128-byte node per visit, nearest child
    # first, the other hit children on a stack in LDS): the same picture up to such ties, on the SAH tree and on the reference-shaped one,
    # far fewer node visits (each counts the child boxes it tests: <= 8); a tile rendered alone equals the full frame bit for bit
    for h, base in ((hs, img), (ref_scene, img2)):
        wide = gpu.upload(h.desc, pkg._abi.RT_LAYOUT_WIDE_NODES)
        img4, st4 = gpu.render(wide, cam, prm)
        assert (np.abs(base - img4).max(axis=2) > 0).mean() < 2e-3 and abs(st4["segments"] - st["segments"]) <= 2e-4 * st["segments"]
        assert st4["node_tests"] < 8 * 0.5 * st3["node_tests"]          # fewer than half as many visits as the one-order binary walk makes box tests
        tile, _ = gpu.render(wide, cam, pkg.make_params(W, H, 16, tile_size=32, shard_index=4, shard_count=9))
        assert np.array_equal(tile.reshape(32, 32, 3), img4[32:64, 32:64])
        again, _ = gpu.render(wide, cam, pkg.make_params(W, H, 16, pool_slots=8192))
        assert np.array_equal(again, img4)                                # independent of the pool size (no hand-over to the per-path kernel)


def test_imported_obj_mesh(pkg, orc, gpu, tmp_path):
    from test_host import CUBE_OBJ
    path = tmp_path / "cube.obj"
    path.write_text(CUBE_OBJ)
    hs = pkg.HostScene("obj:" + str(path), 1)
    check(pkg, orc, gpu, hs.desc, hs.camera(1.5), 96, 64, 8)


def test_time_survives_a_glass_bounce(pkg, orc, gpu):
    """A ray keeps its time through a Dielectric bounce (material.rs:131-155 builds the scattered ray with r_in.time()) and meets
    MovingSpheres where they are at THAT time. Regression: the F_ALL instance of k_shade once left the Schlick draw in the time
    slot, so rays reflected off glass saw moving spheres elsewhere (one wrong sample in a few thousand — inside every statistical
    tolerance; it was caught by comparing the wavefront kernels with the fused per-path kernel, which shares their code but was
    compiled right). Checked three ways: against the f64 oracle, wavefront == fused bit for bit, and with time-frozen cameras."""
    import os
    A = pkg._abi
    rng = np.random.default_rng(1)
    b = pkg.SceneBuilder(background=(0.7, 0.8, 1.0), background_mode=A.RT_BG_SKY_GRADIENT)
    lam, glass = b.lambertian((0.5, 0.5, 0.5)), b.dielectric(1.5)
    ids = []
    for i in range(24):
        c = np.array([rng.uniform(-2.5, 2.5), 0.3, rng.uniform(-2.5, 2.5)])
        ids.append(b.moving_sphere(c, c + (0, rng.uniform(0.1, 0.5), 0), 0.0, 1.0, 0.3, lam) if i % 2 else b.sphere(c, 0.3, glass))
    desc = b.desc(b.bvh(ids, 0.0, 1.0))
    scene = gpu.upload(desc)
    for t1 in (0.0, 1.0):
        cam = pkg.camera_new((5, 2, 2), (0, 0.3, 0), (0, 1, 0), 40, 1.5, 0.0, 10.0, 0.0, t1)
        for depth in (2, 50):
            w, sw = gpu.render(scene, cam, pkg.make_params(96, 64, 8, seed=3, max_depth=depth, flags=A.RT_FLAG_COUNTERS, tail_paths=1))   # 1 = never hand over
            f, sf = gpu.render(scene, cam, pkg.make_params(96, 64, 8, seed=3, max_depth=depth, flags=A.RT_FLAG_COUNTERS | A.RT_FLAG_FUSED))
            assert np.array_equal(w, f) and sw["node_tests"] == sf["node_tests"] and sw["prim_tests"] == sf["prim_tests"], (t1, depth)
    cam = pkg.camera_new((5, 2, 2), (0, 0.3, 0), (0, 1, 0), 40, 1.5, 0.0, 10.0, 0.0, 1.0)
    check(pkg, orc, gpu, desc, cam, 96, 64, 16)


def test_list_culling_and_prologue_change_nothing_but_the_counters(pkg, orc, gpu):
    """A root list with one of everything next to a BVH (so that the compiler culls its members, scene_compile.cpp emit_list_culled):
    a hollow glass sphere (negative radius), a moving sphere (prologue), a fog that holds the whole scene (prologue), a small medium in
    a rotated box, an instanced box, loose rects, triangles and a nested list. The culled layout, the layout forced for every scene
    (RT_LIST_CULL=2) and the reference's layout (every member in front of every ray) render the same frame bit for bit; the oracle agrees
    within the usual tolerances; the culled walk makes fewer primitive tests."""
    rng = np.random.default_rng(11)
    b = pkg.SceneBuilder(background=(0.55, 0.65, 0.9))
    grey, red, glass, steel = b.lambertian((0.6, 0.6, 0.6)), b.lambertian((0.7, 0.2, 0.15)), b.dielectric(1.5), b.metal((0.8, 0.8, 0.9), 0.05)
    cloud = [b.sphere((float(rng.uniform(-6, 6)), float(rng.uniform(0.2, 0.5)), float(rng.uniform(-6, 3))), float(rng.uniform(0.15, 0.35)),
                      [grey, red, steel][i % 3]) for i in range(48)]
    members = [
        b.bvh(cloud),
        b.sphere((0, -1000, 0), 1000, grey),
        b.sphere((-2.5, 1.0, 0), 1.0, glass), b.sphere((-2.5, 1.0, 0), -0.9, glass),
        b.moving_sphere((2.5, 1.0, -1), (2.5, 1.5, -1), 0.0, 1.0, 0.6, red),
        b.constant_medium(b.sphere((0, 0, 0), 60, glass), 0.01, (1, 1, 1)),
        b.constant_medium(b.translate(b.rotate_y(b.box((0, 0, 0), (1.2, 1.2, 1.2), grey), 25), (0.5, 0.0, 1.5)), 0.8, (0.1, 0.1, 0.1)),
        b.translate(b.rotate_y(b.box((0, 0, 0), (1, 2, 1), steel), -18), (3.5, 0, -3)),
        b.xz_rect(-1, 1, -4, -2, 3.5, b.diffuse_light((6, 6, 6))),
        b.xy_rect(-6, -4, 0, 2, -5, red), b.yz_rect(0, 2, -5, -3, 6, grey),
        b.triangle((-1, 0.01, 3), (1, 0.01, 3), (0, 1.5, 2.5), red), b.triangle((4, 0, 2), (5, 0, 2), (4.5, 1, 2.2), steel),
        b.hittable_list([b.sphere((5, 0.5, 1), 0.5, steel), b.sphere((5, 1.4, 1), 0.4, red), b.box((-5, 0, 1), (-4, 1, 2), grey)]),
    ]
    desc = b.desc(b.hittable_list(members))
    cam = pkg.camera_new((0, 3, 13), (0, 0.8, 0), (0, 1, 0), 40, 1.5, 0.0, 13.0, 0, 1)
    W, H, SPP = 150, 100, 16
    img, ref, st, ost = check(pkg, orc, gpu, desc, cam, W, H, SPP, bad_tol=0.03, seg_tol=3e-3)
    prm = pkg.make_params(W, H, SPP, flags=pkg._abi.RT_FLAG_COUNTERS)
    plain = gpu.upload(desc, pkg._abi.RT_LAYOUT_REFERENCE_COUNTERS)
    img_p, sp = gpu.render(plain, cam, prm)
    forced = gpu.upload(desc, pkg._abi.RT_LAYOUT_LISTS_CULLED)
    img_f, sf = gpu.render(forced, cam, prm)
    assert np.array_equal(img, img_p) and np.array_equal(img, img_f)
    assert st["segments"] == sp["segments"] == sf["segments"] and st["prim_tests"] == sf["prim_tests"]
    assert abs(sp["prim_tests"][1] - sp["segments"]) <= 8 and abs(st["prim_tests"][1] - st["segments"]) <= 8     # the moving sphere: every segment, either way
    assert sp["prim_tests"][4] >= 2 * sp["segments"] - 8 and st["prim_tests"][4] < 1.5 * st["segments"]            # both media for every ray / the small one culled
    assert sum(st["prim_tests"][:5]) < 0.6 * sum(sp["prim_tests"][:5])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_graphs_render_the_same_under_every_layout(pkg, gpu, seed):
    """Random object graphs — lists in lists, BVHs in lists, instanced boxes and spheres, media with sphere and box boundaries, moving and
    hollow spheres, loose rects and triangles, in random order — rendered under the layouts the uploader can choose: list members culled
    and every-ray members in the prologue (forced), pair members boxed (forced), versus the reference's shape. Same frame, bit for bit."""
    rng = np.random.default_rng(100 + seed)
    b = pkg.SceneBuilder(background=(0.6, 0.7, 0.95))
    mats = [b.lambertian(tuple(rng.uniform(0.2, 0.8, 3))) for _ in range(4)] + [b.metal((0.8, 0.8, 0.8), float(rng.uniform(0, 0.3))), b.dielectric(1.5)]
    U = lambda lo, hi, n=None: rng.uniform(lo, hi, n)

    def rand_prim():
        k = int(rng.integers(0, 7))
        c = U(-5, 5, 3); c[1] = abs(c[1]) * 0.4
        m = mats[int(rng.integers(0, len(mats)))]
        if k == 0: return b.sphere(tuple(c), float(U(0.2, 0.8)), m)
        if k == 1: return b.moving_sphere(tuple(c), tuple(c + (0, float(U(0.1, 0.6)), 0)), 0.0, 1.0, float(U(0.2, 0.5)), mats[int(rng.integers(0, 4))])
        if k == 2: return b.box(tuple(c), tuple(c + U(0.3, 1.5, 3)), m)
        if k == 3: return b.translate(b.rotate_y(b.box((0, 0, 0), tuple(U(0.4, 1.4, 3)), m), float(U(-60, 60))), tuple(c))
        if k == 4: return b.xz_rect(c[0], c[0] + 1.5, c[2], c[2] + 1.5, float(U(0.05, 2.5)), m)
        if k == 5: return b.triangle(tuple(c), tuple(c + (1, 0, 0.2)), tuple(c + (0.3, 1, 0)), m)
        return b.translate(b.sphere((0, 0, 0), float(U(0.2, 0.6)), m), tuple(c))

    def rand_group(depth):
        n = int(rng.integers(2, 7))
        kids = []
        for _ in range(n):
            r = rng.random()
            if depth < 2 and r < 0.2: kids.append(rand_group(depth + 1))
            elif r < 0.3: kids.append(b.constant_medium(b.sphere(tuple(U(-4, 4, 3)), float(U(0.5, 1.2)), mats[5]), float(U(0.2, 1.5)), tuple(U(0.1, 0.9, 3))))
            elif r < 0.36: kids.append(b.constant_medium(b.translate(b.box((0, 0, 0), (1, 1, 1), mats[0]), tuple(U(-4, 4, 3))), float(U(0.3, 1.0)), (0.9, 0.9, 0.9)))
            else: kids.append(rand_prim())
        return b.bvh(kids, 0.0, 1.0) if rng.random() < 0.4 else b.hittable_list(kids)

    top = [b.sphere((0, -1000, 0), 1000, mats[0]), b.bvh([b.sphere(tuple(U(-6, 6, 3) * (1, 0.1, 1) + (0, 0.3, 0)), 0.25, mats[int(rng.integers(0, 6))]) for _ in range(40)]),
           b.sphere((2, 1, 0), 1.0, mats[5]), b.sphere((2, 1, 0), -0.85, mats[5])]
    top += [rand_group(0) for _ in range(3)] + [rand_prim() for _ in range(4)]
    if seed % 2: top.append(b.constant_medium(b.sphere((0, 0, 0), 80, mats[5]), 0.005, (1, 1, 1)))
    order = rng.permutation(len(top))
    desc = b.desc(b.hittable_list([top[i] for i in order]))
    cam = pkg.camera_new((0, 3, 14), (0, 0.8, 0), (0, 1, 0), 40, 1.5, 0.05, 14.0, 0, 1)
    prm = pkg.make_params(96, 64, 4, seed=seed, flags=pkg._abi.RT_FLAG_COUNTERS)
    plain = gpu.upload(desc, pkg._abi.RT_LAYOUT_REFERENCE_COUNTERS)
    ref, sr = gpu.render(plain, cam, prm)
    assert np.isfinite(ref).all() and sr["segments"] > 96 * 64 * 4
    forced = gpu.upload(desc, pkg._abi.RT_LAYOUT_LISTS_CULLED | pkg._abi.RT_LAYOUT_MEMBER_BOXES)
    img, st = gpu.render(forced, cam, prm)
    assert np.array_equal(img, ref) and st["segments"] == sr["segments"], seed
    img_d, sd = gpu.render(gpu.upload(desc), cam, prm)          # what the uploader picks by itself
    assert np.array_equal(img_d, ref) and sd["segments"] == sr["segments"], seed
    assert sum(st["prim_tests"][:5]) < sum(sr["prim_tests"][:5])


@pytest.mark.parametrize("ground", ["rect", "sphere", "two_spheres"])
def test_scene_sized_primitive_tested_first(pkg, orc, gpu, ground):
    """A rect or sphere of the root BVH as large as the scene is tested before the walk and kept out of the tree (where nothing moves and it
    is the only one: where the ray is made). Same frame, bit for bit, as with it IN the tree (RT_LAYOUT_LISTS_AS_REFERENCE), as a counting
    render (tested in the walk's first pass), and on the SAH tree — the spheres float above the ground, so no two hits tie — and the oracle's."""
    A = pkg._abi
    frames, visits = [], []
    for builder in (A.RT_BVH_REFERENCE, A.RT_BVH_SAH):
        b = pkg.SceneBuilder(background=(0.7, 0.8, 1.0), bvh_seed=3, bvh_builder=builder)
        grey, glass, metal = b.lambertian((0.5, 0.5, 0.5)), b.dielectric(1.5), b.metal((0.8, 0.6, 0.2), 0.1)
        r = np.random.default_rng(4)
        ids = []
        for k in range(80):
            rad = r.uniform(0.2, 0.6)
            ids.append(b.sphere((r.uniform(-8, 8), rad + 0.05 + r.uniform(0, 0.5), r.uniform(-8, 8)), rad, [b.lambertian(tuple(r.uniform(0.1, 0.9, 3))), glass, metal][k % 3]))
        if ground == "rect":
            ids.append(b.xz_rect(-60, 60, -60, 60, 0.0, grey))
        else:
            ids.append(b.sphere((0, -1000, 0), 1000, grey))
            if ground == "two_spheres":
                ids.append(b.sphere((0, 0, -1030), 1000, metal))          # a second scene-sized sphere: both go first, tested in the walk's first pass
        desc = b.desc(b.bvh(ids))
        info = pkg.compile_info(desc)
        assert len(info["first"]) == (1 if not (builder == A.RT_BVH_SAH and ground == "two_spheres") else 0)
        if info["first"]:
            assert (info["first"][0] >> 28) == (3 if ground == "rect" else 1) and ((info["first"][0] >> 24) & 15) == (2 if ground == "two_spheres" else 1)
        cam = pkg.camera_new((10, 3, 12), (0, 0.5, 0), (0, 1, 0), 35, 4 / 3, 0.05, 15.0, 0, 0)
        prm = pkg.make_params(128, 96, 8, seed=2)
        img, _ = gpu.render(gpu.upload(desc), cam, prm)
        cnt, st = gpu.render(gpu.upload(desc), cam, pkg.make_params(128, 96, 8, seed=2, flags=A.RT_FLAG_COUNTERS))
        ref, st_ref = gpu.render(gpu.upload(desc, A.RT_LAYOUT_LISTS_AS_REFERENCE), cam, pkg.make_params(128, 96, 8, seed=2, flags=A.RT_FLAG_COUNTERS))
        assert np.array_equal(img, cnt) and np.array_equal(img, ref)
        if info["first"]:
            assert st["node_tests"] < st_ref["node_tests"]
        frames.append(img); visits.append(st["node_tests"])
        if builder == A.RT_BVH_REFERENCE:
            o, _ = orc.render(desc, cam, prm, precision=64, n_threads=8)
            d = np.abs(img.astype(np.float64) - o) / 8
            assert d.mean() < 4e-5 and float((d.max(axis=2) > 2e-3).mean()) < 0.01
    assert np.array_equal(frames[0], frames[1])
