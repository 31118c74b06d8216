"""Known-answer tests that pin the CPU oracle to the reference SOURCE.

The reference has no tests, no golden vectors and cannot be run here, so each expected value below
is derived by hand from the cited reference lines."""
import ctypes as C
import math

import numpy as np
import pytest


def d(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


def test_write_color(orc):
    # main.rs:141-169: sqrt(sum/spp), clamp to [0, 0.999], *256, truncate; NaN on the SUM -> 0
    assert orc.write_color((25.0, 25.0, 25.0), 100) == (128, 128, 128)      # sqrt(0.25) = 0.5 -> 128
    assert orc.write_color((100.0, 250.0, 1e9), 100) == (255, 255, 255)     # clamp 0.999*256 = 255.74
    assert orc.write_color((float("nan"), 0.0, 1.0), 100) == (0, 0, 25)     # sqrt(0.01)=0.1 -> 25.6
    assert orc.write_color((float("inf"), -1.0, 0.0), 10) == (255, 0, 0)    # inf survives the scrub; sqrt(-0.1) = NaN -> (NaN as u8) = 0


def test_write_color_matches_host_mirror(orc, pkg):
    rng = np.random.default_rng(0)
    for _ in range(200):
        c = rng.uniform(0, 300, 3)
        assert orc.write_color(c, 200) == pkg.write_color(c, 200)


def test_cornell_camera(orc, pkg):
    # Camera::new with main.rs:688-718: lookfrom (278,278,-800), lookat (278,278,0), vfov 40, aspect 1, aperture 0, focus 10
    cam = pkg._abi.RtCamera()
    orc.lib().orc_camera_new(d((278, 278, -800)), d((278, 278, 0)), d((0, 1, 0)), d((40.0, 1.0, 0.0, 10.0)), 0.0, 1.0, C.byref(cam))
    h = math.tan(math.radians(20.0))
    assert cam.w.tuple() == pytest.approx((0, 0, -1))
    assert cam.u.tuple() == pytest.approx((-1, 0, 0))
    assert cam.v.tuple() == pytest.approx((0, 1, 0))
    assert h == pytest.approx(0.363970, abs=1e-6)
    assert cam.horizontal.tuple() == pytest.approx((-20 * h, 0, 0))
    assert cam.vertical.tuple() == pytest.approx((0, 20 * h, 0))
    assert cam.lower_left_corner.tuple() == pytest.approx((278 + 10 * h, 278 - 10 * h, -790))
    assert cam.lower_left_corner.tuple() == pytest.approx((281.63970, 274.36030, -790.0), abs=1e-5)
    assert cam.lens_radius == 0.0 and cam.time0 == 0.0 and cam.time1 == 1.0
    # the host mirror is the product's Camera::new: field-for-field equal
    cam2 = pkg.camera_new((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    for f in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
        assert getattr(cam, f).tuple() == getattr(cam2, f).tuple()


def test_sphere_hit(orc):
    # sphere.rs:41-65 with ray (0,0,0)->(0,0,-1), c=(0,0,-1), r=0.5: t=0.5, p=(0,0,-0.5), n=(0,0,1), front face;
    # get_sphere_uv (sphere.rs:32-37) of n: theta=acos(0)=pi/2, phi=atan2(-1,0)+pi=pi/2 -> u=0.25, v=0.5
    h = orc.sphere_hit((0, 0, -1), 0.5, (0, 0, 0), (0, 0, -1))
    assert h["t"] == pytest.approx(0.5) and h["p"] == pytest.approx((0, 0, -0.5)) and h["normal"] == pytest.approx((0, 0, 1))
    assert h["front_face"] and h["u"] == pytest.approx(0.25) and h["v"] == pytest.approx(0.5)
    # from inside: first root negative -> second root, normal flipped against the ray, front_face false
    h = orc.sphere_hit((0, 0, 0), 2.0, (0, 0, 0), (0, 0, -3))
    assert h["t"] == pytest.approx(2.0 / 3.0) and h["normal"] == pytest.approx((0, 0, 1)) and not h["front_face"]
    # inclusive bounds (sphere.rs:52): a root exactly at t_max is accepted, just beyond is not
    assert orc.sphere_hit((0, 0, -1), 0.5, (0, 0, 0), (0, 0, -1), t_max=0.5) is not None
    assert orc.sphere_hit((0, 0, -1), 0.5, (0, 0, 0), (0, 0, -1), t_max=0.4999999) is None
    # direction is not normalised: t scales with 1/|d|
    assert orc.sphere_hit((0, 0, -1), 0.5, (0, 0, 0), (0, 0, -10))["t"] == pytest.approx(0.05)


def test_rect_hit_and_pdf(orc):
    # aarect.rs:81-98 XzRect; normal +y flipped against the ray
    h = orc.rect_hit(1, (0, 2, 0, 4, 1.0), (1, 0, 1), (0, 2, 0))
    assert h["t"] == pytest.approx(0.5) and h["u"] == pytest.approx(0.5) and h["v"] == pytest.approx(0.25)
    assert h["normal"] == pytest.approx((0, -1, 0)) and not h["front_face"]
    assert orc.rect_hit(1, (0, 2, 0, 4, 1.0), (3, 0, 1), (0, 2, 0)) is None
    # edges are inclusive (aarect.rs:88)
    assert orc.rect_hit(1, (0, 2, 0, 4, 1.0), (2, 0, 4), (0, 1, 0)) is not None
    # XzRect::pdf_value (aarect.rs:107-117), Cornell light from o=(278,0,279.5), v=(0,554,0):
    # area 130*105 = 13650, t=1, dist^2 = 554^2 = 306916, cos = 1 -> 22.4847
    pdf = orc.lib().orc_xzrect_pdf_value(d((213, 343, 227, 332, 554)), d((278, 0, 279.5)), d((0, 554, 0)))
    assert pdf == pytest.approx(306916.0 / 13650.0) and pdf == pytest.approx(22.4847, abs=1e-4)
    assert orc.lib().orc_xzrect_pdf_value(d((213, 343, 227, 332, 554)), d((278, 0, 279.5)), d((554, 0, 0))) == 0.0


def test_dielectric_terms(orc):
    # material.rs:123-127 reflectance(1, 1.5) = r0 = ((1-1.5)/(1+1.5))^2 = 0.04
    assert orc.lib().orc_reflectance(1.0, 1.5) == pytest.approx(0.04)
    assert orc.lib().orc_reflectance(0.0, 1.5) == pytest.approx(1.0)
    out = (C.c_double * 3)()
    orc.lib().orc_refract(d((0, 0, -1)), d((0, 0, 1)), 1 / 1.5, out)          # vec3.rs:246-251: normal incidence goes straight
    assert tuple(out) == pytest.approx((0, 0, -1))
    s = math.sin(math.radians(30))
    orc.lib().orc_refract(d((s, 0, -math.cos(math.radians(30)))), d((0, 0, 1)), 1 / 1.5, out)   # Snell: sin t = sin i / 1.5
    assert out[0] == pytest.approx(s / 1.5) and math.hypot(*out[:3:2]) == pytest.approx(1.0)
    orc.lib().orc_reflect(d((1, -1, 0)), d((0, 1, 0)), out)                   # vec3.rs:115-117
    assert tuple(out) == pytest.approx((1, 1, 0))


def test_onb(orc):
    # onb.rs:19-30 build_from_w((0,1,0)): w=(0,1,0); |w.x| <= 0.9 -> a=(1,0,0); v = unit(cross(w,a)) = (0,0,-1); u = cross(w,v) = (-1,0,0)
    out = (C.c_double * 9)()
    orc.lib().orc_onb_build_from_w(d((0, 1, 0)), out)
    assert tuple(out[0:3]) == pytest.approx((-1, 0, 0)) and tuple(out[3:6]) == pytest.approx((0, 0, -1)) and tuple(out[6:9]) == pytest.approx((0, 1, 0))
    orc.lib().orc_onb_build_from_w(d((5, 0, 0)), out)                         # |w.x| > 0.9 -> a=(0,1,0)
    assert tuple(out[6:9]) == pytest.approx((1, 0, 0)) and tuple(out[3:6]) == pytest.approx((0, 0, 1))


def test_aabb_quirk_f7(orc):
    # aabb.rs:31-55 shadows t_min/t_max per axis (F7): a ray that passes each slab inside [t_min,t_max] but at
    # DISJOINT t-intervals is accepted by the literal code and rejected by the carried-interval test.
    mn, mx, o, dd = (0, 0, 0), (1, 1, 1), (-1, 2.5, 0.5), (1, -1, 0)   # x-slab t in [1,2], y-slab t in [1.5,2.5]... overlapping
    assert orc.lib().orc_aabb_hit(d(mn), d(mx), d(o), d(dd), 0.001, 100.0, 0) == 1
    o2 = (-1, 4.5, 0.5)                                               # x-slab [1,2], y-slab [3.5,4.5]: disjoint
    assert orc.lib().orc_aabb_hit(d(mn), d(mx), d(o2), d(dd), 0.001, 100.0, 1) == 1   # reference quirk: "hit"
    assert orc.lib().orc_aabb_hit(d(mn), d(mx), d(o2), d(dd), 0.001, 100.0, 0) == 0   # intended slab test: miss
    # axis-parallel ray on the boundary plane: 0*inf = NaN compares false, the box is kept (conservative)
    assert orc.lib().orc_aabb_hit(d(mn), d(mx), d((0.0, 0.5, -1)), d((0, 0, 1)), 0.001, 100.0, 0) == 1


def test_wrappers_force_front_face(orc, pkg):
    # hittable.rs:82-83,173: Translate/RotateY call set_face_normal with the already-flipped normal -> front_face = true
    b = pkg.SceneBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    inner = b.sphere((0, 0, 0), 1.0, m)
    plain = b.hittable_list([inner])
    hit = orc.world_hit(b.desc(plain), (0, 0, 0), (0, 0, 1))          # from inside: back face
    assert hit["t"] == pytest.approx(1.0) and not hit["front_face"]
    b2 = pkg.SceneBuilder()
    m2 = b2.lambertian((0.5, 0.5, 0.5))
    w = b2.translate(b2.rotate_y(b2.sphere((0, 0, 0), 1.0, m2), 30.0), (5, 0, 0))
    hit = orc.world_hit(b2.desc(b2.hittable_list([w])), (5, 0, 0), (0, 0, 1))
    assert hit["t"] == pytest.approx(1.0) and hit["front_face"] and hit["p"] == pytest.approx((5, 0, 1))
    assert hit["normal"] == pytest.approx((0, 0, -1))
    # FlipFace flips the flag only, never the normal (hittable.rs:199)
    b3 = pkg.SceneBuilder()
    f = b3.flip_face(b3.xz_rect(-1, 1, -1, 1, 2.0, b3.diffuse_light((1, 1, 1))))
    hit = orc.world_hit(b3.desc(b3.hittable_list([f])), (0, 0, 0), (0, 1, 0))
    assert hit["normal"] == pytest.approx((0, -1, 0)) and hit["front_face"] is True   # rect alone: back face (false) -> flipped to true


def test_empty_world_is_background(orc, pkg):
    # KAT 9: nothing to hit -> every sample returns the background (main.rs:74-76)
    b = pkg.SceneBuilder(background=(0.25, 0.5, 1.0))
    m = b.lambertian((0.5, 0.5, 0.5))
    far = b.sphere((0, 0, 1e6), 1.0, m)     # behind the camera
    desc = b.desc(b.hittable_list([far]))
    cam = pkg.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), 40, 2.0, 0.0, 1.0, 0, 0)
    prm = pkg.make_params(16, 8, 4)
    img, st = orc.render(desc, cam, prm)
    assert np.allclose(img, np.array([0.25, 0.5, 1.0]) * 4)
    assert orc.write_color(img[0, 0], 4) == (128, 181, 255)


def test_bvh_equals_list(orc, pkg):
    # KAT 10: a BVH is a pure accelerator — closest hit equals the brute-force list's
    rng = np.random.default_rng(5)
    b = pkg.SceneBuilder(bvh_seed=7)
    m = b.lambertian((0.5, 0.5, 0.5))
    ids = [b.sphere(rng.uniform(-10, 10, 3), rng.uniform(0.2, 1.5), m) for _ in range(200)]
    lst, bvh = b.hittable_list(ids), b.bvh(ids)
    dl, db = b.desc(lst), b.desc(bvh)
    for _ in range(300):
        o, dd = rng.uniform(-12, 12, 3), rng.normal(size=3)
        a, c = orc.world_hit(dl, o, dd), orc.world_hit(db, o, dd)
        assert (a is None) == (c is None)
        if a is not None:
            assert a["t"] == pytest.approx(c["t"], rel=1e-12) and a["front_face"] == c["front_face"]


def test_rng_contract(orc):
    # the counter RNG that replaces rand::random: SplitMix64 finaliser; f64 = top 53 bits, f32 = top 24 bits
    raw, f64, f32 = orc.rng_stream(1, 0, 0, 8)
    assert len(set(raw.tolist())) == 8
    assert np.all((f64 >= 0) & (f64 < 1)) and np.all((f32 >= 0) & (f32 < 1))
    assert np.array_equal(f64, (raw >> np.uint64(11)).astype(np.float64) / 2.0 ** 53)
    assert np.array_equal(f32, ((raw >> np.uint64(40)).astype(np.float32) / np.float32(2.0 ** 24)))
    assert np.max(np.abs(f64 - f32.astype(np.float64))) < 2.0 ** -24
    # independent python restatement of the generator
    M = (1 << 64) - 1

    def fin(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    G = 0x9E3779B97F4A7C15
    for seed, pix, smp in [(1, 0, 0), (7, 123456, 499), (2 ** 63 + 5, 4095 * 4096 + 17, 2047)]:
        h = fin((seed + G * (pix + 1)) & M)
        s = fin((h + 0xD1B54A32D192ED03 * (smp + 1)) & M)
        exp = []
        for _ in range(5):
            s = (s + G) & M
            exp.append(fin(s))
        got, _, _ = orc.rng_stream(seed, pix, smp, 5)
        assert got.tolist() == exp


def test_rng_uniformity(orc):
    # first draws of consecutive pixels / samples look uniform and uncorrelated
    xs = np.array([orc.rng_stream(1, p, s, 2)[1] for p in range(64) for s in range(32)])
    assert abs(xs[:, 0].mean() - 0.5) < 0.02 and abs(xs[:, 1].mean() - 0.5) < 0.02
    assert abs(np.corrcoef(xs[:, 0], xs[:, 1])[0, 1]) < 0.06
    h, _ = np.histogram(xs[:, 0], bins=16, range=(0, 1))
    assert h.min() > 80 and h.max() < 180


def test_furnace(orc, pkg):
    # white furnace: albedo-1 Lambertian spheres under a constant background c: every path that escapes carries
    # exactly c (cosine sampling: scattering_pdf / pdf = 1, main.rs:130-138), so pixel mean = c * P(escape within depth)
    b = pkg.SceneBuilder(background=(0.5, 0.75, 1.0))
    m = b.lambertian((1.0, 1.0, 1.0))
    ids = [b.sphere((x * 2.2, 0, z * 2.2 - 6), 1.0, m) for x in (-1, 0, 1) for z in (-1, 0, 1)]
    desc = b.desc(b.bvh(ids))
    cam = pkg.camera_new((0, 4, 6), (0, 0, -6), (0, 1, 0), 40, 1.5, 0.0, 10.0, 0, 0)
    prm = pkg.make_params(48, 32, 16, max_depth=50)
    for prec in (64, 32):
        img, st = orc.render(desc, cam, prm, precision=prec, n_threads=4)
        mean = img.reshape(-1, 3).mean(0) / 16
        assert np.allclose(mean, (0.5, 0.75, 1.0), rtol=2e-3)
        assert st["nonfinite_samples"] == 0


def test_f32_vs_f64_converged(orc, pkg):
    # the oracle in device arithmetic (f32) and in reference arithmetic (f64) agree on the converged image
    hs = pkg.HostScene("book1", 1)
    cam = hs.camera(1.5)
    prm = pkg.make_params(48, 32, 64)
    a, _ = orc.render(hs.desc, cam, prm, precision=64, n_threads=8)
    b, _ = orc.render(hs.desc, cam, prm, precision=32, n_threads=8)
    assert abs(a.mean() - b.mean()) / a.mean() < 5e-3
    assert np.mean(np.abs(a - b)) / 64 < 5e-3


def test_oracle_reproduces_its_golden_fixture(orc, pkg):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "book1_64x40_8spp_f64.npy"))
    hs = pkg.HostScene("book1", 1)
    img, _ = orc.render(hs.desc, hs.camera(64 / 40), pkg.make_params(64, 40, 8, seed=1), precision=64, n_threads=3)
    assert np.array_equal(img, g)          # bit-exact, and independent of the thread count
