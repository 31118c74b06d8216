"""The scene compiler (graph -> threaded BVH), checked on the CPU against the oracle's object graph."""
import numpy as np
import pytest


def sphere_root(spheres, k, o, d, a, tmin, best):
    c, r = spheres[k, :3].astype(float), float(spheres[k, 3])
    oc = o - c
    hb, cc = oc @ d, oc @ oc - r * r
    det = hb * hb - a * cc
    if det < 0:
        return None
    sq = np.sqrt(det)
    root = (-hb - sq) / a
    if root < tmin or best < root:
        root = (-hb + sq) / a
        if root < tmin or best < root:
            return None
    return root


def traverse(nodes, spheres, o, d, tmin=0.001, first=()):
    """Python walk of the dumped threaded BVH (spheres only): returns (t, sphere index, node visits). `first`: spheres tested before the
    walk begins (RtCompileInfo.first)."""
    i, n = 0, len(nodes)
    best, hit, visits = np.inf, -1, 0
    o, d = np.asarray(o, float), np.asarray(d, float)
    a = d @ d
    for k in first:
        root = sphere_root(spheres, k, o, d, a, tmin, best)
        if root is not None:
            best, hit = root, k
    while i < n:
        nd = nodes[i]
        boxed = np.isfinite(nd["mn"][0])
        ok = True
        if boxed:
            visits += 1
            lo, hi = tmin, best
            with np.errstate(divide="ignore", invalid="ignore"):
                for ax in range(3):
                    inv = 1.0 / d[ax]
                    t0, t1 = (nd["mn"][ax] - o[ax]) * inv, (nd["mx"][ax] - o[ax]) * inv
                    if inv < 0:
                        t0, t1 = t1, t0
                    lo = lo if lo > t0 else t0
                    hi = hi if hi < t1 else t1
                    if hi <= lo:
                        ok = False
                        break
        if not ok:
            assert nd["skip"] > i
            i = int(nd["skip"])
            continue
        leaf = int(nd["leaf"])
        if leaf:
            typ, cnt, first = leaf >> 28, (leaf >> 24) & 15, leaf & 0xFFFFFF
            assert typ == 1
            for k in range(first, first + cnt):
                c, r = spheres[k, :3].astype(float), float(spheres[k, 3])
                oc = o - c
                hb, cc = oc @ d, oc @ oc - r * r
                det = hb * hb - a * cc
                if det >= 0:
                    sq = np.sqrt(det)
                    root = (-hb - sq) / a
                    if root < tmin or best < root:
                        root = (-hb + sq) / a
                        if root < tmin or best < root:
                            continue
                    best, hit = root, k
        i += 1
    return best, hit, visits


@pytest.mark.parametrize("layout", ["default", "reference"])
def test_threaded_bvh_equals_oracle_world_hit(pkg, orc, layout):
    """The compiled records, walked in Python, give the oracle's world.hit — in the default layout (the ground sphere is tested when a walk
    begins and is not in the tree: RtCompileInfo.first) and in the reference's (every member in the tree)."""
    hs = pkg.HostScene("book1", 1)
    flags = pkg._abi.RT_LAYOUT_LISTS_AS_REFERENCE if layout == "reference" else 0
    nodes, spheres, meta = pkg.compile_dump(hs.desc, flags)
    info = pkg.compile_info(hs.desc, flags)
    assert info["n_box_nodes"] == len(nodes) == 511 or info["n_box_nodes"] <= len(nodes)
    assert info["fits_lds"] == 1 and info["features"] == 0
    if layout == "reference":
        assert info["first"] == []
    else:
        assert len(info["first"]) == 1 and info["first"][0] >> 28 == 1          # one sphere: the r = 1000 ground
        assert spheres[info["first"][0] & 0xFFFFFF, 3] == 1000.0
    first = [k for w in info["first"] for k in range(w & 0xFFFFFF, (w & 0xFFFFFF) + ((w >> 24) & 15))]
    # every skip edge goes forward; the last subtree's skip is the end sentinel
    assert np.all(nodes["skip"] > np.arange(len(nodes))) and nodes["skip"].max() == len(nodes)
    rng = np.random.default_rng(11)
    n_hit = 0
    for _ in range(150):
        o = np.array([13, 2, 3]) + rng.normal(size=3) * 0.5
        d = rng.normal(size=3) * [1, 0.3, 1] - o / 4
        t, k, _ = traverse(nodes, spheres, o, d, first=first)
        ref = orc.world_hit(hs.desc, o, d)
        assert (ref is None) == (k < 0)
        if ref is not None:
            n_hit += 1
            assert t == pytest.approx(ref["t"], rel=1e-5)
    assert n_hit > 50


def test_bvh_leaf_order_and_node_visits_match_the_oracle(pkg, orc):
    """Same tree as the oracle's BVHNode::construct (bvh.rs:77-130 with the sub-range fix): same leaf
    order, and the threaded walk performs exactly the oracle's Aabb::hit count on every ray."""
    import ctypes as C
    rng = np.random.default_rng(4)
    b = pkg.SceneBuilder(bvh_seed=99)
    m = b.lambertian((0.5, 0.5, 0.5))
    ids = [b.sphere(rng.uniform(-8, 8, 3), rng.uniform(0.1, 1.0), m) for _ in range(137)]
    root = b.bvh(ids)
    desc = b.desc(root)
    nodes, spheres, meta = pkg.compile_dump(desc, pkg._abi.RT_LAYOUT_REFERENCE_COUNTERS)   # the reference's tree as it is: no extra boxes around spheres that share a node
    out = (C.c_int32 * 256)()
    n = orc.lib().orc_bvh_leaf_order(C.byref(desc), root, out, 256)
    assert n == 137
    oracle_centres = np.array([[desc.hittables[out[i]].p[k] for k in range(4)] for i in range(n)], dtype=np.float32)
    assert np.array_equal(oracle_centres, spheres)          # compiler emits primitives in leaf order
    cam = pkg.camera_new((0, 0, 30), (0, 0, 0), (0, 1, 0), 40, 1.0, 0.0, 10.0, 0, 0)
    prm = pkg.make_params(8, 8, 1, max_depth=1)
    _, st = orc.render(desc, cam, prm, precision=64, count=True)
    # replay the same 64 camera rays through the python walk
    visits = 0
    _, f64, _ = orc.rng_stream(1, 0, 0, 1)
    for y in range(8):
        for x in range(8):
            _, u, _ = orc.rng_stream(prm.seed, y * 8 + x, 0, 5)
            j = 7 - y
            s, t = (x + u[0]) / 7.0, (j + u[1]) / 7.0
            o = np.array(cam.origin.tuple())
            d = np.array(cam.lower_left_corner.tuple()) + np.array(cam.horizontal.tuple()) * s + np.array(cam.vertical.tuple()) * t - o
            visits += traverse(nodes, spheres, o, d)[2]
    assert visits == st["node_tests"]


def test_compile_cornell(pkg):
    hs = pkg.HostScene("cornell", 0)
    info = pkg.compile_info(hs.desc)
    assert info["n_rects"] == 12 and info["n_spheres"] == 1 and info["n_lights"] == 2 and info["n_xforms"] == 2
    F_RECT, F_XFORM, F_LIGHTS = 2, 16, 64
    assert info["features"] == F_RECT | F_XFORM | F_LIGHTS
    nodes, _, _ = pkg.compile_dump(hs.desc)
    types = [(int(n["leaf"]) >> 28, (int(n["leaf"]) >> 24) & 15) for n in nodes]
    # list order of main.rs:353-431: 6 wall/light rects, the transform record that opens the wrapper, the Box (one record; its six
    # sides are rects 6..11), the record that closes it (XFORM_EXIT = bit 23 of the payload), the glass sphere
    assert types == [(3, 6), (6, 0), (7, 1), (6, 0), (1, 1)]
    assert (int(nodes[1]["leaf"]) >> 23) & 1 == 0 and (int(nodes[3]["leaf"]) >> 23) & 1 == 1


def test_compile_rejects_malformed_graphs(pkg):
    A = pkg._abi
    b = pkg.SceneBuilder()
    m = b.lambertian((1, 1, 1))
    s = b.sphere((0, 0, 0), 1, m)
    bad = b._hit(A.RT_HIT_SPHERE, 99, p=[0, 0, 0, 1])                 # material id out of range
    with pytest.raises(pkg.RtError) as e:
        pkg.compile_info(b.desc(b.hittable_list([s, bad])))
    assert e.value.code == A.RT_ERR_INVALID
    b = pkg.SceneBuilder()
    with pytest.raises(pkg.RtError):
        pkg.compile_info(b.desc(5))                                     # world id out of range
    b = pkg.SceneBuilder()
    m = b.lambertian((1, 1, 1))
    tri = b.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), m)
    med = b.constant_medium(tri, 0.1, (1, 1, 1))                        # boundary must be a sphere or a box
    with pytest.raises(pkg.RtError) as e:
        pkg.compile_info(b.desc(b.hittable_list([med])))
    assert e.value.code == A.RT_ERR_UNSUPPORTED
    b = pkg.SceneBuilder()
    with pytest.raises(pkg.RtError):
        pkg.compile_info(b.desc(b.bvh([])))                             # empty BVH


def test_checker_nesting_and_image_validation(pkg):
    """Checkers nest (texture.rs:41-69) up to 8 deep on the device; deeper chains and cycles are refused, not mis-rendered.
    An image with data but a zero dimension is refused (its last row/column index would underflow on the device)."""
    import ctypes as C
    A = pkg._abi
    b = pkg.SceneBuilder()
    t = b.solid_color((1, 0, 0))
    for _ in range(8):
        t = b.checker_textures(t, b.solid_color((0, 1, 0)))
    s = b.sphere((0, 0, 0), 1, b.lambertian(texture=t))
    assert pkg.compile_info(b.desc(b.hittable_list([s])))["n_materials"] == 1           # 8 levels: fine
    t9 = b.checker_textures(t, b.solid_color((0, 0, 1)))
    s9 = b.sphere((0, 0, 0), 1, b.lambertian(texture=t9))
    with pytest.raises(pkg.RtError) as e:
        pkg.compile_info(b.desc(b.hittable_list([s9])))
    assert e.value.code == A.RT_ERR_UNSUPPORTED
    b = pkg.SceneBuilder()
    c = b.checker_textures(0, 0)                                                          # texture 0 = itself: a cycle
    with pytest.raises(pkg.RtError) as e:
        pkg.compile_info(b.desc(b.hittable_list([b.sphere((0, 0, 0), 1, b.lambertian(texture=c))])))
    assert e.value.code == A.RT_ERR_UNSUPPORTED
    b = pkg.SceneBuilder()
    px = np.zeros((4, 4, 3), dtype=np.uint8)
    tid = b.image(px)
    b.images[0] = A.RtImage(px.ctypes.data_as(C.POINTER(C.c_uint8)), 4, 0)              # data, width 4, height 0
    with pytest.raises(pkg.RtError) as e:
        pkg.compile_info(b.desc(b.hittable_list([b.sphere((0, 0, 0), 1, b.lambertian(texture=tid))])))
    assert e.value.code == A.RT_ERR_INVALID


def test_big_scene_compiles_fast(pkg):
    import time
    t = time.time()
    hs = pkg.HostScene("big", 5, 200000, 64)
    info = pkg.compile_info(hs.desc)
    dt = time.time() - t
    assert info["n_spheres"] == 200000 and info["n_tris"] == 2 * 64 * 32 and info["n_rects"] == 1
    assert info["fits_lds"] == 0
    assert dt < 60


def test_wrap_records(pkg):
    """Every distinct Translate/RotateY/FlipFace chain above a primitive becomes one wrap; a chain that moves
    the ray becomes one ENTER/EXIT pair around the child's nodes."""
    b = pkg.SceneBuilder()
    m = b.lambertian((1, 1, 1))
    box = b.box((0, 0, 0), (1, 1, 1), m)
    world = b.hittable_list([b.translate(b.rotate_y(box, 15), (1, 2, 3)), b.flip_face(b.xz_rect(0, 1, 0, 1, 5, m)), b.sphere((0, 0, 0), 1, m)])
    nodes, _, _ = pkg.compile_dump(b.desc(world))
    kinds = [int(n["leaf"]) >> 28 for n in nodes]
    assert kinds == [6, 7, 6, 3, 1]                       # enter the wrapper, the Box, leave it, flipped rect, sphere
    info = pkg.compile_info(b.desc(world))
    assert info["n_xforms"] == 2 and info["n_rects"] == 7
    b2 = pkg.SceneBuilder()
    m2 = b2.lambertian((1, 1, 1))
    s = b2.sphere((0, 0, 0), 1, m2)
    for _ in range(7):
        s = b2.translate(s, (1, 0, 0))
    with pytest.raises(pkg.RtError):
        pkg.compile_info(b2.desc(b2.hittable_list([s])))       # more than 6 wrappers in one chain


def test_sah_tree_is_a_valid_accelerator(pkg, orc):
    """RT_BVH_SAH: the python walk of the dumped tree finds the oracle's closest hit on every ray, with fewer visits."""
    A = pkg._abi
    rng = np.random.default_rng(21)
    visits = {}
    for builder, flags in ((A.RT_BVH_REFERENCE, 0), (A.RT_BVH_SAH, 0), (A.RT_BVH_REFERENCE, A.RT_LAYOUT_LISTS_AS_REFERENCE)):
        b = pkg.SceneBuilder(bvh_seed=5, bvh_builder=builder)
        m = b.lambertian((0.5, 0.5, 0.5))
        r2 = np.random.default_rng(8)
        ids = [b.sphere(r2.uniform(-10, 10, 3) * [1, 0.2, 1], r2.uniform(0.1, 0.8), m) for _ in range(300)] + [b.sphere((0, -1000, 0), 998, m)]
        desc = b.desc(b.bvh(ids))
        nodes, spheres, _ = pkg.compile_dump(desc, flags)
        first = [k for w in pkg.compile_info(desc, flags)["first"] for k in range(w & 0xFFFFFF, (w & 0xFFFFFF) + ((w >> 24) & 15))]
        assert len(first) == (0 if flags else 1)                # the ground sphere is tested before the walk, unless the layout is the reference's
        assert np.all(nodes["skip"] > np.arange(len(nodes)))
        total = 0
        rr = np.random.default_rng(9)
        for _ in range(120):
            o = rr.uniform(-12, 12, 3) * [1, 0.3, 1] + [0, 3, 0]
            d = rr.normal(size=3)
            t, k, v = traverse(nodes, spheres, o, d, first=first)
            ref = orc.world_hit(desc, o, d)
            assert (ref is None) == (k < 0)
            if ref is not None:
                assert t == pytest.approx(ref["t"], rel=1e-5)
            total += v
        visits[(builder, flags)] = total
    print("box visits:", visits)
    assert visits[(A.RT_BVH_SAH, 0)] < 0.8 * visits[(A.RT_BVH_REFERENCE, 0)]
    # the ground sphere out of the reference-shaped tree: the boxes above it shrink and the walk starts with its t_max
    assert visits[(A.RT_BVH_REFERENCE, 0)] < 0.8 * visits[(A.RT_BVH_REFERENCE, A.RT_LAYOUT_LISTS_AS_REFERENCE)]


def test_top_of_tree_layout(pkg):
    """Scenes that do not fit LDS keep the top of the tree there (kernels.hip M_TOP): the linked two-memory layout is checked on the host
    for a sphere BVH, a mixed scene with instance wrappers (ENTER/EXIT records) and media, and several sizes of the top."""
    import ctypes as C
    L = pkg.lib()

    def check(desc, max_top):
        n = C.c_uint64(0)
        rc = L.rt_scene_top_layout_check(C.byref(desc), max_top, C.byref(n))
        assert rc == 0, L.rt_last_error(None)
        return n.value
    big = pkg.HostScene("big", 5, 20000, 32)
    info = pkg.compile_info(big.desc)
    tops = [check(big.desc, k) for k in (1, 7, 64, 2048, 5000)]
    assert tops[0] == 1 and all(0 < t <= k for t, k in zip(tops, (1, 7, 64, 2048, 5000))) and tops == sorted(tops)
    assert check(big.desc, info["n_nodes"] + 10) == 0                 # everything fits: no top is built
    sah = pkg.HostScene("big_sah", 5, 20000, 32)
    assert 0 < check(sah.desc, 2048) <= 2048
    # wrappers, lists, media: records without boxes and ENTER/EXIT pairs are ordinary records of the walk
    for name in ("final", "cornell_smoke", "book1_ref"):
        hs = pkg.HostScene(name, 1)
        for k in (3, 50, 400):
            check(hs.desc, k)


def test_wide_tree_of_a_static_bvh(pkg, tmp_path):
    """The 8-wide tree scenes in HBM are walked through (csrc/wide_bvh.cpp), built and checked on the host: every primitive in exactly one
    leaf entry of <= 8 members of one kind, every entry's box — decoded with the device's float arithmetic — containing what is below it,
    depth within the walk's stack. Reference-shaped and SAH binary trees, spheres + triangles + a ground rect; scenes with lists, wrappers,
    media or moving spheres are not of that shape and keep the binary walk."""
    import crops as K
    obj = tmp_path / "t.obj"
    K.write_torus_obj(str(obj), 64, 32)                        # 4096 triangles
    for name in ("big_obj:", "big_obj_sah:"):
        hs = pkg.HostScene(name + str(obj), 5, 20000)
        w = pkg.wide_layout_check(hs.desc)
        assert w["n_prims"] == 20000 + 4096 + 1 and w["depth"] < 16
        assert w["mean_children"] > 5.0 and w["mean_leaf_members"] > 1.5, w
        print(name, w)
        assert w["n_nodes"] * 128 < 0.5 * (20000 + 4096) * 2 * 16 * 4          # smaller than the four binary record orders it replaces
    hs = pkg.HostScene("book1", 1)
    w = pkg.wide_layout_check(hs.desc)
    assert w["n_prims"] == 484
    for name in ("cornell", "cornell_smoke", "book1_ref"):
        hs = pkg.HostScene(name, 1)                      # (kept alive: its desc points into it)
        with pytest.raises(pkg.RtError) as e:
            pkg.wide_layout_check(hs.desc)
        assert e.value.code == pkg._abi.RT_ERR_UNSUPPORTED
