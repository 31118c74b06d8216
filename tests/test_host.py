"""Host-side mirror of the reference API (host/rt_host.hpp): scene functions, write_color, PNG, tiling."""
import ctypes as C

import numpy as np
import pytest


def hittables(desc):
    return [desc.hittables[i] for i in range(desc.n_hittables)]


def test_book1_scene_matches_the_book_definition(pkg):
    A = pkg._abi
    hs = pkg.HostScene("book1_list", 1)
    d = hs.desc
    hl = hittables(d)
    world = hl[d.world]
    assert world.kind == A.RT_HIT_LIST and d.lights == -1 and d.background_mode == A.RT_BG_SKY_GRADIENT
    ids = [d.children[world.first_child + i] for i in range(world.n_children)]
    sph = [hl[i] for i in ids]
    assert all(s.kind == A.RT_HIT_SPHERE for s in sph)
    assert 440 <= len(sph) <= 488                                   # 22*22 grid minus rejects + ground + 3 big ones
    g = sph[0]                                                      # main.rs:174-179 ground
    assert tuple(g.p[:4]) == (0.0, -1000.0, 0.0, 1000.0)
    big = [tuple(s.p[:4]) for s in sph[-3:]]                        # main.rs:220-239
    assert big == [(0.0, 1.0, 0.0, 1.0), (-4.0, 1.0, 0.0, 1.0), (4.0, 1.0, 0.0, 1.0)]
    mats = [d.materials[s.material] for s in sph]
    assert mats[-3].kind == A.RT_MAT_DIELECTRIC and mats[-3].ir == 1.5
    assert mats[-1].kind == A.RT_MAT_METAL and mats[-1].fuzz == 0.0 and mats[-1].albedo.tuple() == (0.7, 0.6, 0.5)
    small = sph[1:-3]
    kinds = np.array([d.materials[s.material].kind for s in small])
    assert 0.7 < np.mean(kinds == A.RT_MAT_LAMBERTIAN) < 0.9 and 0.08 < np.mean(kinds == A.RT_MAT_METAL) < 0.22
    for s in small:
        c = np.array(s.p[:3])
        assert s.p[3] == 0.2 and c[1] == 0.2 and np.linalg.norm(c - np.array([4, 0.2, 0])) > 0.9
        m = d.materials[s.material]
        if m.kind == A.RT_MAT_METAL:
            assert 0.5 <= m.albedo.x < 1 and 0 <= m.fuzz < 0.5
    # same seed -> same scene; the BVH variant holds the same spheres
    hs2 = pkg.HostScene("book1", 1)
    h2 = hittables(hs2.desc)
    assert h2[hs2.desc.world].kind == A.RT_HIT_BVH and h2[hs2.desc.world].n_children == len(sph)
    cam = hs.camera(1.5)                                            # main.rs:706-709 + book: vfov 20, aperture 0.1, focus 10
    assert cam.origin.tuple() == (13.0, 2.0, 3.0) and cam.lens_radius == 0.05


def test_cornell_box_matches_main_rs(pkg):
    A = pkg._abi
    hs = pkg.HostScene("cornell", 0)
    d = hs.desc
    hl = hittables(d)
    world = hl[d.world]
    ids = [d.children[world.first_child + i] for i in range(world.n_children)]
    kinds = [hl[i].kind for i in ids]
    assert kinds == [A.RT_HIT_YZ_RECT, A.RT_HIT_YZ_RECT, A.RT_HIT_FLIP_FACE, A.RT_HIT_XZ_RECT, A.RT_HIT_XZ_RECT, A.RT_HIT_XY_RECT,
                     A.RT_HIT_TRANSLATE, A.RT_HIT_SPHERE]                          # main.rs:353-431
    light = hl[hl[ids[2]].first_child]
    assert tuple(light.p[:5]) == (213.0, 343.0, 227.0, 332.0, 554.0)
    assert d.materials[light.material].kind == A.RT_MAT_DIFFUSE_LIGHT
    assert d.textures[d.materials[light.material].texture].color.tuple() == (15.0, 15.0, 15.0)
    tr = hl[ids[6]]
    assert tuple(tr.p[:3]) == (265.0, 0.0, 295.0)
    rot = hl[tr.first_child]
    assert rot.kind == A.RT_HIT_ROTATE_Y and rot.p[0] == 15.0
    box = hl[rot.first_child]
    assert box.kind == A.RT_HIT_BOX and tuple(box.p[:6]) == (0, 0, 0, 165.0, 330.0, 165.0)
    assert tuple(hl[ids[7]].p[:4]) == (190.0, 90.0, 190.0, 90.0)
    L = hl[d.lights]                                                               # main.rs:669-684
    lids = [d.children[L.first_child + i] for i in range(L.n_children)]
    assert [hl[i].kind for i in lids] == [A.RT_HIT_XZ_RECT, A.RT_HIT_SPHERE]
    assert d.background.tuple() == (0.0, 0.0, 0.0) and d.background_mode == A.RT_BG_CONSTANT


def test_final_scene_shape(pkg, earth):
    A = pkg._abi
    hs = pkg.HostScene("final", 3, image=earth)
    d = hs.desc
    info = pkg.compile_info(d)
    assert info["n_rects"] == 400 * 6 + 1 and info["n_spheres"] == 1000 + 5 + 2   # world spheres + 2 private medium boundaries
    assert info["n_moving"] == 1 and info["n_media"] == 2 and info["n_xforms"] == 2
    assert d.n_images == 1 and d.images[0].width == 1024 and d.images[0].height == 512
    assert d.n_perlins == 1
    perm = sorted(d.perlins[0].perm_x[i] for i in range(256))
    assert perm == list(range(256))                                                # perlin.rs:53-66: a permutation
    v = np.array([[d.perlins[0].ranvec[i][k] for k in range(3)] for i in range(256)])
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0)                             # perlin.rs:16-18: unit vectors


def test_png_writer_roundtrip(pkg, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    path = tmp_path / "a.png"
    assert pkg.write_png(path, img) == 0
    back = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(back, img)
    assert pkg.write_png(tmp_path / "no_such_dir" / "a.png", img) == -1            # encode failure is reported, not fatal (main.rs:793-796)


def test_jpeg_writer_and_output_layout(pkg, tmp_path):
    """The reference writes its frame as JPEG, quality 100, into output/book3/image12.jpg after creating the directories
    (main.rs:653-656, 721, 791-796). rt_host_write_image does both; the file is a baseline JFIF any decoder reads (PIL here), and at
    quality 100 without chroma subsampling it is the picture to within +-2 of 255 (a gradient, noise, and a frame smaller than a block)."""
    from PIL import Image
    rng = np.random.default_rng(3)
    y, x = np.mgrid[0:150, 0:203]
    grad = np.stack([(x * 255 // 202), (y * 255 // 149), ((x + y) * 255 // 351)], axis=2).astype(np.uint8)
    noise = rng.integers(0, 256, size=(64, 48, 3), dtype=np.uint8)
    tiny = rng.integers(0, 256, size=(5, 3, 3), dtype=np.uint8)
    for name, img, tol in (("grad", grad, 2), ("tiny", tiny, 3), ("noise", noise, 3)):
        path = tmp_path / "output" / "book3" / (name + ".jpg")
        assert pkg.write_image(path, img, 100) == 0
        back = np.asarray(Image.open(path).convert("RGB"))
        assert back.shape == img.shape
        assert np.abs(back.astype(int) - img.astype(int)).max() <= tol, (name, np.abs(back.astype(int) - img.astype(int)).max())
        assert open(path, "rb").read(4)[:2] == b"\xff\xd8" and Image.open(path).format == "JPEG"
    lo = tmp_path / "output" / "book3" / "q50.jpg"
    assert pkg.write_image(lo, grad, 50) == 0 and lo.stat().st_size < (tmp_path / "output" / "book3" / "grad.jpg").stat().st_size
    assert np.abs(np.asarray(Image.open(lo).convert("RGB")).astype(int) - grad.astype(int)).mean() < 3.0
    assert pkg.write_image(tmp_path / "a" / "b.png", grad) == 0 and np.array_equal(np.asarray(Image.open(tmp_path / "a" / "b.png")), grad)
    assert pkg.write_image(tmp_path / "c.bmp", grad) == -1


def test_tonemap_is_write_color(pkg):
    rng = np.random.default_rng(2)
    s = rng.uniform(0, 40, size=(5, 7, 3)).astype(np.float32)
    s[0, 0, 0] = np.nan
    out = pkg.tonemap(s, 16)
    for y in range(5):
        for x in range(7):
            assert tuple(out[y, x]) == pkg.write_color(s[y, x].astype(np.float64), 16)
    assert out[0, 0, 0] == 0


def py_tile_layout(width, height, ts, world):
    """Independent statement of the shard layout of include/rt_hip.h: returns per-rank index maps."""
    tiles_x, tiles_y = -(-width // ts), -(-height // ts)
    n_tiles = tiles_x * tiles_y
    per = -(-n_tiles // world) * ts * ts * 3
    maps = []
    for r in range(world):
        m = np.full(per // 3, -1, dtype=np.int64)
        for lt, tile in enumerate(range(r, n_tiles, world)):
            tx, ty = tile % tiles_x, tile // tiles_x
            for py in range(ts):
                for px in range(ts):
                    x, y = tx * ts + px, ty * ts + py
                    if x < width and y < height:
                        m[lt * ts * ts + py * ts + px] = y * width + x
        maps.append(m)
    return maps


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_untile(pkg, world):
    W, H, ts = 70, 45, 16
    full = np.random.default_rng(3).uniform(0, 9, size=(H * W, 3)).astype(np.float32)
    maps = py_tile_layout(W, H, ts, world)
    bufs = []
    for m in maps:
        b = np.zeros((len(m), 3), dtype=np.float32)
        b[m >= 0] = full[m[m >= 0]]
        bufs.append(b.reshape(-1))
    prm = pkg.make_params(W, H, 1, tile_size=ts, shard_index=0, shard_count=world)
    assert pkg.output_floats(prm) == len(bufs[0]) or world == 1
    out = pkg.untile(prm, np.concatenate(bufs)) if world > 1 else None
    if world > 1:
        assert np.array_equal(out.reshape(-1, 3), full)


CUBE_OBJ = """# unit cube, quads + a triangle fan, mixed index forms
v -1 -1 -1
v  1 -1 -1
v  1  1 -1
v -1  1 -1
v -1 -1  1
v  1 -1  1
v  1  1  1
v -1  1  1
vn 0 0 1
f 1 2 3 4
f 5/1/1 6/2/1 7/3/1 8/4/1
f 1//1 2//1 6//1 5//1
f -6 -5 -1 -2
f 2 3 7
f 2 7 6
f 1 4 8 5
"""


def test_obj_loader(pkg, tmp_path):
    A = pkg._abi
    path = tmp_path / "cube.obj"
    path.write_text(CUBE_OBJ)
    hs = pkg.HostScene("obj:" + str(path), 1)
    d = hs.desc
    info = pkg.compile_info(d)
    assert info["n_tris"] == 12 and info["n_rects"] == 1            # 5 quads (2 each) + 2 triangles; the ground
    tris = [d.hittables[i] for i in range(d.n_hittables) if d.hittables[i].kind == A.RT_HIT_TRIANGLE]
    assert tuple(tris[0].p[:9]) == (-1, -1, -1, 1, -1, -1, 1, 1, -1)  # f 1 2 3 (first fan triangle of the first quad)
    assert tuple(tris[6].p[:9]) == (1, 1, -1, -1, 1, -1, -1, 1, 1)    # f -6 -5 -1 -2 -> v3 v4 v8 v7: first fan triangle
    with pytest.raises(pkg.RtError):
        pkg.HostScene("obj:" + str(tmp_path / "missing.obj"), 1)
