"""SceneBuilder: builds an RtSceneDesc from Python with the reference's constructor names.

Used by the tests to make small ad-hoc scenes; the scene functions of main.rs live in the C++ host
mirror (host/rt_host.hpp, exposed as api.HostScene)."""
import ctypes as C

import numpy as np

from . import _abi as A


class SceneBuilder:
    def __init__(self, background=(0.0, 0.0, 0.0), background_mode=A.RT_BG_CONSTANT, bvh_seed=1, bvh_builder=A.RT_BVH_REFERENCE):
        self.hittables, self.children, self.materials, self.textures, self.perlins, self.images = [], [], [], [], [], []
        self._image_arrays = []
        self.background, self.background_mode, self.bvh_seed, self.bvh_builder = background, background_mode, bvh_seed, bvh_builder
        self._keep = None

    # ---- textures (texture.rs) ----
    def solid_color(self, c):
        self.textures.append(A.RtTexture(A.RT_TEX_SOLID, -1, -1, 0, A.RtVec3(*map(float, c)), 0.0))
        return len(self.textures) - 1

    def checker(self, c1, c2):
        e, o = self.solid_color(c1), self.solid_color(c2)
        self.textures.append(A.RtTexture(A.RT_TEX_CHECKER, e, o, 0, A.RtVec3(0, 0, 0), 0.0))
        return len(self.textures) - 1

    def checker_textures(self, even, odd):
        """CheckerTexture over two arbitrary textures (texture.rs:41-50: `odd` and `even` are Arc<dyn Texture>, so checkers nest)."""
        self.textures.append(A.RtTexture(A.RT_TEX_CHECKER, int(even), int(odd), 0, A.RtVec3(0, 0, 0), 0.0))
        return len(self.textures) - 1

    def noise(self, scale, rng):
        """NoiseTexture::construct(scale) with Perlin::new (perlin.rs:14-25) drawn from numpy `rng`."""
        p = A.RtPerlin()
        for i in range(256):
            v = rng.uniform(-1.0, 1.0, 3)
            v = v / np.linalg.norm(v)
            for k in range(3):
                p.ranvec[i][k] = float(v[k])
        for arr in (p.perm_x, p.perm_y, p.perm_z):
            perm = list(range(256))
            for i in range(255, 0, -1):
                t = int(rng.integers(0, i + 1))
                perm[i], perm[t] = perm[t], perm[i]
            for i in range(256):
                arr[i] = perm[i]
        self.perlins.append(p)
        self.textures.append(A.RtTexture(A.RT_TEX_NOISE, len(self.perlins) - 1, -1, 0, A.RtVec3(0, 0, 0), float(scale)))
        return len(self.textures) - 1

    def image(self, rgb8):
        tid = -1
        if rgb8 is not None:
            arr = np.ascontiguousarray(rgb8, dtype=np.uint8)
            self._image_arrays.append(arr)
            h, w, _ = arr.shape
            self.images.append(A.RtImage(arr.ctypes.data_as(C.POINTER(C.c_uint8)), w, h))
            tid = len(self.images) - 1
        self.textures.append(A.RtTexture(A.RT_TEX_IMAGE, tid, -1, 0, A.RtVec3(0, 0, 0), 0.0))
        return len(self.textures) - 1

    # ---- materials (material.rs) ----
    def _mat(self, kind, texture=-1, albedo=(0, 0, 0), fuzz=0.0, ir=0.0):
        self.materials.append(A.RtMaterial(kind, texture, A.RtVec3(*map(float, albedo)), float(fuzz), float(ir)))
        return len(self.materials) - 1

    def lambertian(self, color=None, texture=None):
        return self._mat(A.RT_MAT_LAMBERTIAN, self.solid_color(color) if texture is None else texture)

    def metal(self, albedo, fuzz):
        return self._mat(A.RT_MAT_METAL, -1, albedo, fuzz if fuzz < 1.0 else 1.0)

    def dielectric(self, ir):
        return self._mat(A.RT_MAT_DIELECTRIC, -1, ir=ir)

    def diffuse_light(self, color):
        return self._mat(A.RT_MAT_DIFFUSE_LIGHT, self.solid_color(color))

    def isotropic(self, color):
        return self._mat(A.RT_MAT_ISOTROPIC, self.solid_color(color))

    # ---- hittables ----
    def _hit(self, kind, material=-1, first_child=-1, n_children=0, p=()):
        arr = (C.c_double * 10)(*([float(x) for x in p] + [0.0] * (10 - len(p))))
        self.hittables.append(A.RtHittable(kind, material, first_child, n_children, arr))
        return len(self.hittables) - 1

    def sphere(self, center, radius, mat):
        return self._hit(A.RT_HIT_SPHERE, mat, p=list(center) + [radius])

    def moving_sphere(self, c0, c1, t0, t1, radius, mat):
        return self._hit(A.RT_HIT_MOVING_SPHERE, mat, p=list(c0) + list(c1) + [t0, t1, radius])

    def xy_rect(self, x0, x1, y0, y1, k, mat):
        return self._hit(A.RT_HIT_XY_RECT, mat, p=[x0, x1, y0, y1, k])

    def xz_rect(self, x0, x1, z0, z1, k, mat):
        return self._hit(A.RT_HIT_XZ_RECT, mat, p=[x0, x1, z0, z1, k])

    def yz_rect(self, y0, y1, z0, z1, k, mat):
        return self._hit(A.RT_HIT_YZ_RECT, mat, p=[y0, y1, z0, z1, k])

    def triangle(self, v0, v1, v2, mat):
        return self._hit(A.RT_HIT_TRIANGLE, mat, p=list(v0) + list(v1) + list(v2))

    def box(self, p0, p1, mat):
        return self._hit(A.RT_HIT_BOX, mat, p=list(p0) + list(p1))

    def _group(self, kind, ids, p=()):
        first = len(self.children)
        self.children.extend(int(i) for i in ids)
        return self._hit(kind, -1, first, len(ids), p)

    def hittable_list(self, ids):
        return self._group(A.RT_HIT_LIST, ids)

    def bvh(self, ids, time0=0.0, time1=0.0):
        return self._group(A.RT_HIT_BVH, ids, [time0, time1])

    def translate(self, child, offset):
        return self._hit(A.RT_HIT_TRANSLATE, -1, child, 1, list(offset))

    def rotate_y(self, child, degrees):
        return self._hit(A.RT_HIT_ROTATE_Y, -1, child, 1, [degrees])

    def flip_face(self, child):
        return self._hit(A.RT_HIT_FLIP_FACE, -1, child, 1)

    def constant_medium(self, boundary, density, color):
        return self._hit(A.RT_HIT_CONSTANT_MEDIUM, self.isotropic(color), boundary, 1, [density])

    # ---- finish ----
    def desc(self, world, lights=-1):
        def arr(T, items):
            return (T * max(1, len(items)))(*items)
        keep = dict(h=arr(A.RtHittable, self.hittables), c=arr(C.c_int32, self.children), m=arr(A.RtMaterial, self.materials),
                    t=arr(A.RtTexture, self.textures), p=arr(A.RtPerlin, self.perlins), i=arr(A.RtImage, self.images))
        d = A.RtSceneDesc()
        d.abi_version = A.RT_ABI_VERSION
        d.hittables, d.n_hittables = keep["h"], len(self.hittables)
        d.children, d.n_children = keep["c"], len(self.children)
        d.materials, d.n_materials = keep["m"], len(self.materials)
        d.textures, d.n_textures = keep["t"], len(self.textures)
        d.perlins, d.n_perlins = keep["p"], len(self.perlins)
        d.images, d.n_images = keep["i"], len(self.images)
        d.world, d.lights = world, lights
        d.background_mode = self.background_mode
        d.background = A.RtVec3(*map(float, self.background))
        d.bvh_seed = self.bvh_seed
        d.bvh_builder = self.bvh_builder
        self._keep = keep   # the desc points into these arrays
        return d
