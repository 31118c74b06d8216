"""JSON scene description -> RtSceneDesc + RtCamera.

The reference picks its scene by editing main() (main.rs:666-703); doc/rust-bonus.md (Track 5)
suggests a scene file. This is that front-end for the flat C ABI: a JSON document names textures,
materials and hittables with the reference's constructor names and argument order and is turned into
SceneBuilder calls — nothing here touches the device.

    {
      "background": [0, 0, 0],                      # or {"sky": [0.5, 0.7, 1.0]} for the book-1 gradient
      "camera": {"lookfrom": [278, 278, -800], "lookat": [278, 278, 0], "vup": [0, 1, 0],
                 "vfov": 40, "aperture": 0, "focus_dist": 10, "time0": 0, "time1": 1},
      "textures":  {"white": {"solid": [0.73, 0.73, 0.73]}, "tiles": {"checker": ["white", [0.2, 0.3, 0.1]]},
                    "marble": {"noise": 0.1, "seed": 7}, "earth": {"image": "earthmap.png"}},
      "materials": {"wall": {"lambertian": "white"}, "lamp": {"diffuse_light": [15, 15, 15]},
                    "glass": {"dielectric": 1.5}, "steel": {"metal": [0.8, 0.85, 0.88], "fuzz": 0.0},
                    "smoke": {"isotropic": [1, 1, 1]}},
      "objects":   {"floor": {"xz_rect": [0, 555, 0, 555, 0], "material": "wall"},
                    "ball":  {"sphere": [[190, 90, 190], 90], "material": "glass"},
                    "block": {"box": [[0, 0, 0], [165, 330, 165]], "material": "wall"},
                    "tall":  {"translate": ["turned", [265, 0, 295]]}, "turned": {"rotate_y": ["block", 15]},
                    "fog":   {"constant_medium": ["ball", 0.01], "material": "smoke"},
                    "mesh":  {"obj": "bunny.obj", "material": "steel", "scale": 1.0, "offset": [0, 0, 0]}},
      "world":  {"list": ["floor", "tall", "fog"]},  # or {"bvh": [...]} = BVHNode::construct2 over the members
      "lights": ["lamp_rect", "ball"],              # optional: XzRect / Sphere objects (main.rs:669-686)
      "bvh": {"seed": 1, "builder": "reference"}    # or "sah"
    }

Hittable kinds: sphere [center, radius]; moving_sphere [c0, c1, t0, t1, radius]; xy_rect / xz_rect /
yz_rect [a0, a1, b0, b1, k]; triangle [v0, v1, v2]; box [p0, p1]; translate [child, offset]; rotate_y
[child, degrees]; flip_face child; constant_medium [boundary, density] (+ "material": an isotropic);
list [...]; bvh [...]; obj "file.obj" (triangles of a Wavefront OBJ, in a BVH).
Errors raise ValueError with the offending key.
"""
import json
import os

import numpy as np

from . import _abi as A
from .scene import SceneBuilder


def _vec3(v, what):
    if not (isinstance(v, (list, tuple)) and len(v) == 3 and all(isinstance(x, (int, float)) for x in v)):
        raise ValueError(f"{what}: expected [x, y, z], got {v!r}")
    return [float(x) for x in v]


def parse_obj(path):
    """Triangles (n, 3, 3) of a Wavefront OBJ: `v` records and `f` records (fans; v/vt/vn forms, negative indices)."""
    verts, tris = [], []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                if len(t) < 4:
                    raise ValueError(f"{path}:{ln}: vertex needs three coordinates")
                verts.append([float(t[1]), float(t[2]), float(t[3])])
            elif t[0] == "f":
                idx = []
                for w in t[1:]:
                    i = int(w.split("/")[0])
                    i = i - 1 if i > 0 else len(verts) + i
                    if not 0 <= i < len(verts):
                        raise ValueError(f"{path}:{ln}: face index {w} out of range")
                    idx.append(i)
                for k in range(1, len(idx) - 1):
                    tris.append([verts[idx[0]], verts[idx[k]], verts[idx[k + 1]]])
    return np.asarray(tris, dtype=np.float64).reshape(-1, 3, 3)


class JsonScene:
    """desc (RtSceneDesc), camera(aspect) -> RtCamera; keeps the arrays the desc points into alive."""

    def __init__(self, doc, base_dir="."):
        if isinstance(doc, (str, bytes)):
            doc = json.loads(doc)
        self.doc, self.base_dir = doc, base_dir
        bg = doc.get("background", [0, 0, 0])
        mode = A.RT_BG_CONSTANT
        if isinstance(bg, dict):
            if "sky" not in bg:
                raise ValueError("background: expected [r, g, b] or {\"sky\": [r, g, b]}")
            bg, mode = bg["sky"], A.RT_BG_SKY_GRADIENT
        bvh = doc.get("bvh", {})
        builders = {"reference": A.RT_BVH_REFERENCE, "sah": A.RT_BVH_SAH}
        if bvh.get("builder", "reference") not in builders:
            raise ValueError(f"bvh.builder: expected one of {sorted(builders)}")
        self.b = SceneBuilder(background=_vec3(bg, "background"), background_mode=mode, bvh_seed=int(bvh.get("seed", 1)),
                              bvh_builder=builders[bvh.get("builder", "reference")])
        self._tex, self._mat, self._obj, self._busy = {}, {}, {}, set()
        if "world" not in doc:
            raise ValueError("world: missing")
        world = self._group(doc["world"], "world")
        lights = -1
        if doc.get("lights"):
            lights = self.b.hittable_list([self._object(n) for n in doc["lights"]])
        self.desc = self.b.desc(world, lights)

    # ---- textures / materials ----
    def _colour_texture(self, v, what):
        """A texture reference: a name from "textures" or an inline [r, g, b]."""
        if isinstance(v, str):
            return self._texture(v)
        return self.b.solid_color(_vec3(v, what))

    def _texture(self, name):
        if name in self._tex:
            return self._tex[name]
        t = self.doc.get("textures", {}).get(name)
        if t is None:
            raise ValueError(f"textures.{name}: not defined")
        if "solid" in t:
            tid = self.b.solid_color(_vec3(t["solid"], f"textures.{name}.solid"))
        elif "checker" in t:
            if len(t["checker"]) != 2:
                raise ValueError(f"textures.{name}.checker: expected [even, odd]")
            # CheckerTexture::construct_color: two solid colours (a name must itself be a solid colour)
            cs = []
            for v in t["checker"]:
                if isinstance(v, str):
                    s = self.doc.get("textures", {}).get(v, {})
                    if "solid" not in s:
                        raise ValueError(f"textures.{name}.checker: {v!r} is not a solid colour")
                    v = s["solid"]
                cs.append(_vec3(v, f"textures.{name}.checker"))
            tid = self.b.checker(cs[0], cs[1])
        elif "noise" in t:
            tid = self.b.noise(float(t["noise"]), np.random.default_rng(int(t.get("seed", 1))))
        elif "image" in t:
            from PIL import Image
            path = os.path.join(self.base_dir, t["image"])
            tid = self.b.image(np.asarray(Image.open(path).convert("RGB")))
        else:
            raise ValueError(f"textures.{name}: expected one of solid / checker / noise / image")
        self._tex[name] = tid
        return tid

    def _material(self, name):
        if name in self._mat:
            return self._mat[name]
        m = self.doc.get("materials", {}).get(name)
        if m is None:
            raise ValueError(f"materials.{name}: not defined")
        w = f"materials.{name}"
        if "lambertian" in m:
            v = m["lambertian"]
            mid = self.b.lambertian(texture=self._texture(v)) if isinstance(v, str) else self.b.lambertian(color=_vec3(v, w))
        elif "metal" in m:
            mid = self.b.metal(_vec3(m["metal"], w), float(m.get("fuzz", 0.0)))
        elif "dielectric" in m:
            mid = self.b.dielectric(float(m["dielectric"]))
        elif "diffuse_light" in m:
            mid = self.b.diffuse_light(_vec3(m["diffuse_light"], w))
        elif "isotropic" in m:
            mid = self.b.isotropic(_vec3(m["isotropic"], w))
        else:
            raise ValueError(f"{w}: expected one of lambertian / metal / dielectric / diffuse_light / isotropic")
        self._mat[name] = mid
        return mid

    # ---- hittables ----
    def _group(self, g, what):
        if not isinstance(g, dict) or len([k for k in ("list", "bvh") if k in g]) != 1:
            raise ValueError(f"{what}: expected {{\"list\": [...]}} or {{\"bvh\": [...]}}")
        kind = "list" if "list" in g else "bvh"
        ids = [self._object(n) for n in g[kind]]
        return self.b.hittable_list(ids) if kind == "list" else self.b.bvh(ids, float(g.get("time0", 0.0)), float(g.get("time1", 0.0)))

    def _object(self, name):
        if name in self._obj:
            return self._obj[name]
        o = self.doc.get("objects", {}).get(name)
        if o is None:
            raise ValueError(f"objects.{name}: not defined")
        if name in self._busy:
            raise ValueError(f"objects.{name}: refers to itself")
        self._busy.add(name)
        w = f"objects.{name}"
        b = self.b

        def mat():
            if "material" not in o:
                raise ValueError(f"{w}: needs a material")
            return self._material(o["material"])

        def args(key, n):
            a = o[key]
            if not isinstance(a, (list, tuple)) or len(a) != n:
                raise ValueError(f"{w}.{key}: expected {n} arguments")
            return a

        if "sphere" in o:
            c, r = args("sphere", 2)
            hid = b.sphere(_vec3(c, w), float(r), mat())
        elif "moving_sphere" in o:
            c0, c1, t0, t1, r = args("moving_sphere", 5)
            hid = b.moving_sphere(_vec3(c0, w), _vec3(c1, w), float(t0), float(t1), float(r), mat())
        elif any(k in o for k in ("xy_rect", "xz_rect", "yz_rect")):
            k = next(k for k in ("xy_rect", "xz_rect", "yz_rect") if k in o)
            hid = getattr(b, k)(*[float(x) for x in args(k, 5)], mat())
        elif "triangle" in o:
            v0, v1, v2 = args("triangle", 3)
            hid = b.triangle(_vec3(v0, w), _vec3(v1, w), _vec3(v2, w), mat())
        elif "box" in o:
            p0, p1 = args("box", 2)
            hid = b.box(_vec3(p0, w), _vec3(p1, w), mat())
        elif "translate" in o:
            child, off = args("translate", 2)
            hid = b.translate(self._object(child), _vec3(off, w))
        elif "rotate_y" in o:
            child, deg = args("rotate_y", 2)
            hid = b.rotate_y(self._object(child), float(deg))
        elif "flip_face" in o:
            hid = b.flip_face(self._object(o["flip_face"]))
        elif "constant_medium" in o:
            boundary, density = args("constant_medium", 2)
            m = self.doc.get("materials", {}).get(o.get("material", ""), {})
            if "isotropic" not in m:
                raise ValueError(f"{w}: a constant_medium needs an isotropic material")
            hid = b.constant_medium(self._object(boundary), float(density), _vec3(m["isotropic"], w))
        elif "list" in o or "bvh" in o:
            hid = self._group(o, w)
        elif "obj" in o:
            tris = parse_obj(os.path.join(self.base_dir, o["obj"])) * float(o.get("scale", 1.0)) + np.asarray(_vec3(o.get("offset", [0, 0, 0]), w))
            if len(tris) == 0:
                raise ValueError(f"{w}.obj: no faces")
            m = mat()
            hid = b.bvh([b.triangle(list(t[0]), list(t[1]), list(t[2]), m) for t in tris])
        else:
            raise ValueError(f"{w}: unknown hittable kind")
        self._busy.discard(name)
        self._obj[name] = hid
        return hid

    # ---- camera ----
    def camera(self, aspect_ratio):
        from .api import camera_new
        c = self.doc.get("camera")
        if c is None:
            raise ValueError("camera: missing")
        return camera_new(_vec3(c["lookfrom"], "camera.lookfrom"), _vec3(c["lookat"], "camera.lookat"), _vec3(c.get("vup", [0, 1, 0]), "camera.vup"),
                          float(c.get("vfov", 40.0)), float(aspect_ratio), float(c.get("aperture", 0.0)), float(c.get("focus_dist", 10.0)),
                          float(c.get("time0", 0.0)), float(c.get("time1", 0.0)))


def load_scene(path):
    """Scene file -> JsonScene (image / obj paths are relative to the file)."""
    with open(path) as f:
        return JsonScene(json.load(f), os.path.dirname(os.path.abspath(path)))
