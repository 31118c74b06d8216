"""ctypes binding of the CPU oracle (oracle/oracle.cpp). TEST INFRASTRUCTURE: import only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
RT_N_PRIM_TYPES = 6


class OrcOpts(C.Structure):
    _fields_ = [("precision", C.c_int32), ("n_threads", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("count", C.c_int32), ("_pad", C.c_int32)]


class OrcStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("node_tests", C.c_uint64), ("draws", C.c_uint64), ("nonfinite_samples", C.c_uint64),
                ("prim_tests", C.c_uint64 * RT_N_PRIM_TYPES), ("seconds", C.c_double)]

    def as_dict(self):
        return dict(samples=self.samples, segments=self.segments, node_tests=self.node_tests, draws=self.draws,
                    nonfinite_samples=self.nonfinite_samples, prim_tests=list(self.prim_tests), seconds=self.seconds)


def build(force=False):
    src = os.path.join(HERE, "oracle.cpp")
    hdr = os.path.join(HERE, "..", "include", "rt_hip.h")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        r = subprocess.run(["make", "-C", HERE] + (["-B"] if force else []), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("RT_ORACLE_LIB", LIB_PATH)     # a sanitizer build of the checker (scripts/asan_host.sh)
        if path == LIB_PATH and not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(path)
        vp, f64p = C.c_void_p, C.POINTER(C.c_double)
        L.orc_last_error.restype = C.c_char_p
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [vp, vp, vp, C.POINTER(OrcOpts), f64p, C.POINTER(OrcStats)]
        L.orc_render_samples.restype = C.c_int
        L.orc_render_samples.argtypes = [vp, vp, vp, C.POINTER(OrcOpts), f64p, f64p, C.POINTER(OrcStats)]
        L.orc_render_crops.restype = C.c_int
        L.orc_render_crops.argtypes = [vp, vp, vp, C.POINTER(OrcOpts), C.c_int32, C.POINTER(C.c_int32), f64p, C.POINTER(OrcStats)]
        L.orc_render_reference_shaped.restype = C.c_int
        L.orc_render_reference_shaped.argtypes = [vp, vp, vp, C.c_int32, C.POINTER(C.c_int32), f64p, C.POINTER(OrcStats)]
        L.orc_write_color.restype = None
        L.orc_write_color.argtypes = [f64p, C.c_uint32, C.POINTER(C.c_uint8)]
        L.orc_camera_new.restype = None
        L.orc_camera_new.argtypes = [f64p, f64p, f64p, f64p, C.c_double, C.c_double, vp]
        L.orc_sphere_hit.restype = C.c_int
        L.orc_sphere_hit.argtypes = [f64p, C.c_double, f64p, f64p, C.c_double, C.c_double, C.c_double, f64p]
        L.orc_sphere_pdf_value.restype = C.c_double
        L.orc_sphere_pdf_value.argtypes = [f64p, C.c_double, f64p, f64p]
        L.orc_rect_hit.restype = C.c_int
        L.orc_rect_hit.argtypes = [C.c_int, f64p, f64p, f64p, C.c_double, C.c_double, f64p]
        L.orc_xzrect_pdf_value.restype = C.c_double
        L.orc_xzrect_pdf_value.argtypes = [f64p, f64p, f64p]
        L.orc_onb_build_from_w.restype = None
        L.orc_onb_build_from_w.argtypes = [f64p, f64p]
        L.orc_reflectance.restype = C.c_double
        L.orc_reflectance.argtypes = [C.c_double, C.c_double]
        L.orc_refract.restype = None
        L.orc_refract.argtypes = [f64p, f64p, C.c_double, f64p]
        L.orc_reflect.restype = None
        L.orc_reflect.argtypes = [f64p, f64p, f64p]
        L.orc_aabb_hit.restype = C.c_int
        L.orc_aabb_hit.argtypes = [f64p, f64p, f64p, f64p, C.c_double, C.c_double, C.c_int]
        L.orc_world_hit.restype = C.c_int
        L.orc_world_hit.argtypes = [vp, f64p, f64p, C.c_double, C.c_double, C.c_double, f64p]
        L.orc_texture_value.restype = C.c_int
        L.orc_texture_value.argtypes = [vp, C.c_int, C.c_double, C.c_double, f64p, f64p]
        L.orc_rng_stream.restype = None
        L.orc_rng_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), f64p, C.POINTER(C.c_float)]
        L.orc_bvh_leaf_order.restype = C.c_int
        L.orc_bvh_leaf_order.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_uint64]
        _lib = L
    return _lib


def _d(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


def render(desc, cam, params, precision=64, n_threads=1, rect=None, count=False, per_sample=False):
    """Returns (rgb_sum float64 (H,W,3), stats dict[, per_sample float64 (h,w,spp,3)])."""
    L = lib()
    H, W, spp = params.height, params.width, params.samples_per_pixel
    x0, y0, x1, y1 = rect if rect else (0, 0, W, H)
    opts = OrcOpts(precision, n_threads, x0, y0, x1, y1, 1 if count else 0, 0)
    out = np.zeros((H, W, 3), dtype=np.float64)
    st = OrcStats()
    if per_sample:
        ps = np.zeros((y1 - y0, x1 - x0, spp, 3), dtype=np.float64)
        rc = L.orc_render_samples(C.byref(desc), C.byref(cam), C.byref(params), C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_double)),
                                  ps.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    else:
        rc = L.orc_render(C.byref(desc), C.byref(cam), C.byref(params), C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle: " + L.orc_last_error().decode())
    return (out, st.as_dict(), ps) if per_sample else (out, st.as_dict())


def render_crops(desc, cam, params, rects, precision=64, n_threads=1, count=False):
    """Several rectangles (x0, y0, x1, y1) of one frame with one scene build. Returns ([rgb_sum float64 (h,w,3)], [stats dict])."""
    L = lib()
    opts = OrcOpts(precision, n_threads, 0, 0, 0, 0, 1 if count else 0, 0)
    flat = (C.c_int32 * (4 * len(rects)))(*[int(v) for r in rects for v in r])
    sizes = [(r[3] - r[1], r[2] - r[0]) for r in rects]
    out = np.zeros(sum(h * w for h, w in sizes) * 3, dtype=np.float64)
    st = (OrcStats * len(rects))()
    rc = L.orc_render_crops(C.byref(desc), C.byref(cam), C.byref(params), C.byref(opts), len(rects), flat, out.ctypes.data_as(C.POINTER(C.c_double)), st)
    if rc != 0:
        raise RuntimeError("oracle: " + L.orc_last_error().decode())
    crops, at = [], 0
    for h, w in sizes:
        crops.append(out[at:at + h * w * 3].reshape(h, w, 3).copy())
        at += h * w * 3
    return crops, [s.as_dict() for s in st]


def render_reference_shaped(desc, cam, params, thread_num, rect):
    """main.rs:730-778 as written (one pixel at a time, thread_num threads spawned per pixel). Returns (rgb_sum (h,w,3), stats)."""
    L = lib()
    x0, y0, x1, y1 = rect
    out = np.zeros((y1 - y0, x1 - x0, 3), dtype=np.float64)
    st = OrcStats()
    r4 = (C.c_int32 * 4)(x0, y0, x1, y1)
    rc = L.orc_render_reference_shaped(C.byref(desc), C.byref(cam), C.byref(params), thread_num, r4, out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle: " + L.orc_last_error().decode())
    return out, st.as_dict()


def write_color(pixel_color, spp):
    out = (C.c_uint8 * 3)()
    lib().orc_write_color(_d(pixel_color), spp, out)
    return tuple(out)


def hit_out(n, buf):
    if n <= 0:
        return None
    return dict(t=buf[0], p=tuple(buf[1:4]), normal=tuple(buf[4:7]), u=buf[7], v=buf[8], front_face=bool(buf[9]))


def sphere_hit(center, radius, o, d, tm=0.0, t_min=0.001, t_max=float("inf")):
    buf = (C.c_double * 10)()
    return hit_out(lib().orc_sphere_hit(_d(center), radius, _d(o), _d(d), tm, t_min, t_max, buf), buf)


def rect_hit(kaxis, abk5, o, d, t_min=0.001, t_max=float("inf")):
    buf = (C.c_double * 10)()
    return hit_out(lib().orc_rect_hit(kaxis, _d(abk5), _d(o), _d(d), t_min, t_max, buf), buf)


def world_hit(desc, o, d, tm=0.0, t_min=0.001, t_max=float("inf")):
    buf = (C.c_double * 10)()
    n = lib().orc_world_hit(C.byref(desc), _d(o), _d(d), tm, t_min, t_max, buf)
    if n < 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())
    return hit_out(n, buf)


def rng_stream(seed, pixel_index, sample_index, n):
    raw = np.zeros(n, dtype=np.uint64)
    f64 = np.zeros(n, dtype=np.float64)
    f32 = np.zeros(n, dtype=np.float32)
    lib().orc_rng_stream(seed, pixel_index, sample_index, n, raw.ctypes.data_as(C.POINTER(C.c_uint64)), f64.ctypes.data_as(C.POINTER(C.c_double)),
                         f32.ctypes.data_as(C.POINTER(C.c_float)))
    return raw, f64, f32
