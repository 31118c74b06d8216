"""Thin object wrappers over librt_hip.so. Names follow the C ABI (include/rt_hip.h, include/rt_host.h)."""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from .build import LIB_PATH

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def lib_path():
    # RT_HIP_LIB selects a tuning build (scripts/ only); the default is the in-tree lib/librt_hip.so
    return os.environ.get("RT_HIP_LIB", LIB_PATH)


def lib():
    """Load librt_hip.so. Raises if it has not been built (python __graft_entry__.py / build.py)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RtError(A.RT_ERR_NO_DEVICE, f"{path} is missing: build it first (ray-tracer-archive_amd/build.py); "
                          "the product path has no fallback")
        _lib = A.declare(C.CDLL(path))
        if _lib.rt_abi_version() != A.RT_ABI_VERSION:
            raise RtError(A.RT_ERR_INVALID, "librt_hip.so ABI version mismatch")
    return _lib


def _check(code, ctx=None):
    if code != A.RT_OK:
        msg = lib().rt_last_error(ctx).decode() if ctx is not None else lib().rt_last_error(None).decode()
        raise RtError(code, msg)


def make_params(width, height, spp, max_depth=50, seed=1, nan_policy=A.RT_NAN_PER_SAMPLE, flags=0, tile_size=0, shard_index=0,
                shard_count=1, pool_slots=0, tail_paths=0):
    return A.RtParams(width, height, spp, max_depth, seed, nan_policy, flags, tile_size, shard_index, shard_count, pool_slots, tail_paths, 0)


def output_floats(params):
    n = C.c_uint64(0)
    _check(lib().rt_output_floats(C.byref(params), C.byref(n)))
    return n.value


def _vec(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


def camera_new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time0, time1):
    """Camera::new (camera.rs:21-59)."""
    cam = A.RtCamera()
    lib().rt_host_camera_new(_vec(lookfrom), _vec(lookat), _vec(vup), _vec([vfov, aspect_ratio, aperture, focus_dist]), time0, time1, C.byref(cam))
    return cam


def write_color(pixel_color, spp):
    """write_color (main.rs:141-169) for one pixel sum."""
    out = (C.c_uint8 * 3)()
    lib().rt_host_write_color(_vec(pixel_color), spp, out)
    return tuple(out)


def tonemap(rgb_sum, spp):
    rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float32)
    h, w, _ = rgb_sum.shape
    out = np.empty((h, w, 3), dtype=np.uint8)
    _check(lib().rt_host_tonemap(rgb_sum.ctypes.data_as(C.POINTER(C.c_float)), w, h, spp, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def write_png(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    return lib().rt_host_write_png(str(path).encode(), rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)


def write_image(path, rgb8, quality=100):
    """rt_host_write_image: creates the parent directories and encodes by extension (.jpg at `quality` — the reference's
    output/book3/image12.jpg, main.rs:653-656,791-796 — or .png)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    return lib().rt_host_write_image(str(path).encode(), rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, quality)


def untile(params, gathered):
    gathered = np.ascontiguousarray(gathered, dtype=np.float32)
    out = np.zeros((params.height, params.width, 3), dtype=np.float32)
    _check(lib().rt_untile(C.byref(params), gathered.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def untile_rgb8(params, gathered):
    gathered = np.ascontiguousarray(gathered, dtype=np.uint8)
    out = np.zeros((params.height, params.width, 3), dtype=np.uint8)
    _check(lib().rt_untile_rgb8(C.byref(params), gathered.ctypes.data_as(C.POINTER(C.c_uint8)), out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def comm_unique_id():
    """rt_comm_unique_id: the 128 bytes rank 0 hands to every other rank (over the launcher's own channel)."""
    buf = (C.c_uint8 * A.RT_COMM_ID_BYTES)()
    _check(lib().rt_comm_unique_id(buf))
    return bytes(buf)


def compile_info(desc, layout_flags=0):
    """rt_scene_compile_info[_ex]: what the scene compiler makes of a graph (host only)."""
    info = A.RtCompileInfo()
    opt = upload_options(layout_flags)
    _check(lib().rt_scene_compile_info_ex(C.byref(desc), C.byref(opt), C.byref(info)))
    out = {n: getattr(info, n) for n, _ in info._fields_ if n not in ("first", "_pad")}
    out["first"] = [int(info.first[k]) for k in range(info.n_first)]
    return out


def wide_layout_check(desc):
    """rt_scene_wide_layout_check: builds and verifies the 8-wide tree of a static BVH on the host; returns its statistics."""
    info = A.RtWideInfo()
    _check(lib().rt_scene_wide_layout_check(C.byref(desc), C.byref(info)))
    return {n: getattr(info, n) for n, _ in info._fields_ if not n.startswith("_")}


def compile_dump(desc, layout_flags=0):
    """rt_scene_compile_dump[_ex]: (nodes structured array, spheres (n,4) f32, sphere_meta u32)."""
    info = compile_info(desc, layout_flags)
    opt = upload_options(layout_flags)
    node_t = np.dtype([("mn", np.float32, 3), ("skip", np.uint32), ("mx", np.float32, 3), ("leaf", np.uint32)])
    nodes = np.zeros(info["n_nodes"], dtype=node_t)
    n_s = max(1, info["n_spheres"] + info["n_media"])   # media boundaries add private spheres
    spheres = np.zeros((n_s, 4), dtype=np.float32)
    meta = np.zeros(n_s, dtype=np.uint32)
    _check(lib().rt_scene_compile_dump_ex(C.byref(desc), C.byref(opt), nodes.ctypes.data_as(C.c_void_p), len(nodes), spheres.ctypes.data_as(C.POINTER(C.c_float)),
                                          meta.ctypes.data_as(C.POINTER(C.c_uint32)), n_s))
    return nodes, spheres[:info["n_spheres"]], meta[:info["n_spheres"]]


class HostScene:
    """A scene function of main.rs, built by the C++ host mirror (host/rt_host.hpp)."""

    def __init__(self, name, scene_seed=1, arg0=0, arg1=0, image=None):
        self._h = C.c_void_p()
        self._image = None
        ip, iw, ih = None, 0, 0
        if image is not None:
            self._image = np.ascontiguousarray(image, dtype=np.uint8)
            ih, iw, _ = self._image.shape
            ip = self._image.ctypes.data_as(C.POINTER(C.c_uint8))
        _check(lib().rt_host_scene_create(name.encode(), scene_seed, arg0, arg1, ip, iw, ih, C.byref(self._h)))
        self.name = name

    @property
    def desc(self):
        return lib().rt_host_scene_desc(self._h).contents

    def camera(self, aspect_ratio):
        cam = A.RtCamera()
        _check(lib().rt_host_scene_camera(self._h, float(aspect_ratio), C.byref(cam)))
        return cam

    def close(self):
        if self._h:
            lib().rt_host_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upload_options(layout_flags=0, lds_top_records=0, octant_axes=0, leaf_collapse=0, list_park_cost=0.0):
    """RtUploadOptions (include/rt_hip.h): how the scene is laid out on the device; never what it looks like."""
    return A.RtUploadOptions(C.sizeof(A.RtUploadOptions), layout_flags, lds_top_records, octant_axes, leaf_collapse, list_park_cost)


def runtime_libraries():
    """rt_runtime_libraries: (paths of the mapped libamdhip64 / libhsa-runtime64 / librccl objects, ok) — ok is False when one is mapped twice."""
    buf = C.create_string_buffer(8192)
    code = lib().rt_runtime_libraries(buf, len(buf))
    return [p for p in buf.value.decode().split("\n") if p], code == A.RT_OK


class Scene:
    def __init__(self, ctx, desc, options=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        _check(lib().rt_scene_upload_ex(ctx._h, C.byref(desc), C.byref(options) if options is not None else None, C.byref(self._h)), ctx._h)

    def close(self):
        if self._h and self.ctx._h:
            lib().rt_scene_destroy(self.ctx._h, self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """rt_ctx_create: one per (process, device, stream)."""

    def __init__(self, device_id=0, stream=None):
        self._h = C.c_void_p()
        _check(lib().rt_ctx_create(device_id, C.c_void_p(stream) if stream else None, C.byref(self._h)))

    def upload(self, desc, layout_flags=0, **more):
        """rt_scene_upload_ex; layout_flags = RT_LAYOUT_* (A.RT_LAYOUT_REFERENCE_COUNTERS: the layout whose test counts are the reference's)."""
        return Scene(self, desc, upload_options(layout_flags, **more) if (layout_flags or more) else None)

    def fail_next_renders(self, n):
        """rt_test_fail_next_renders: fault injection for the failure-path tests."""
        _check(lib().rt_test_fail_next_renders(self._h, n), self._h)

    def render(self, scene, cam, params):
        """rt_render: returns (rgb_sum float32 array, stats dict). Full frame -> (H, W, 3); sharded -> flat."""
        n = output_floats(params)
        out = np.empty(n, dtype=np.float32)
        st = A.RtStats()
        _check(lib().rt_render(self._h, scene._h, C.byref(cam), C.byref(params), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st)), self._h)
        if params.shard_count <= 1:
            out = out.reshape(params.height, params.width, 3)
        return out, st.as_dict()

    def render_device(self, scene, cam, params, device_ptr):
        """rt_render_device: result stays in caller-owned device memory (e.g. a torch tensor's data_ptr())."""
        st = A.RtStats()
        _check(lib().rt_render_device(self._h, scene._h, C.byref(cam), C.byref(params), C.c_void_p(device_ptr), C.byref(st)), self._h)
        return st.as_dict()

    # ---- one process per GPU: RCCL communicator on this context (rt_multi.cpp) ----
    def comm_init_rank(self, unique_id, rank, world):
        buf = (C.c_uint8 * A.RT_COMM_ID_BYTES)(*unique_id)
        _check(lib().rt_comm_init_rank(self._h, buf, rank, world), self._h)

    def comm_selftest(self):
        _check(lib().rt_comm_selftest(self._h), self._h)

    def render_gather(self, scene, cam, params, output_kind=A.RT_OUT_RGB_SUM_F32, frame_ptr=None):
        """rt_render_gather (collective): this rank's shard is rendered and sent to rank 0, which leaves the full frame at frame_ptr."""
        st = A.RtStats()
        _check(lib().rt_render_gather(self._h, scene._h, C.byref(cam), C.byref(params), output_kind, C.c_void_p(frame_ptr) if frame_ptr else None, C.byref(st)),
               self._h)
        return st.as_dict()

    def untile_device(self, params, output_kind, gathered_ptr, frame_ptr):
        _check(lib().rt_untile_device(self._h, C.byref(params), output_kind, C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr)), self._h)

    def resolve_device(self, rgb_sum_ptr, width, height, spp, rgb8_ptr):
        _check(lib().rt_resolve_device(self._h, C.c_void_p(rgb_sum_ptr), width, height, spp, C.c_void_p(rgb8_ptr)), self._h)

    def close(self):
        if self._h:
            lib().rt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiScene:
    def __init__(self, mctx, desc, options=None):
        self.mctx = mctx
        self._h = C.c_void_p()
        code = lib().rt_scene_upload_multi_ex(mctx._h, C.byref(desc), C.byref(options) if options is not None else None, C.byref(self._h))
        if code != A.RT_OK:
            raise RtError(code, lib().rt_last_error_multi(mctx._h).decode())

    def close(self):
        if self._h and self.mctx._h:
            lib().rt_scene_destroy_multi(self.mctx._h, self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiContext:
    """rt_ctx_create_multi: ONE process driving n GPUs (what the reference's single-process host binds). The framebuffer is
    tile-sharded over the devices and gathered on the first one with RCCL inside rt_render_multi."""

    def __init__(self, device_ids):
        self._h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        _check(lib().rt_ctx_create_multi(ids, len(device_ids), C.byref(self._h)))
        self.n = len(device_ids)

    def upload(self, desc, layout_flags=0, **more):
        return MultiScene(self, desc, upload_options(layout_flags, **more) if (layout_flags or more) else None)

    def _render(self, fn, scene, cam, params, out, ptr_t):
        st = A.RtStats()
        code = fn(self._h, scene._h, C.byref(cam), C.byref(params), out.ctypes.data_as(C.POINTER(ptr_t)), C.byref(st))
        if code != A.RT_OK:
            raise RtError(code, lib().rt_last_error_multi(self._h).decode())
        return out, st.as_dict()

    def render(self, scene, cam, params):
        """rt_render_multi: full frame of f32 RGB sums on the host."""
        return self._render(lib().rt_render_multi, scene, cam, params, np.empty((params.height, params.width, 3), dtype=np.float32), C.c_float)

    def render_rgb8(self, scene, cam, params):
        """rt_render_multi_rgb8: write_color applied per shard on the devices, RGB8 gathered (3 B/pixel over xGMI)."""
        return self._render(lib().rt_render_multi_rgb8, scene, cam, params, np.empty((params.height, params.width, 3), dtype=np.uint8), C.c_uint8)

    def close(self):
        if self._h:
            lib().rt_ctx_destroy_multi(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
