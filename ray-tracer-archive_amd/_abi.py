"""ctypes mirror of include/rt_hip.h and include/rt_host.h (struct layouts must match the headers)."""
import ctypes as C

RT_ABI_VERSION = 3
RT_N_PRIM_TYPES = 6

RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_DEVICE, RT_ERR_NO_DEVICE, RT_ERR_OOM, RT_ERR_PEER = 0, -1, -2, -3, -4, -5, -6

RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_NOISE, RT_TEX_IMAGE = 0, 1, 2, 3
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = 0, 1, 2, 3, 4
(RT_HIT_SPHERE, RT_HIT_MOVING_SPHERE, RT_HIT_XY_RECT, RT_HIT_XZ_RECT, RT_HIT_YZ_RECT, RT_HIT_TRIANGLE, RT_HIT_BOX, RT_HIT_LIST,
 RT_HIT_BVH, RT_HIT_TRANSLATE, RT_HIT_ROTATE_Y, RT_HIT_FLIP_FACE, RT_HIT_CONSTANT_MEDIUM) = range(13)
RT_BG_CONSTANT, RT_BG_SKY_GRADIENT = 0, 1
RT_BVH_REFERENCE, RT_BVH_SAH = 0, 1
RT_NAN_PER_SAMPLE, RT_NAN_REFERENCE = 0, 1
RT_FLAG_COUNTERS, RT_FLAG_TIMING, RT_FLAG_SAMPLE_BLOCKS, RT_FLAG_FUSED = 1, 2, 4, 8
RT_OUT_RGB_SUM_F32, RT_OUT_RGB8 = 0, 1
RT_COMM_ID_BYTES = 128
# RtUploadOptions.layout_flags
(RT_LAYOUT_LISTS_AS_REFERENCE, RT_LAYOUT_LISTS_CULLED, RT_LAYOUT_NO_MEMBER_BOXES, RT_LAYOUT_MEMBER_BOXES, RT_LAYOUT_CHILD_ORDER_AS_REFERENCE,
 RT_LAYOUT_SCENE_IN_HBM, RT_LAYOUT_NODES_32B, RT_LAYOUT_NO_SHADE_TABLES_IN_LDS, RT_LAYOUT_NO_EXTEND_TABLES_IN_LDS, RT_LAYOUT_WIDE_NODES) = (1 << k for k in range(10))
RT_LAYOUT_REFERENCE_COUNTERS = RT_LAYOUT_LISTS_AS_REFERENCE | RT_LAYOUT_NO_MEMBER_BOXES | RT_LAYOUT_CHILD_ORDER_AS_REFERENCE


class RtVec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class RtCamera(C.Structure):
    _fields_ = [("origin", RtVec3), ("lower_left_corner", RtVec3), ("horizontal", RtVec3), ("vertical", RtVec3),
                ("u", RtVec3), ("v", RtVec3), ("w", RtVec3), ("lens_radius", C.c_double), ("time0", C.c_double), ("time1", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("_pad", C.c_int32), ("color", RtVec3), ("scale", C.c_double)]


class RtPerlin(C.Structure):
    _fields_ = [("ranvec", (C.c_double * 3) * 256), ("perm_x", C.c_uint32 * 256), ("perm_y", C.c_uint32 * 256), ("perm_z", C.c_uint32 * 256)]


class RtImage(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", RtVec3), ("fuzz", C.c_double), ("ir", C.c_double)]


class RtHittable(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("first_child", C.c_int32), ("n_children", C.c_int32), ("p", C.c_double * 10)]


class RtSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("_pad0", C.c_uint32),
                ("hittables", C.POINTER(RtHittable)), ("n_hittables", C.c_uint64),
                ("children", C.POINTER(C.c_int32)), ("n_children", C.c_uint64),
                ("materials", C.POINTER(RtMaterial)), ("n_materials", C.c_uint64),
                ("textures", C.POINTER(RtTexture)), ("n_textures", C.c_uint64),
                ("perlins", C.POINTER(RtPerlin)), ("n_perlins", C.c_uint64),
                ("images", C.POINTER(RtImage)), ("n_images", C.c_uint64),
                ("world", C.c_int32), ("lights", C.c_int32), ("background_mode", C.c_int32), ("bvh_builder", C.c_int32),
                ("background", RtVec3), ("bvh_seed", C.c_uint64)]


class RtParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("seed", C.c_uint64), ("nan_policy", C.c_uint32), ("flags", C.c_uint32),
                ("tile_size", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("pool_slots", C.c_uint32),
                ("tail_paths", C.c_uint32), ("_pad", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("render_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double), ("other_ms", C.c_double),
                ("samples", C.c_uint64), ("segments", C.c_uint64), ("node_tests", C.c_uint64), ("prim_tests", C.c_uint64 * RT_N_PRIM_TYPES),
                ("iterations", C.c_uint32), ("extend_launches", C.c_uint32), ("shade_launches", C.c_uint32), ("pool_slots", C.c_uint32),
                ("scene_nodes", C.c_uint64), ("scene_prims", C.c_uint64), ("scene_bytes", C.c_uint64), ("bvh_in_lds", C.c_uint32), ("_pad", C.c_uint32), ("debug", C.c_uint64 * 8),
                ("gather_ms", C.c_double), ("n_devices", C.c_uint32), ("lds_top_nodes", C.c_uint32),
                ("drain_ms", C.c_double), ("drain_paths", C.c_uint32), ("_pad2", C.c_uint32)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            if name.startswith("_"):
                continue
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


class RtCompileInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_box_nodes", C.c_uint64), ("n_spheres", C.c_uint64), ("n_moving", C.c_uint64), ("n_rects", C.c_uint64),
                ("n_tris", C.c_uint64), ("n_media", C.c_uint64), ("n_xforms", C.c_uint64), ("n_lights", C.c_uint64), ("n_materials", C.c_uint64),
                ("features", C.c_uint32), ("fits_lds", C.c_uint32), ("n_first", C.c_uint32), ("first", C.c_uint32 * 4), ("_pad", C.c_uint32)]


class RtUploadOptions(C.Structure):
    _fields_ = [("struct_bytes", C.c_uint32), ("layout_flags", C.c_uint32), ("lds_top_records", C.c_uint32), ("octant_axes", C.c_uint32),
                ("leaf_collapse", C.c_uint32), ("list_park_cost", C.c_float)]


class RtWideInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_leaf_entries", C.c_uint64), ("n_inner_entries", C.c_uint64), ("n_prims", C.c_uint64),
                ("depth", C.c_uint32), ("_pad", C.c_uint32), ("mean_children", C.c_double), ("mean_leaf_members", C.c_double)]


# every symbol include/rt_hip.h and include/rt_host.h declare
RT_HIP_SYMBOLS = ["rt_ctx_create", "rt_ctx_destroy", "rt_scene_upload", "rt_scene_destroy", "rt_output_floats", "rt_render",
                  "rt_render_device", "rt_untile", "rt_resolve_device", "rt_last_error", "rt_abi_version", "rt_scene_compile_info",
                  "rt_scene_compile_dump", "rt_ctx_create_multi", "rt_ctx_destroy_multi", "rt_scene_upload_multi", "rt_scene_destroy_multi",
                  "rt_render_multi", "rt_render_multi_rgb8", "rt_last_error_multi", "rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_selftest",
                  "rt_render_gather", "rt_untile_rgb8", "rt_untile_device", "rt_scene_top_layout_check", "rt_scene_upload_ex", "rt_scene_upload_multi_ex",
                  "rt_runtime_libraries", "rt_test_fail_next_renders", "rt_test_device_workers", "rt_scene_compile_info_ex", "rt_scene_compile_dump_ex", "rt_scene_wide_layout_check"]
RT_HOST_SYMBOLS = ["rt_host_scene_create", "rt_host_scene_desc", "rt_host_scene_camera", "rt_host_scene_destroy", "rt_host_camera_new",
                   "rt_host_write_color", "rt_host_tonemap", "rt_host_write_png", "rt_host_write_jpeg", "rt_host_write_image"]


def declare(lib):
    """Attach argtypes/restypes to a loaded librt_hip.so."""
    vp, i32, u32, u64, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
    P = C.POINTER
    lib.rt_abi_version.restype = u32
    lib.rt_abi_version.argtypes = []
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_ctx_create.restype = i32
    lib.rt_ctx_create.argtypes = [i32, vp, P(vp)]
    lib.rt_ctx_destroy.restype = i32
    lib.rt_ctx_destroy.argtypes = [vp]
    lib.rt_scene_upload.restype = i32
    lib.rt_scene_upload.argtypes = [vp, P(RtSceneDesc), P(vp)]
    lib.rt_scene_upload_ex.restype = i32
    lib.rt_scene_upload_ex.argtypes = [vp, P(RtSceneDesc), P(RtUploadOptions), P(vp)]
    lib.rt_scene_upload_multi_ex.restype = i32
    lib.rt_scene_upload_multi_ex.argtypes = [vp, P(RtSceneDesc), P(RtUploadOptions), P(vp)]
    lib.rt_runtime_libraries.restype = i32
    lib.rt_runtime_libraries.argtypes = [C.c_char_p, u64]
    lib.rt_test_fail_next_renders.restype = i32
    lib.rt_test_fail_next_renders.argtypes = [vp, u32]
    lib.rt_test_device_workers.restype = i32
    lib.rt_test_device_workers.argtypes = [i32, i32]
    lib.rt_scene_destroy.restype = i32
    lib.rt_scene_destroy.argtypes = [vp, vp]
    lib.rt_output_floats.restype = i32
    lib.rt_output_floats.argtypes = [P(RtParams), P(u64)]
    lib.rt_render.restype = i32
    lib.rt_render.argtypes = [vp, vp, P(RtCamera), P(RtParams), P(C.c_float), P(RtStats)]
    lib.rt_render_device.restype = i32
    lib.rt_render_device.argtypes = [vp, vp, P(RtCamera), P(RtParams), vp, P(RtStats)]
    lib.rt_untile.restype = i32
    lib.rt_untile.argtypes = [P(RtParams), P(C.c_float), P(C.c_float)]
    lib.rt_resolve_device.restype = i32
    lib.rt_resolve_device.argtypes = [vp, vp, u32, u32, u32, vp]
    lib.rt_scene_compile_info.restype = i32
    lib.rt_scene_compile_info.argtypes = [P(RtSceneDesc), P(RtCompileInfo)]
    lib.rt_scene_compile_info_ex.restype = i32
    lib.rt_scene_compile_info_ex.argtypes = [P(RtSceneDesc), P(RtUploadOptions), P(RtCompileInfo)]
    lib.rt_scene_compile_dump_ex.restype = i32
    lib.rt_scene_compile_dump_ex.argtypes = [P(RtSceneDesc), P(RtUploadOptions), vp, u64, P(C.c_float), P(u32), u64]
    lib.rt_scene_wide_layout_check.restype = i32
    lib.rt_scene_wide_layout_check.argtypes = [P(RtSceneDesc), P(RtWideInfo)]
    lib.rt_scene_compile_dump.restype = i32
    lib.rt_scene_compile_dump.argtypes = [P(RtSceneDesc), vp, u64, P(C.c_float), P(u32), u64]
    lib.rt_untile_rgb8.restype = i32
    lib.rt_untile_rgb8.argtypes = [P(RtParams), P(C.c_uint8), P(C.c_uint8)]
    lib.rt_ctx_create_multi.restype = i32
    lib.rt_ctx_create_multi.argtypes = [P(C.c_int), i32, P(vp)]
    lib.rt_ctx_destroy_multi.restype = i32
    lib.rt_ctx_destroy_multi.argtypes = [vp]
    lib.rt_scene_upload_multi.restype = i32
    lib.rt_scene_upload_multi.argtypes = [vp, P(RtSceneDesc), P(vp)]
    lib.rt_scene_destroy_multi.restype = i32
    lib.rt_scene_destroy_multi.argtypes = [vp, vp]
    lib.rt_render_multi.restype = i32
    lib.rt_render_multi.argtypes = [vp, vp, P(RtCamera), P(RtParams), P(C.c_float), P(RtStats)]
    lib.rt_render_multi_rgb8.restype = i32
    lib.rt_render_multi_rgb8.argtypes = [vp, vp, P(RtCamera), P(RtParams), P(C.c_uint8), P(RtStats)]
    lib.rt_last_error_multi.restype = C.c_char_p
    lib.rt_last_error_multi.argtypes = [vp]
    lib.rt_comm_unique_id.restype = i32
    lib.rt_comm_unique_id.argtypes = [P(C.c_uint8)]
    lib.rt_comm_init_rank.restype = i32
    lib.rt_comm_init_rank.argtypes = [vp, P(C.c_uint8), i32, i32]
    lib.rt_comm_selftest.restype = i32
    lib.rt_comm_selftest.argtypes = [vp]
    lib.rt_render_gather.restype = i32
    lib.rt_render_gather.argtypes = [vp, vp, P(RtCamera), P(RtParams), u32, vp, P(RtStats)]
    lib.rt_untile_device.restype = i32
    lib.rt_untile_device.argtypes = [vp, P(RtParams), u32, vp, vp]
    lib.rt_scene_top_layout_check.restype = i32
    lib.rt_scene_top_layout_check.argtypes = [P(RtSceneDesc), u32, P(u64)]
    lib.rt_host_scene_create.restype = i32
    lib.rt_host_scene_create.argtypes = [C.c_char_p, u64, u64, u64, P(C.c_uint8), u32, u32, P(vp)]
    lib.rt_host_scene_desc.restype = P(RtSceneDesc)
    lib.rt_host_scene_desc.argtypes = [vp]
    lib.rt_host_scene_camera.restype = i32
    lib.rt_host_scene_camera.argtypes = [vp, f64, P(RtCamera)]
    lib.rt_host_scene_destroy.restype = None
    lib.rt_host_scene_destroy.argtypes = [vp]
    lib.rt_host_camera_new.restype = None
    lib.rt_host_camera_new.argtypes = [P(f64), P(f64), P(f64), P(f64), f64, f64, P(RtCamera)]
    lib.rt_host_write_color.restype = None
    lib.rt_host_write_color.argtypes = [P(f64), u32, P(C.c_uint8)]
    lib.rt_host_tonemap.restype = i32
    lib.rt_host_tonemap.argtypes = [P(C.c_float), u32, u32, u32, P(C.c_uint8)]
    lib.rt_host_write_png.restype = i32
    lib.rt_host_write_png.argtypes = [C.c_char_p, P(C.c_uint8), u32, u32]
    lib.rt_host_write_jpeg.restype = i32
    lib.rt_host_write_jpeg.argtypes = [C.c_char_p, P(C.c_uint8), u32, u32, i32]
    lib.rt_host_write_image.restype = i32
    lib.rt_host_write_image.argtypes = [C.c_char_p, P(C.c_uint8), u32, u32, i32]
    return lib
