//! gpu_ffi.rs — `extern "C"` bindings of include/rt_hip.h (ABI version 3) for the reference crate.
//!
//! Where it goes: `raytracer/src/gpu_ffi.rs`, with `mod gpu_ffi;` added to the module list of `main.rs:7-24`.
//! It replaces nothing by itself: docs/main_rs.patch swaps the pixel loops `main.rs:730-784` for one call.
//!
//! UNVERIFIED TEXT: this image has no Rust toolchain, so this file has never been compiled. What IS
//! checked (tests/test_abi.py::test_rust_shim_covers_the_header) is that every function the header declares is bound
//! here under the same name and that every `#[repr(C)]` struct lists the header's fields in the header's order.
//! Edition 2018 / rustc 1.60 (the reference's pin, rust-toolchain:1): no `c"..."` literals, no `unsafe extern`.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: u32 = 3;

// RtStatus
pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_UNSUPPORTED: c_int = -2;
pub const RT_ERR_DEVICE: c_int = -3;
pub const RT_ERR_NO_DEVICE: c_int = -4;
pub const RT_ERR_OOM: c_int = -5;
pub const RT_ERR_PEER: c_int = -6;        // a collective render was called off because another rank failed; the message names it

/// vec3.rs:5-8
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtVec3 { pub x: f64, pub y: f64, pub z: f64 }

/// camera.rs:6-18, field for field
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtCamera {
    pub origin: RtVec3, pub lower_left_corner: RtVec3, pub horizontal: RtVec3, pub vertical: RtVec3,
    pub u: RtVec3, pub v: RtVec3, pub w: RtVec3, pub lens_radius: f64, pub time0: f64, pub time1: f64,
}

// RtTextureKind (texture.rs)
pub const RT_TEX_SOLID: i32 = 0;
pub const RT_TEX_CHECKER: i32 = 1;
pub const RT_TEX_NOISE: i32 = 2;
pub const RT_TEX_IMAGE: i32 = 3;
#[repr(C)] #[derive(Clone, Copy)]
pub struct RtTexture { pub kind: i32, pub a: i32, pub b: i32, pub _pad: i32, pub color: RtVec3, pub scale: f64 }
/// perlin.rs:7-12
#[repr(C)]
pub struct RtPerlin { pub ranvec: [[f64; 3]; 256], pub perm_x: [u32; 256], pub perm_y: [u32; 256], pub perm_z: [u32; 256] }
/// texture.rs:99-104: RGB8, row-major
#[repr(C)] #[derive(Clone, Copy)]
pub struct RtImage { pub data: *const u8, pub width: u32, pub height: u32 }

// RtMaterialKind (material.rs)
pub const RT_MAT_LAMBERTIAN: i32 = 0;
pub const RT_MAT_METAL: i32 = 1;
pub const RT_MAT_DIELECTRIC: i32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: i32 = 3;
pub const RT_MAT_ISOTROPIC: i32 = 4;
#[repr(C)] #[derive(Clone, Copy)]
pub struct RtMaterial { pub kind: i32, pub texture: i32, pub albedo: RtVec3, pub fuzz: f64, pub ir: f64 }

// RtHittableKind: one record per Arc<dyn Hittable>
pub const RT_HIT_SPHERE: i32 = 0;
pub const RT_HIT_MOVING_SPHERE: i32 = 1;
pub const RT_HIT_XY_RECT: i32 = 2;
pub const RT_HIT_XZ_RECT: i32 = 3;
pub const RT_HIT_YZ_RECT: i32 = 4;
pub const RT_HIT_TRIANGLE: i32 = 5;
pub const RT_HIT_BOX: i32 = 6;
pub const RT_HIT_LIST: i32 = 7;
pub const RT_HIT_BVH: i32 = 8;
pub const RT_HIT_TRANSLATE: i32 = 9;
pub const RT_HIT_ROTATE_Y: i32 = 10;
pub const RT_HIT_FLIP_FACE: i32 = 11;
pub const RT_HIT_CONSTANT_MEDIUM: i32 = 12;
#[repr(C)] #[derive(Clone, Copy)]
pub struct RtHittable { pub kind: i32, pub material: i32, pub first_child: i32, pub n_children: i32, pub p: [f64; 10] }

pub const RT_BG_CONSTANT: i32 = 0;       // main.rs:692 `background`
pub const RT_BG_SKY_GRADIENT: i32 = 1;   // book-1 sky (not in the reference)
pub const RT_BVH_REFERENCE: i32 = 0;     // BVHNode::construct as intended (bvh.rs:77-130)
pub const RT_BVH_SAH: i32 = 1;           // binned SAH: same picture, fewer node visits

#[repr(C)]
pub struct RtSceneDesc {
    pub abi_version: u32, pub _pad0: u32,
    pub hittables: *const RtHittable, pub n_hittables: u64,
    pub children: *const i32, pub n_children: u64,
    pub materials: *const RtMaterial, pub n_materials: u64,
    pub textures: *const RtTexture, pub n_textures: u64,
    pub perlins: *const RtPerlin, pub n_perlins: u64,
    pub images: *const RtImage, pub n_images: u64,
    pub world: i32, pub lights: i32, pub background_mode: i32, pub bvh_builder: i32,
    pub background: RtVec3, pub bvh_seed: u64,
}

pub const RT_NAN_PER_SAMPLE: u32 = 0;
pub const RT_NAN_REFERENCE: u32 = 1;     // main.rs:146-155: the pixel SUM is scrubbed
pub const RT_FLAG_COUNTERS: u32 = 1;
pub const RT_FLAG_TIMING: u32 = 2;
pub const RT_FLAG_SAMPLE_BLOCKS: u32 = 4;
pub const RT_FLAG_FUSED: u32 = 8;         // diagnostic: the whole render by the fused per-path (tail) kernel
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtParams {
    pub width: u32, pub height: u32, pub samples_per_pixel: u32, pub max_depth: u32, pub seed: u64,
    pub nan_policy: u32, pub flags: u32, pub tile_size: u32, pub shard_index: u32, pub shard_count: u32, pub pool_slots: u32,
    pub tail_paths: u32, pub _pad: u32,
}

// RtUploadOptions.layout_flags: how a scene is laid out on the device (never what it looks like); 0 = the library's defaults
pub const RT_LAYOUT_LISTS_AS_REFERENCE: u32 = 1;
pub const RT_LAYOUT_LISTS_CULLED: u32 = 2;
pub const RT_LAYOUT_NO_MEMBER_BOXES: u32 = 4;
pub const RT_LAYOUT_MEMBER_BOXES: u32 = 8;
pub const RT_LAYOUT_CHILD_ORDER_AS_REFERENCE: u32 = 16;
pub const RT_LAYOUT_SCENE_IN_HBM: u32 = 32;
pub const RT_LAYOUT_NODES_32B: u32 = 64;
pub const RT_LAYOUT_NO_SHADE_TABLES_IN_LDS: u32 = 128;
pub const RT_LAYOUT_NO_EXTEND_TABLES_IN_LDS: u32 = 256;
pub const RT_LAYOUT_WIDE_NODES: u32 = 512;
pub const RT_LAYOUT_REFERENCE_COUNTERS: u32 = 1 | 4 | 16;   // RtStats test counts = the reference's own
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtUploadOptions {
    pub struct_bytes: u32, pub layout_flags: u32, pub lds_top_records: u32, pub octant_axes: u32, pub leaf_collapse: u32, pub list_park_cost: f32,
}

pub const RT_N_PRIM_TYPES: usize = 6;
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtStats {
    pub render_ms: f64, pub extend_ms: f64, pub shade_ms: f64, pub other_ms: f64,
    pub samples: u64, pub segments: u64, pub node_tests: u64, pub prim_tests: [u64; RT_N_PRIM_TYPES],
    pub iterations: u32, pub extend_launches: u32, pub shade_launches: u32, pub pool_slots: u32,
    pub scene_nodes: u64, pub scene_prims: u64, pub scene_bytes: u64, pub bvh_in_lds: u32, pub _pad: u32, pub debug: [u64; 8],
    pub gather_ms: f64, pub n_devices: u32, pub lds_top_nodes: u32, pub drain_ms: f64, pub drain_paths: u32, pub _pad2: u32,
}

#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtWideInfo {
    pub n_nodes: u64, pub n_leaf_entries: u64, pub n_inner_entries: u64, pub n_prims: u64, pub depth: u32, pub _pad: u32,
    pub mean_children: f64, pub mean_leaf_members: f64,
}

#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct RtCompileInfo {
    pub n_nodes: u64, pub n_box_nodes: u64, pub n_spheres: u64, pub n_moving: u64, pub n_rects: u64, pub n_tris: u64,
    pub n_media: u64, pub n_xforms: u64, pub n_lights: u64, pub n_materials: u64, pub features: u32, pub fits_lds: u32,
    pub n_first: u32, pub first: [u32; 4], pub _pad: u32,
}

pub const RT_OUT_RGB_SUM_F32: u32 = 0;
pub const RT_OUT_RGB8: u32 = 1;
pub const RT_COMM_ID_BYTES: usize = 128;

#[repr(C)] pub struct RtCtx { _p: [u8; 0] }
#[repr(C)] pub struct RtScene { _p: [u8; 0] }
#[repr(C)] pub struct RtMultiCtx { _p: [u8; 0] }
#[repr(C)] pub struct RtMultiScene { _p: [u8; 0] }

#[link(name = "rt_hip")]
extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_last_error(ctx: *const RtCtx) -> *const c_char;

    // ---- one GPU ----
    pub fn rt_ctx_create(device_id: c_int, stream: *mut c_void, out_ctx: *mut *mut RtCtx) -> c_int;
    pub fn rt_ctx_destroy(ctx: *mut RtCtx) -> c_int;
    pub fn rt_scene_upload(ctx: *mut RtCtx, desc: *const RtSceneDesc, out_scene: *mut *mut RtScene) -> c_int;
    /// the same with the device layout chosen per upload (NULL options = defaults); a threaded host cannot use process environment for that
    pub fn rt_scene_upload_ex(ctx: *mut RtCtx, desc: *const RtSceneDesc, options: *const RtUploadOptions, out_scene: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(ctx: *mut RtCtx, scene: *mut RtScene) -> c_int;
    pub fn rt_output_floats(params: *const RtParams, out_n: *mut u64) -> c_int;
    /// replaces the body of the pixel loops main.rs:731-784 (without write_color)
    pub fn rt_render(ctx: *mut RtCtx, scene: *const RtScene, cam: *const RtCamera, params: *const RtParams,
                     rgb_sum_host: *mut f32, stats: *mut RtStats) -> c_int;
    pub fn rt_render_device(ctx: *mut RtCtx, scene: *const RtScene, cam: *const RtCamera, params: *const RtParams,
                            rgb_sum_device: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_untile(params: *const RtParams, gathered: *const f32, rgb_sum: *mut f32) -> c_int;
    /// write_color (main.rs:141-169) on the device
    pub fn rt_resolve_device(ctx: *mut RtCtx, rgb_sum_device: *const c_void, width: u32, height: u32,
                             samples_per_pixel: u32, rgb8_device: *mut c_void) -> c_int;

    // ---- all GPUs of the node from this one process: what main.rs binds (docs/main_rs.patch) ----
    pub fn rt_ctx_create_multi(device_ids: *const c_int, n_devices: c_int, out_ctx: *mut *mut RtMultiCtx) -> c_int;
    pub fn rt_ctx_destroy_multi(ctx: *mut RtMultiCtx) -> c_int;
    pub fn rt_scene_upload_multi(ctx: *mut RtMultiCtx, desc: *const RtSceneDesc, out_scene: *mut *mut RtMultiScene) -> c_int;
    pub fn rt_scene_upload_multi_ex(ctx: *mut RtMultiCtx, desc: *const RtSceneDesc, options: *const RtUploadOptions,
                                    out_scene: *mut *mut RtMultiScene) -> c_int;
    pub fn rt_scene_destroy_multi(ctx: *mut RtMultiCtx, scene: *mut RtMultiScene) -> c_int;
    pub fn rt_render_multi(ctx: *mut RtMultiCtx, scene: *const RtMultiScene, cam: *const RtCamera, params: *const RtParams,
                           rgb_sum_host: *mut f32, stats: *mut RtStats) -> c_int;
    /// write_color applied on the devices: the bytes main.rs:781 stores, 3 B/pixel over xGMI
    pub fn rt_render_multi_rgb8(ctx: *mut RtMultiCtx, scene: *const RtMultiScene, cam: *const RtCamera, params: *const RtParams,
                                rgb8_host: *mut u8, stats: *mut RtStats) -> c_int;
    pub fn rt_last_error_multi(ctx: *const RtMultiCtx) -> *const c_char;

    // ---- one process per GPU (MPI / torchrun style launchers) ----
    pub fn rt_comm_unique_id(id_out: *mut u8) -> c_int;
    pub fn rt_comm_init_rank(ctx: *mut RtCtx, id: *const u8, rank: c_int, world: c_int) -> c_int;
    pub fn rt_comm_selftest(ctx: *mut RtCtx) -> c_int;
    pub fn rt_render_gather(ctx: *mut RtCtx, scene: *const RtScene, cam: *const RtCamera, params: *const RtParams,
                            output_kind: u32, frame_device: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_untile_device(ctx: *mut RtCtx, params: *const RtParams, output_kind: u32, gathered_device: *const c_void,
                            frame_device: *mut c_void) -> c_int;
    pub fn rt_untile_rgb8(params: *const RtParams, gathered: *const u8, rgb8: *mut u8) -> c_int;

    // ---- the process's ROCm runtime libraries (refuses two HIP runtimes in one process); fault injection for failure-path tests ----
    pub fn rt_runtime_libraries(out: *mut c_char, cap: u64) -> c_int;
    pub fn rt_test_fail_next_renders(ctx: *mut RtCtx, n: u32) -> c_int;
    pub fn rt_test_device_workers(n_workers: c_int, rounds: c_int) -> c_int;

    // ---- scene-compiler introspection (host only) ----
    pub fn rt_scene_compile_info(desc: *const RtSceneDesc, out: *mut RtCompileInfo) -> c_int;
    pub fn rt_scene_compile_info_ex(desc: *const RtSceneDesc, options: *const RtUploadOptions, out: *mut RtCompileInfo) -> c_int;
    pub fn rt_scene_compile_dump_ex(desc: *const RtSceneDesc, options: *const RtUploadOptions, nodes: *mut c_void, cap_nodes: u64,
                                    spheres: *mut f32, sphere_meta: *mut u32, cap_spheres: u64) -> c_int;
    pub fn rt_scene_wide_layout_check(desc: *const RtSceneDesc, out: *mut RtWideInfo) -> c_int;
    pub fn rt_scene_top_layout_check(desc: *const RtSceneDesc, max_top: u32, out_n_top: *mut u64) -> c_int;
    pub fn rt_scene_compile_dump(desc: *const RtSceneDesc, nodes: *mut c_void, cap_nodes: u64, spheres: *mut f32,
                                 sphere_meta: *mut u32, cap_spheres: u64) -> c_int;
}

/// The reference's error convention is panic (main.rs:656,762,777,779): a negative status becomes one, with the library's message.
pub unsafe fn check_multi(code: c_int, ctx: *const RtMultiCtx) {
    if code != RT_OK {
        let msg = std::ffi::CStr::from_ptr(rt_last_error_multi(ctx)).to_string_lossy().into_owned();
        panic!("rt_hip error {}: {}", code, msg);
    }
}

impl From<crate::vec3::Vec3> for RtVec3 {
    fn from(v: crate::vec3::Vec3) -> Self { RtVec3 { x: v.x(), y: v.y(), z: v.z() } }
}
