/*
 * rt_hip.h — C ABI of the MI355X path-tracing hot path.
 *
 * This is the drop-in boundary for the reference renderer's per-pixel sample loop
 * (reference: raytracer/src/main.rs:730-784 — the two pixel loops, minus write_color and the
 * pixel store). Scene construction (main.rs:668-718), tone-map (main.rs:141-169) and image encode
 * (main.rs:791-796) stay on the host side of this boundary.
 *
 * A scene crosses the boundary as the reference's own object graph, serialised into flat arrays:
 * one RtHittable per `Arc<dyn Hittable>` (hittable.rs:51-60), one RtMaterial per
 * `Arc<dyn Material>` (material.rs:11-21), one RtTexture per `Arc<dyn Texture>` (texture.rs:7-9).
 * The library compiles that graph into its device layout (threaded BVH + per-type primitive
 * arrays) at upload; the caller never sees device structures.
 *
 * Plain C: pointers and sizes only, no C++ or torch types. All geometry is f64 at the boundary
 * (the reference is f64 throughout, vec3.rs:5-8); the device path computes in f32.
 *
 * Error convention: every entry point returns 0 on success or a negative RtStatus; a message is
 * available from rt_last_error(). Nothing aborts and no C++ exception crosses the boundary
 * (the reference panics instead: main.rs:656,762,777,779).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3u

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument / malformed scene graph */
    RT_ERR_UNSUPPORTED = -2,  /* graph shape the device compiler does not handle */
    RT_ERR_DEVICE = -3,       /* HIP runtime error (message has hipGetErrorString) */
    RT_ERR_NO_DEVICE = -4,    /* no usable GPU: the product path has no CPU fallback */
    RT_ERR_OOM = -5,
    RT_ERR_PEER = -6          /* a collective render (rt_render_gather / rt_render_multi) was called off because ANOTHER rank failed
                                 its part; the message names the rank. Every rank of the collective returns — none waits for a
                                 shard that will not come (the reference's convention is to panic: main.rs:762,777) */
} RtStatus;

/* vec3.rs:5-8 */
typedef struct RtVec3 { double x, y, z; } RtVec3;

/* camera.rs:6-18, field for field. Filled by the host-side Camera::new (camera.rs:21-59). */
typedef struct RtCamera {
    RtVec3 origin;
    RtVec3 lower_left_corner;
    RtVec3 horizontal;
    RtVec3 vertical;
    RtVec3 u, v, w;
    double lens_radius;
    double time0, time1;
} RtCamera;

/* texture.rs */
typedef enum RtTextureKind {
    RT_TEX_SOLID = 0,    /* texture.rs:12-38  */
    RT_TEX_CHECKER = 1,  /* texture.rs:41-69  */
    RT_TEX_NOISE = 2,    /* texture.rs:72-96  */
    RT_TEX_IMAGE = 3     /* texture.rs:99-140 */
} RtTextureKind;

typedef struct RtTexture {
    int32_t kind;
    int32_t a;      /* checker: `even` texture id; noise: perlin id; image: image id (-1 = empty) */
    int32_t b;      /* checker: `odd` texture id */
    int32_t _pad;
    RtVec3 color;   /* solid: color_value */
    double scale;   /* noise: scale */
} RtTexture;

/* perlin.rs:7-12 — tables are built on the host (perlin.rs:14-25,53-66) */
typedef struct RtPerlin {
    double ranvec[256][3];
    uint32_t perm_x[256];
    uint32_t perm_y[256];
    uint32_t perm_z[256];
} RtPerlin;

/* texture.rs:99-104 — RGB8, row-major, bytes_per_scanline = 3*width */
typedef struct RtImage {
    const uint8_t* data;
    uint32_t width;
    uint32_t height;
} RtImage;

/* material.rs */
typedef enum RtMaterialKind {
    RT_MAT_LAMBERTIAN = 0,     /* material.rs:23-72   */
    RT_MAT_METAL = 1,          /* material.rs:74-108  */
    RT_MAT_DIELECTRIC = 2,     /* material.rs:110-156 */
    RT_MAT_DIFFUSE_LIGHT = 3,  /* material.rs:158-191 */
    RT_MAT_ISOTROPIC = 4       /* material.rs:193-220 (commented out in the reference: spec only) */
} RtMaterialKind;

typedef struct RtMaterial {
    int32_t kind;
    int32_t texture;  /* lambertian albedo / diffuse-light emit / isotropic albedo: texture id */
    RtVec3 albedo;    /* metal */
    double fuzz;      /* metal; the constructor clamps to <= 1 (material.rs:90) */
    double ir;        /* dielectric */
} RtMaterial;

/* One record per `Arc<dyn Hittable>` of the reference graph. */
typedef enum RtHittableKind {
    RT_HIT_SPHERE = 0,          /* sphere.rs:11-24         p = center[3], radius                     */
    RT_HIT_MOVING_SPHERE = 1,   /* moving_sphere.rs:8-34   p = center0[3], center1[3], time0, time1, radius */
    RT_HIT_XY_RECT = 2,         /* aarect.rs:10-29         p = x0, x1, y0, y1, k                      */
    RT_HIT_XZ_RECT = 3,         /* aarect.rs:60-79         p = x0, x1, z0, z1, k                      */
    RT_HIT_YZ_RECT = 4,         /* aarect.rs:129-148       p = y0, y1, z0, z1, k                      */
    RT_HIT_TRIANGLE = 5,        /* not in the reference (README.md:151-153): p = v0[3], v1[3], v2[3]  */
    RT_HIT_BOX = 6,             /* boxes.rs:11-75          p = p0[3], p1[3]  (six rects, boxes.rs order) */
    RT_HIT_LIST = 7,            /* hittable_list.rs:11-36  children                                   */
    RT_HIT_BVH = 8,             /* bvh.rs:10-14,74         children, p = time0, time1                 */
    RT_HIT_TRANSLATE = 9,       /* hittable.rs:62-74       first_child = child id, p = offset[3]      */
    RT_HIT_ROTATE_Y = 10,       /* hittable.rs:98-145      first_child = child id, p = angle (degrees) */
    RT_HIT_FLIP_FACE = 11,      /* hittable.rs:183-193     first_child = child id                     */
    RT_HIT_CONSTANT_MEDIUM = 12 /* constant_medium.rs:9-29 (commented spec) first_child = boundary id,
                                   material = phase function (isotropic), p = density                 */
} RtHittableKind;

typedef struct RtHittable {
    int32_t kind;
    int32_t material;      /* material id for primitives, boxes and media; -1 otherwise */
    int32_t first_child;   /* LIST/BVH: offset into children[]; wrappers/medium: child hittable id */
    int32_t n_children;    /* LIST/BVH: number of children; wrappers/medium: 1; primitives: 0 */
    double p[10];
} RtHittable;

typedef enum RtBackgroundMode {
    RT_BG_CONSTANT = 0,     /* main.rs:692 `background` colour returned on a miss (main.rs:74-76) */
    RT_BG_SKY_GRADIENT = 1  /* book-1 sky: (1-t)*white + t*background, t = 0.5*(unit(d).y + 1);
                               not in the reference (it only has the constant colour) */
} RtBackgroundMode;

typedef enum RtBvhBuilder {
    RT_BVH_REFERENCE = 0,  /* BVHNode::construct as intended (bvh.rs:77-130): random axis, median split */
    RT_BVH_SAH = 1         /* binned surface-area heuristic: same pictures (a BVH only culls), far fewer
                              node visits on large scenes; SURVEY.md 8(f) rank 1 */
} RtBvhBuilder;

typedef struct RtSceneDesc {
    uint32_t abi_version;   /* RT_ABI_VERSION */
    uint32_t _pad0;
    const RtHittable* hittables;  uint64_t n_hittables;
    const int32_t*    children;   uint64_t n_children;
    const RtMaterial* materials;  uint64_t n_materials;
    const RtTexture*  textures;   uint64_t n_textures;
    const RtPerlin*   perlins;    uint64_t n_perlins;
    const RtImage*    images;     uint64_t n_images;
    int32_t world;            /* root hittable id (main.rs:668 `world`) */
    int32_t lights;           /* root of the lights list (main.rs:669-686) or -1: no lights, the
                                 integrator then samples CosinePdf only (book-1/2 behaviour) */
    int32_t background_mode;  /* RtBackgroundMode */
    int32_t bvh_builder;      /* RtBvhBuilder: how RT_HIT_BVH objects are built on the device side */
    RtVec3 background;        /* constant colour, or the sky gradient's far colour (0.5,0.7,1.0) */
    uint64_t bvh_seed;        /* seeds the per-node axis draw of BVHNode::construct (bvh.rs:87) */
} RtSceneDesc;

typedef enum RtNanPolicy {
    RT_NAN_PER_SAMPLE = 0,  /* a non-finite sample contributes 0 (documented deviation) */
    RT_NAN_REFERENCE = 1    /* samples are summed as they are; write_color scrubs the pixel SUM
                               (main.rs:146-155): one NaN sample blacks the pixel */
} RtNanPolicy;

enum {
    RT_FLAG_COUNTERS = 1u,  /* count AABB tests and primitive tests on the device (slower) */
    RT_FLAG_TIMING = 2u,    /* bracket every kernel launch with HIP events (RtStats *_ms) */
    RT_FLAG_SAMPLE_BLOCKS = 4u, /* work items of 16 consecutive samples of a pixel, as for images of 2^32 - 2^28 samples and more
                                   (default below that: one sample per item). Per-pixel sums then differ in the last bits. */
    RT_FLAG_FUSED = 8u          /* diagnostic: the whole render by the fused per-path kernel that normally carries only the tail (one
                                   lane per path from first ray to last bounce). Bit-identical frame, slower. */
};

typedef struct RtParams {
    uint32_t width, height;       /* main.rs:660-661 */
    uint32_t samples_per_pixel;   /* main.rs:662 */
    uint32_t max_depth;           /* main.rs:663 */
    uint64_t seed;                /* render seed: output is a pure function of (scene, camera, params) */
    uint32_t nan_policy;          /* RtNanPolicy */
    uint32_t flags;               /* RT_FLAG_* */
    /* framebuffer sharding (one process per GPU): the image is cut into tile_size x tile_size
       tiles, numbered row-major; this call renders tiles t with t % shard_count == shard_index.
       shard_count <= 1 renders the whole image. */
    uint32_t tile_size;           /* 0 = default (32) */
    uint32_t shard_index;
    uint32_t shard_count;
    uint32_t pool_slots;          /* paths in flight; 0 = one per work item, up to 2^28 and to 70 % of the free device memory */
    uint32_t tail_paths;          /* once at most this many paths are alive, ONE launch of the per-path kernel carries each of them to its end
                                     instead of one launch pair per bounce (same frame bit for bit). 0 = default (2^18); 1 = never */
    uint32_t _pad;
} RtParams;

#define RT_N_PRIM_TYPES 6  /* sphere, moving sphere, rect, triangle, medium, instance transform */

typedef struct RtStats {
    double render_ms;         /* first launch to framebuffer resident at the destination */
    double extend_ms;         /* sum of traversal-kernel durations (HIP events; RT_FLAG_TIMING) */
    double shade_ms;          /* sum of shade-kernel durations */
    double other_ms;          /* generate / resolve kernels */
    uint64_t samples;         /* camera paths traced */
    uint64_t segments;        /* ray segments traced (world.hit calls, main.rs:74) */
    uint64_t node_tests;      /* box tests the device made (RT_FLAG_COUNTERS). They are the reference's Aabb::hit evaluations when the scene was
                                 uploaded with RT_LIST_CULL=0 and RT_OCTANT_ORDER=0 in the environment; the default layouts cull HittableList
                                 members behind boxes of their own and visit BVH children near-first: same hits, fewer tests */
    uint64_t prim_tests[RT_N_PRIM_TYPES];   /* primitive hit() evaluations, by kind (a Box counts its six rects) */
    uint32_t iterations;      /* wavefront iterations */
    uint32_t extend_launches;
    uint32_t shade_launches;
    uint32_t pool_slots;
    uint64_t scene_nodes;     /* threaded-BVH nodes on the device */
    uint64_t scene_prims;
    uint64_t scene_bytes;     /* device bytes of nodes + primitives */
    uint32_t bvh_in_lds;      /* 1 if the whole node/primitive set is staged in LDS */
    uint32_t _pad;
    uint64_t debug[8];        /* diagnostic builds only (in-kernel cycle stamps); 0 otherwise */
    double gather_ms;         /* multi-GPU: RCCL gather + untile on the root device (HIP events on the root's stream) */
    uint32_t n_devices;       /* GPUs that took part (1 for rt_render / rt_render_device) */
    uint32_t lds_top_nodes;   /* node records of the top of the tree staged in LDS when the whole scene does not fit (0 otherwise) */
    double drain_ms;          /* duration of the fused kernel that carries the last paths to their end (RT_FLAG_TIMING) */
    uint32_t drain_paths;     /* upper bound of the paths handed to it (0: the wavefront loop ran to the end) */
    uint32_t _pad2;
} RtStats;

typedef struct RtCtx RtCtx;      /* one per (process, device, stream); not re-entrant (one render at a time per context; distinct
                                    contexts may be used from distinct host threads) */
typedef struct RtScene RtScene;  /* device-resident compiled scene, owned by the library */

/* Create a context on `device_id`. `stream` is a hipStream_t to launch on (so a caller such as
   PyTorch can order the work with its own), or NULL for a stream owned by the library. */
int rt_ctx_create(int device_id, void* stream, RtCtx** out_ctx);
int rt_ctx_destroy(RtCtx* ctx);

/* Compile the graph and copy it to the device. The caller keeps ownership of every pointer in
   `desc`; nothing in it is retained after the call returns. */
int rt_scene_upload(RtCtx* ctx, const RtSceneDesc* desc, RtScene** out_scene);
int rt_scene_destroy(RtCtx* ctx, RtScene* scene);

/* ---- how a scene is laid out on the device: per upload, in the ABI (a host with threads cannot set process environment per scene) ----
 * None of these changes a picture: they trade box tests, primitive tests and memory against each other. (Exactly: a ray that meets two
 * primitives at the SAME t — a sphere at its point of contact with the plane it rests on — is given to the one tested later, here as in
 * the reference, hittable_list.rs:40-47; layouts test in different orders. In the 1 M-sphere scene, where every sphere rests on the
 * ground rect, 2e-4 of the pixels hold such a sample; in the books' scenes none does.) The defaults are what
 * the library measures fastest; RT_LAYOUT_REFERENCE_COUNTERS is the layout whose RtStats test counts are the reference's own
 * (HittableList::hit probing every member, hittable_list.rs:39-51; BVHNode::hit visiting left then right, bvh.rs:134-143) — the
 * parity tests compare those counts with the CPU oracle's. Environment variables of the same names as in scripts/ still exist
 * as overrides for experiments; no test and no host depends on them. */
enum {
    RT_LAYOUT_LISTS_AS_REFERENCE = 1u,   /* every HittableList member in front of every ray, nothing tested at the start of a walk (default: members every ray
                                            meets anyway — a moving sphere, an all-enclosing medium, a sphere of the root BVH as large as the scene — are tested
                                            when a walk begins and left out of the tree) */
    RT_LAYOUT_LISTS_CULLED = 2u,         /* members behind culling boxes even in a scene that is only a list (default: when the scene holds a BVH of >= 32 members) */
    RT_LAYOUT_NO_MEMBER_BOXES = 4u,      /* the two members of a span-2 BVH node tested directly, as bvh.rs:99-107 does (default: a sphere gets a box of its own in LDS-sized scenes) */
    RT_LAYOUT_MEMBER_BOXES = 8u,         /* ... boxes of their own in any scene */
    RT_LAYOUT_CHILD_ORDER_AS_REFERENCE = 16u, /* one record array, left child before right (default for scenes in HBM: one array per direction octant, near child first) */
    RT_LAYOUT_SCENE_IN_HBM = 32u,        /* do not stage a small scene in LDS */
    RT_LAYOUT_NODES_32B = 64u,           /* scenes in HBM: 32-byte f32 records (with the top of the tree in LDS) instead of compressed ones */
    RT_LAYOUT_NO_SHADE_TABLES_IN_LDS = 128u,
    RT_LAYOUT_NO_EXTEND_TABLES_IN_LDS = 256u,
    RT_LAYOUT_WIDE_NODES = 512u,         /* static BVH in HBM: walk an 8-wide tree, 8 lanes to a ray, one 128-byte node per visit (measured slower than
                                            the default binary records on MI355X — VALU-bound at 8 rays per wave, DESIGN.md section 5 — kept as an option) */
    RT_LAYOUT_REFERENCE_COUNTERS = 1u | 4u | 16u
};
/* Checked, not dropped: a layout bit this library does not know, a switch set both ways (LISTS_AS_REFERENCE with LISTS_CULLED,
 * NO_MEMBER_BOXES with MEMBER_BOXES), struct_bytes < 8 or a negative list_park_cost is RT_ERR_INVALID with the reason in rt_last_error. */
typedef struct RtUploadOptions {
    uint32_t struct_bytes;     /* sizeof(RtUploadOptions) as the caller compiled it (the struct may grow at its end) */
    uint32_t layout_flags;     /* RT_LAYOUT_* */
    uint32_t lds_top_records;  /* RT_LAYOUT_NODES_32B: records of the top of the tree kept in LDS; 0 = default (1024) */
    uint32_t octant_axes;      /* near-first record arrays: 0 = the library picks the axes that matter; else 8 | mask (x = 1, y = 2, z = 4) */
    uint32_t leaf_collapse;    /* a box node whose subtree is <= n primitives of one kind becomes a leaf; 0/1 = off (default). Measured slower on every
                                  BASELINE scene (more primitive tests). The one switch that can move a sample: a ray that grazes a sphere within the
                                  rounding of its box is culled by that box in the other layouts and tested here (a handful of samples in 5e8) */
    float    list_park_cost;   /* cost of a stop at a leaf in primitive tests, for the grouping of culled list members; 0 = default (6) */
} RtUploadOptions;
int rt_scene_upload_ex(RtCtx* ctx, const RtSceneDesc* desc, const RtUploadOptions* options /* NULL = defaults */, RtScene** out_scene);

/* Number of floats rt_render writes: 3 * pixels covered by this shard's tiles. For
   shard_count <= 1 this is 3*width*height. */
int rt_output_floats(const RtParams* params, uint64_t* out_n);

/* Replaces the body of the pixel loops main.rs:731-784 (without write_color). Writes per-pixel
   RGB *sums* over the samples (what main.rs:772 accumulates), f32:
     - shard_count <= 1: rgb_sum[(y*width + x)*3 + c], row 0 = top of the image, i.e. the
       reference's j = height-1-y (main.rs:733);
     - sharded: this shard's tiles back to back, each tile_size*tile_size*3 floats row-major
       (pixels outside the image are 0); rt_untile() on the host puts gathered shards in place.
       Shards may differ by one tile; a gather uses equal buffers of shard 0's size
       (rt_output_floats with shard_index 0), which is what rt_untile expects.
   rt_render copies to a host buffer; rt_render_device leaves the result in device memory the
   caller owns (e.g. a torch tensor that RCCL then gathers). Both block until done. */
int rt_render(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* params,
              float* rgb_sum_host, RtStats* stats);
int rt_render_device(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* params,
                     void* rgb_sum_device, RtStats* stats);

/* Host-side helper: scatter `shard_count` gathered shard buffers (each rt_output_floats long,
   in shard order) into a full-frame rgb_sum. */
int rt_untile(const RtParams* params, const float* gathered, float* rgb_sum);

/* write_color (main.rs:141-169) on the device: rgb_sum (device, full frame) -> RGB8 (device). */
int rt_resolve_device(RtCtx* ctx, const void* rgb_sum_device, uint32_t width, uint32_t height,
                      uint32_t samples_per_pixel, void* rgb8_device);

/* ---- multi-GPU: the framebuffer tile-sharded over the GPUs of one node, one RCCL gather over xGMI -----------------------
 *
 * The path shards by independent pixels (main.rs:731-784 carries no cross-pixel state): the image is cut into
 * tile_size x tile_size tiles, tile t belongs to GPU t % n, the scene is replicated, every GPU renders its tiles into a compact
 * buffer, and ONE grouped ncclSend/ncclRecv exchange moves the shards to the root, whose device puts the tiles in place. The
 * picture does not depend on n (the RNG is keyed by the global pixel index). RCCL is loaded (dlopen librccl.so.1) at the first of
 * these calls; the single-GPU entry points above never touch it.
 *
 * RT_OUT_RGB8 applies write_color (main.rs:141-169) on every shard BEFORE the gather: 3 bytes per pixel cross xGMI instead of 12. */
typedef enum RtOutputKind {
    RT_OUT_RGB_SUM_F32 = 0,  /* per-pixel RGB sums, f32 (what rt_render writes) */
    RT_OUT_RGB8 = 1          /* write_color applied: RGB8, the bytes main.rs:781 stores into the image buffer */
} RtOutputKind;

/* (a) ONE PROCESS, n GPUs — what the reference's single-process host (main.rs:651-799) binds: replaces the pixel loops
 *     main.rs:730-784 by one call. n device contexts, one host thread per device inside the call, ncclCommInitAll. */
typedef struct RtMultiCtx RtMultiCtx;
typedef struct RtMultiScene RtMultiScene;
int rt_ctx_create_multi(const int* device_ids, int n_devices, RtMultiCtx** out_ctx);
int rt_ctx_destroy_multi(RtMultiCtx* ctx);
int rt_scene_upload_multi(RtMultiCtx* ctx, const RtSceneDesc* desc, RtMultiScene** out_scene);   /* compiled once, replicated on every device */
int rt_scene_upload_multi_ex(RtMultiCtx* ctx, const RtSceneDesc* desc, const RtUploadOptions* options, RtMultiScene** out_scene);
int rt_scene_destroy_multi(RtMultiCtx* ctx, RtMultiScene* scene);
/* Full frame to HOST memory: rgb_sum_host[(y*width + x)*3 + c] f32 sums, or rgb8_host[...] after write_color.
   params->shard_index / shard_count are ignored (the library shards over its devices); tile_size 0 = 32. */
int rt_render_multi(RtMultiCtx* ctx, const RtMultiScene* scene, const RtCamera* cam, const RtParams* params,
                    float* rgb_sum_host, RtStats* stats);
int rt_render_multi_rgb8(RtMultiCtx* ctx, const RtMultiScene* scene, const RtCamera* cam, const RtParams* params,
                         uint8_t* rgb8_host, RtStats* stats);
const char* rt_last_error_multi(const RtMultiCtx* ctx);

/* (b) ONE PROCESS PER GPU (torchrun-style launchers): every rank owns an RtCtx; rank 0 makes the RCCL id, the launcher's own
 *     channel carries its 128 bytes to the other ranks, every rank attaches a communicator to its context. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(uint8_t* id_out /* RT_COMM_ID_BYTES */);
int rt_comm_init_rank(RtCtx* ctx, const uint8_t* id /* RT_COMM_ID_BYTES */, int rank, int world);   /* collective over all ranks */
/* A grouped ncclSend/ncclRecv of a small buffer from this rank to itself through the context's communicator, checked byte for
   byte: proves on a one-GPU box that librccl loads, the communicator works and the exchange completes on the context's stream. */
int rt_comm_selftest(RtCtx* ctx);
/* Collective: every rank renders its shard (shard_index / shard_count of `params` are ignored, the communicator's rank / world
   are used) and sends it to rank 0; rank 0 leaves the FULL frame in `frame_device` (width*height*3 elements of f32 or u8,
   device memory the caller owns; other ranks pass NULL). Blocks until this rank's part is done. */
int rt_render_gather(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* params, uint32_t output_kind,
                     void* frame_device, RtStats* stats);
/* The root's last step on its own, for callers that gather shards themselves (e.g. with torch.distributed): `gathered_device` =
   params->shard_count shard buffers of shard 0's size back to back (f32 rgb sums or, output_kind RT_OUT_RGB8, bytes), device memory;
   `frame_device` = width*height*3 elements. One kernel on the context's stream; blocks until done. */
int rt_untile_device(RtCtx* ctx, const RtParams* params, uint32_t output_kind, const void* gathered_device, void* frame_device);
/* rt_untile for RGB8 shard buffers (host helper, same layout rules as rt_untile). */
int rt_untile_rgb8(const RtParams* params, const uint8_t* gathered, uint8_t* rgb8);

/* ---- introspection of the scene compiler: host only, never touches a GPU ---- */
typedef struct RtCompileInfo {
    uint64_t n_nodes;       /* threaded-BVH records */
    uint64_t n_box_nodes;   /* of which carry a box (= BVHNode count of the reference tree) */
    uint64_t n_spheres, n_moving, n_rects, n_tris, n_media, n_xforms, n_lights, n_materials;
    uint32_t features;      /* kernel feature bits the scene needs */
    uint32_t fits_lds;      /* nodes + sphere records fit the LDS staging budget */
    uint32_t n_first;       /* primitives tested when a walk begins instead of being met by it (none with RT_LAYOUT_LISTS_AS_REFERENCE) ... */
    uint32_t first[4];      /* ... as leaf words: kind << 28 | count << 24 | first index (kind 1 = sphere, 2 = moving sphere, 5 = medium); they are not in the node records */
    uint32_t _pad;
} RtCompileInfo;
int rt_scene_compile_info(const RtSceneDesc* desc, RtCompileInfo* out);
int rt_scene_compile_info_ex(const RtSceneDesc* desc, const RtUploadOptions* options, RtCompileInfo* out);
/* Copies the compiled node records (32 B each: f32 min[3], u32 skip, f32 max[3], u32 leaf) and the
   sphere records (f32 center[3], radius) + per-sphere meta words. Any output pointer may be NULL. */
int rt_scene_compile_dump(const RtSceneDesc* desc, void* nodes, uint64_t cap_nodes,
                          float* spheres, uint32_t* sphere_meta, uint64_t cap_spheres);
int rt_scene_compile_dump_ex(const RtSceneDesc* desc, const RtUploadOptions* options, void* nodes, uint64_t cap_nodes,
                             float* spheres, uint32_t* sphere_meta, uint64_t cap_spheres);

/* Builds the layout used when a scene does not fit LDS as a whole — the array in HBM plus an LDS copy of the top of the tree (at most
   max_top records), linked in one address space — and checks it on the host: the walk that passes every box enumerates all records
   in pre-order, every skip link lands where the plain array's does, both copies of a top record agree. *out_n_top = records in
   the top (0: no top was built, e.g. the whole scene fits). */
int rt_scene_top_layout_check(const RtSceneDesc* desc, uint32_t max_top, uint64_t* out_n_top);

/* ---- the process's ROCm runtime libraries ----
 * Writes the paths of the mapped libamdhip64 / libhsa-runtime64 / librccl objects, one per line, into `out` (NUL-terminated, cut at
 * `cap`). Returns RT_ERR_DEVICE when any of them is mapped TWICE (two copies under different paths): a process then holds two HIP
 * runtimes with separate device state and dies at exit in their static destructors ("double free or corruption") — what happens when
 * PyTorch (which bundles its own copies under other file names) is imported AFTER this library has bound the system's. Load PyTorch
 * first: its copies carry the system's sonames and are then shared. rt_ctx_create makes this check itself and refuses. */
int rt_runtime_libraries(char* out, uint64_t cap);

/* Fault injection for the failure-path tests: the next `n` renders on this context fail with RT_ERR_DEVICE before any kernel is
   launched (n = 0 disarms). Lets a one-GPU box rehearse "one rank of a collective render fails". */
int rt_test_fail_next_renders(RtCtx* ctx, uint32_t n);

/* Host only (no device is touched): the per-device host threads of a multi-GPU context — started with the context, parked between
   frames — rehearsed with `n_workers` threads over `rounds` frames of counting jobs; RT_OK when every worker ran exactly once per
   frame. (rt_render_multi itself needs n > 1 devices; this is what a one-GPU box and the CPU suite can check of it.) */
int rt_test_device_workers(int n_workers, int rounds);

/* Builds the 8-wide tree a scene in HBM is walked through (a static BVH: spheres, rects, triangles, boxes under box nodes) and checks it
   on the host: every primitive sits in exactly one leaf entry of at most eight members of one kind; every entry's box, decoded with the
   device's own float arithmetic, contains everything below it; the depth fits the walk's stack. Fills `out`; RT_ERR_UNSUPPORTED when the
   scene is not of that shape (it then keeps the binary walk). */
typedef struct RtWideInfo {
    uint64_t n_nodes, n_leaf_entries, n_inner_entries, n_prims;
    uint32_t depth, _pad;
    double mean_children;      /* used child slots per node (of 8) */
    double mean_leaf_members;  /* primitives per leaf entry (of 8) */
} RtWideInfo;
int rt_scene_wide_layout_check(const RtSceneDesc* desc, RtWideInfo* out);

const char* rt_last_error(const RtCtx* ctx);  /* ctx may be NULL: last error of this thread */
uint32_t rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
