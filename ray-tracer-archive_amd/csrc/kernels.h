// kernels.h — kernel argument blocks and launch entry points (kernels.hip <-> rt_api.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

namespace rtk {

// features a scene needs beyond "static spheres + Lambertian/Metal/Dielectric + BVH"
enum : uint32_t {
    F_MOVING = 1u, F_RECT = 2u, F_TRI = 4u, F_MEDIUM = 8u, F_XFORM = 16u, F_TEX = 32u, F_LIGHTS = 64u,
    F_ALL = 127u
};
#define RT_N_PRIM_TYPES_K 6
enum : uint32_t { CTR_NODE_TESTS = 0, CTR_PRIM_TESTS = 1, CTR_SAMPLES = 7, CTR_DEBUG = 8, CTR_SEGMENTS = 13, CTR_ITERATIONS = 14, CTR_COUNT = 16 };
#define RT_BG_SKY_GRADIENT_K 1
#define RT_NAN_PER_SAMPLE_K 0u

struct SceneDev {
    const rtd::Float4* nodes; uint32_t n_nodes;
    const rtd::Float4* top_nodes; uint32_t n_top;   // top of the tree staged in LDS when the scene does not fit (0: none)
    uint32_t n_records;      // records of `nodes` in all: n_nodes + DONE + IDLE + one park twin per leaf (device_types.h)
    uint32_t walk_start;     // address of the record a walk begins at: the root (0), or the park twin of the leaf every walk tests first (spheres as large as
    uint32_t first_leaf;     // the scene, scene_compile.cpp emit_bvh) — its leaf word, which M_C16 walks start with instead (they park by leaf word)
    uint32_t rec_unit, rec_b; // 32-byte records: address step from one record to the next, and from a record's first 16 bytes to its second: 32 and 16
                              // (an LDS-resident scene can be laid out otherwise for experiments: rt_api.cpp device_nodes)
    uint32_t nodes16;        // 1: `nodes` holds 16-byte compressed records (device_types.h Node16), corners on the grid below
    float grid_lo[3], grid_scale[3];
    uint32_t oct_mask;       // direction signs that select an array (x = 1, y = 2, z = 4)
    uint32_t oct_stride;     // M_C16: bytes between the record arrays of two direction octants (rt_api.cpp octant_order); 0 = one array
    uint32_t sort_rays;      // 1: k_shade orders the survivors of a workgroup by (octant, origin cell) before it writes them (scenes walked from HBM)
    const uint4* wide; uint32_t n_wide;   // 8-wide nodes (device_types.h), nullptr: the scene is walked through its binary records
    uint32_t n_prologue, prologue[rtd::MAX_PROLOGUE];   // moving spheres / media every ray meets: tested when a walk begins
    uint32_t n_prim_kinds;   // how many of {sphere, moving sphere, rect, triangle, medium} the scene holds
    const rtd::Float4* spheres; const uint32_t* sphere_meta; uint32_t n_spheres;
    const rtd::Float4* sphere_mat_a; const uint32_t* sphere_mat_b;   // small scenes: every sphere's material record beside it (mat_a/mat_b by sphere index), so
                                                                     // that k_shade asks for it together with the sphere instead of after its meta word; else nullptr
    const rtd::Float4* moving; const uint32_t* moving_meta;
    const rtd::Float4* rects; const uint32_t* rect_meta;
    const rtd::Float4* tris; const uint32_t* tri_meta;
    const rtd::Float4* boxes;   // 2 x Float4 per Box (device_types.h); in LDS behind the records when eb_boxes_on
    const rtd::Medium* media;
    const rtd::Xform* xforms;
    const rtd::Wrap* wraps;
    const rtd::Float4* mat_a; const uint32_t* mat_b;
    const rtd::Texture* textures;
    const rtd::PerlinTable* perlins;
    const rtd::Image* images; const uint8_t* image_bytes;
    const rtd::Light* lights; uint32_t n_lights;
    // k_shade's small tables as ONE blob (spheres, sphere meta, rects, rect meta, moving, moving meta, materials, transforms, wrapper
    // chains, lights, textures; every part 16-byte aligned), staged into LDS by each k_shade workgroup when it fits: the chain of
    // dependent look-ups hit -> meta -> wrap -> transform -> primitive -> material -> lights then runs at LDS, not L2, latency.
    const rtd::Float4* shade_blob; uint32_t shade_blob_bytes;   // 0: no staging
    // the same for k_extend's primitive pass on LDS-resident scenes: rects, moving spheres, transforms, media behind the records and the
    // sphere data (book-3 Cornell tests its 12 rects on every segment)
    const rtd::Float4* ext_blob; uint32_t ext_blob_bytes; uint32_t eb_rects, eb_moving, eb_xforms, eb_media, eb_boxes;
    uint32_t eb_rect_stride;   // 32: the rect table as it is (sc.rects is redirected too); 24: without the two padding words, where only that fits;
                               // 0: the rect table is not staged (a scene whose rects are mostly box sides: the boxes are staged, a lone rect reads HBM)
    uint32_t sb_perlin_only;   // 1: the blob holds the Perlin tables alone (a scene whose other tables are too big to stage: book-2 final)
    uint32_t sb_spheres, sb_sphere_meta, sb_rects, sb_rect_meta, sb_moving, sb_moving_meta, sb_mat_a, sb_mat_b, sb_xforms, sb_wraps, sb_lights, sb_textures;   // byte offsets
};

// Path pool: structure of arrays, one lane-contiguous record per array and slot.
//   ray_o = (origin.xyz, time)   ray_d = (direction.xyz, primitive the ray starts on)   hit = (t, primitive id), written by k_extend
//   s0 = (T.rgb, work item)      sd = draws so far << 8 | depth                          s1 = (acc.rgb, sample)  multi-sample items only
// 52 bytes per path (+ 8 for the hit): T = throughput of the sample in flight, acc = sum over the finished samples of the current work
// item. The radiance of the sample in flight needs no slot (kernels.hip PathState); the pixel is decoded from the work item, and so is
// the RNG state: a path's stream is a pure function of (seed, pixel, sample), so its base is recomputed from the work item and only
// the NUMBER of draws made so far travels (round 2 carried the 64-bit counter itself: 60 bytes).
// The pool is kQueues independent queues of queue_cap slots each (slot s of queue q = record q * queue_cap + s): paths stay in
// their queue for life, every counter (pool size, queue head, next work item) exists once per queue, kQStride u32 = 128 bytes apart.
// One counter pair for the whole pool made k_shade wait on same-address atomics (one per 512 paths, ~11 ns each: 29 ms of 39).
#ifndef RT_QUEUES
#define RT_QUEUES 8       // tuning builds: 16, 32, 64 (more counters for k_shade's atomics, fewer waves per queue in k_extend: 32 cost k_extend 4 ms of 52)
#endif
constexpr uint32_t kQueues = RT_QUEUES, kQStride = 32;
constexpr uint32_t kQShift = kQueues == 8 ? 3u : kQueues == 16 ? 4u : kQueues == 32 ? 5u : kQueues == 64 ? 6u : 0u;
static_assert(kQShift != 0u && (1u << kQShift) == kQueues, "kQueues: 8, 16, 32 or 64");
struct PoolDev {
    rtd::Float4* ray_o; rtd::Float4* ray_d; uint2* hit;
    rtd::Float4* s0; uint32_t* sd; rtd::Float4* s1;
};

// Exact u32 division by a launch-invariant divisor: q = (t + ((n - t) >> s1)) >> s2 with t = umulhi(m, n)
// (Granlund-Montgomery round-up; five instructions instead of the ~35 of a hardware-less u32 divide).
struct FastDiv { uint32_t m, s1, s2, d; };
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{0u, 0u, 0u, d};
    if (d <= 1u) return f;                       // n / 1: t = 0, q = n
    uint32_t l = 0; while ((1ull << l) < d) ++l; // ceil(log2 d), 1..32
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1ull);
    f.s1 = 1u; f.s2 = l - 1u;
    return f;
}

struct RenderDev {
    // camera.rs:6-18 in f32
    float cam_origin[3], cam_llc[3], cam_horizontal[3], cam_vertical[3], cam_u[3], cam_v[3];
    float cam_lens_radius, cam_time0, cam_time1;
    uint32_t width, height, spp, max_depth;
    uint64_t seed;
    uint32_t nan_policy;
    int32_t bg_mode; float bg[3];
    // framebuffer tiling / work decomposition
    uint32_t tile_size, tiles_x, tiles_y, shard_index, shard_count;
    uint32_t block_shift;   // a work item covers 1 << block_shift consecutive samples of one pixel
    uint32_t n_blocks;      // work items per pixel = ceil(spp >> block_shift)
    uint32_t total_items;   // in-image pixels of this shard * n_blocks
    uint32_t n_init;        // work items 0 .. n_init-1 are the pool's first fill
    uint32_t lineage;       // the path that finishes work item w goes on with item w + lineage (< total_items): every slot of the first fill owns the
                            // items of its residue class, so a finished path finds its next item without asking anybody (no atomic, no barrier)
    uint32_t queue_cap;     // slots per queue
    uint32_t q_lo, q_n, q_shift;   // this launch serves queues q_lo .. q_lo + q_n - 1 (q_n = 1 << q_shift: all 8, or one half of the pool when two
                                   // halves are in flight on two streams, rt_api.cpp render_impl)
    uint32_t n_local_tiles;
    const uint32_t* tile_prefix;  // [n_local_tiles + 1]: in-image pixels in local tiles before lt
    uint32_t tile_slack;          // the tile of pixel q is within [q / ts^2, q / ts^2 + tile_slack] (edge tiles are clipped)
    FastDiv div_ts2, div_tiles_x, div_sq_row, div_item_tile;   // ts^2, tiles_x, ts / 8, ts^2 * n_blocks
    FastDiv div_nblocks, div_width, div_ts, div_shards;        // n_blocks, width, tile_size, shard_count: a path's item id <-> pixel <-> its slot in blocksum
    // The sphere every walk tests first (SceneDev::walk_start), tested where the ray is MADE — k_generate, k_shade: full waves — instead of
    // in the walk's first primitive pass (a third of a wave's lanes): the ray record's time slot then carries that hit's t (or inf) to k_extend.
    // Only in scenes without motion (the time is used by nothing), with ONE such sphere, and not while counting.
    uint32_t first_in_shade, first_id; float first_prim[8];      // centre + radius, or a rect's two records (a0 a1 b0 b1 | k axis)
    uint32_t tiles_all_full;      // every tile of this shard lies inside the image (4096 x 4096 in 32 x 32 tiles, any shard count): tile = item / (ts^2 * n_blocks), no table
    uint32_t row_items;           // unsharded renders: work items of one full-height row of tiles (tile_size * width * n_blocks), 0 = not used. Every
    FastDiv div_row_items;        // such row holds the same number, clipped edge tile or not, so item -> tile is arithmetic (no search in tile_prefix)
    rtd::Float4* blocksum;  // [total_items]: RGB sum of one work item's samples
};

struct LaunchCfg {
    uint32_t n_cu;            // k_extend runs as a persistent grid: n_cu x (resident workgroups per CU)
    uint32_t* extend_geometry;   // optional out: resident workgroups per CU for 256- and 512-thread groups
    uint32_t features;        // F_* the scene needs
    bool scene_in_lds;
    uint32_t max_rays;        // upper bound of the rays in the queue of this launch of k_extend (sizes its grid when the queue is short)
    uint32_t extend_share;    // 1: k_extend takes every resident slot of a CU; 2: half of them (the other half of the pool is being shaded meanwhile)
};

// what a launcher has to say about the error it has just returned (nullptr: nothing beyond hipGetErrorString)
const char* launch_note();
hipError_t launch_generate(const PoolDev& pool, const RenderDev& rd, uint32_t n_init, uint32_t* out_count, hipStream_t stream);
// One wavefront iteration = launch_extend then launch_shade. No memsets in between: k_extend zeroes
// the count the following k_shade appends to, k_shade zeroes the queue head of the next k_extend.
hipError_t launch_extend(const LaunchCfg& cfg, const SceneDev& sc, const PoolDev& pool, const RenderDev& rd, const uint32_t* count_ptr,
                         uint32_t* head, uint32_t* count_out_to_zero, unsigned long long* counters, bool count, hipStream_t stream);
// The rest of a render in one launch: every path of `pool` (at most max_count) is carried to its end by one lane (kernels.hip DRAIN).
hipError_t launch_drain(const LaunchCfg& cfg, const SceneDev& sc, const PoolDev& pool, const RenderDev& rd, uint32_t max_count, const uint32_t* count_ptr,
                        uint32_t* head, uint32_t* count_out_to_zero, unsigned long long* counters, bool count, hipStream_t stream);
hipError_t launch_shade(const LaunchCfg& cfg, const SceneDev& sc, const PoolDev& in, const PoolDev& out, const RenderDev& rd, uint32_t max_count,
                        const uint32_t* count_in, uint32_t* count_out, uint32_t* head_to_zero, unsigned long long* counters,
                        bool count, hipStream_t stream);
// whether the kernels compiled for these features can test the first sphere where a ray is made (RenderDev::first_in_shade)
bool can_test_first_in_shade(uint32_t features);
hipError_t launch_resolve(const RenderDev& rd, float* out, uint32_t n_valid_pixels, hipStream_t stream);
hipError_t launch_write_color(const float* rgb_sum, uint32_t n_pixels, uint32_t spp, uint8_t* rgb8, hipStream_t stream);
// multi-GPU root: gathered shard buffers -> full frame (rt_multi.cpp)
hipError_t launch_untile_f32(const float* gathered, float* frame, uint32_t width, uint32_t height, uint32_t ts, uint32_t tiles_x, uint32_t world, uint64_t per_shard,
                             hipStream_t stream);
hipError_t launch_untile_u8(const uint8_t* gathered, uint8_t* frame, uint32_t width, uint32_t height, uint32_t ts, uint32_t tiles_x, uint32_t world, uint64_t per_shard,
                            hipStream_t stream);

}  // namespace rtk
