// rt_internal.hpp — what rt_api.cpp (single device) and rt_multi.cpp (multi-GPU + RCCL) share. Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "kernels.h"

namespace rti {

extern thread_local std::string g_last_error;

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        n = std::max<size_t>(n, 256);
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace rti

struct RtCtx {
    int device = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    hipStream_t stream2 = nullptr;               // the second lane of a render whose pool halves overlap (rt_api.cpp render_impl); made on demand
    int n_cu = 256;
    std::string err;
    // grow-only work buffers
    rti::DevBuf pool[2][6]; rti::DevBuf blocksum; rti::DevBuf counters; rti::DevBuf out_tmp; rti::DevBuf tile_prefix;
    uint32_t* h_count = nullptr;                 // pinned
    unsigned long long* h_counters = nullptr;    // pinned
    std::vector<hipEvent_t> events;
    // one-process-per-GPU gather (rt_comm_init_rank): an RCCL communicator, opaque here (rt_multi.cpp)
    void* comm = nullptr; int comm_rank = 0, comm_world = 1;
    rti::DevBuf shard_tmp;                       // this rank's shard before it is sent
    rti::DevBuf comm_words;                      // status words of the agreement that precedes every exchange (rt_multi.cpp agree_on_status)
    uint32_t* h_words = nullptr;                 // pinned: the same words on the host
    hipEvent_t ev_gather[2] = {nullptr, nullptr}; // brackets the exchange (RtStats.gather_ms); made at the first gather, kept
    uint32_t fail_renders = 0;                   // rt_test_fail_next_renders: renders still to fail (fault injection for the failure-path tests)
};

struct RtScene {
    rti::DevBuf nodes, spheres, sphere_meta, moving, moving_meta, rects, rect_meta, tris, tri_meta, boxes, media, xforms, wraps, mat_a, mat_b, textures, perlins, images,
        image_bytes, lights, top_nodes, shade_blob, ext_blob, wide, sphere_mat_a, sphere_mat_b;
    rtk::SceneDev dev{};
    uint32_t features = 0; bool in_lds = false;
    uint32_t first_id = 0; float first_prim[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the ONE sphere or rect every walk tests first (hit id; centre + radius, or the rect's two records), first_id = 0: none or several
    int bg_mode = 0; float bg[3] = {0, 0, 0};
    uint64_t n_nodes = 0, n_prims = 0, bytes = 0, lds_bytes = 0;
};

namespace rti {

inline int set_err(RtCtx* ctx, int code, const std::string& msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return code;
}
#define HIP_TRY(ctx, call)                                                                                        \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) return rti::set_err(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_DEVICE,     \
                                                  std::string(#call) + ": " + hipGetErrorString(e_));             \
    } while (0)

struct Tiling { uint32_t ts, tiles_x, tiles_y, n_tiles, n_local; };
inline int make_tiling(const RtParams& p, Tiling& t) {
    t.ts = p.tile_size ? p.tile_size : 32u;
    if (t.ts < 8u || t.ts > 256u || (t.ts & 7u)) return RT_ERR_INVALID;
    t.tiles_x = (p.width + t.ts - 1) / t.ts; t.tiles_y = (p.height + t.ts - 1) / t.ts;
    t.n_tiles = t.tiles_x * t.tiles_y;
    const uint32_t sc = p.shard_count <= 1u ? 1u : p.shard_count, si = p.shard_count <= 1u ? 0u : p.shard_index;
    if (si >= sc) return RT_ERR_INVALID;
    t.n_local = t.n_tiles > si ? (t.n_tiles - si + sc - 1) / sc : 0u;
    return RT_OK;
}

int validate_params(RtCtx* ctx, const RtParams* p);
// A scene compiled and laid out for the device, still on the host (rt_api.cpp): built once, uploaded to one device or to n
struct UploadOpts;
struct SceneImage;
int scene_image_build(const RtSceneDesc* desc, const RtUploadOptions* options, SceneImage** out, std::string& err);
int scene_image_upload(RtCtx* ctx, const SceneImage& im, RtScene** out_scene);
void scene_image_free(SceneImage* im);
// rt_render_device without the argument checks; drains the stream on failure
int render_checked(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm, void* d_out, RtStats* stats);
void comm_release(RtCtx* ctx);   // rt_multi.cpp: destroys ctx->comm if any
// mapped copies of the ROCm runtime libraries (rt_multi.cpp): false + message when one of them is mapped twice
bool runtime_libraries_ok(std::string& listing, std::string& why);

}  // namespace rti
