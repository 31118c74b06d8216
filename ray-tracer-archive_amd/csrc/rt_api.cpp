// rt_api.cpp — the C ABI of include/rt_hip.h: context, scene upload, the wavefront render loop.
// There is no CPU fallback in this library: without a HIP device every entry point that would
// compute returns RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <limits>
#include <memory>
#include <functional>
#include <vector>

#include "rt_internal.hpp"
#include "scene_compile.hpp"
#include "wide_bvh.hpp"

using namespace rti;

namespace rti { thread_local std::string g_last_error; }

namespace {

constexpr uint64_t kMaxItems = (1ull << 32) - (1ull << 28);   // work items of one render (u32 index, with head room for the allocator's overshoot)
constexpr size_t kLdsSceneBudget = 144 * 1024;  // records + sphere data staged per workgroup (one 1024-thread group per CU then; 160 KB of LDS)

template <class T> int upload(RtCtx* ctx, DevBuf& b, const std::vector<T>& v) {
    HIP_TRY(ctx, b.ensure(v.size() * sizeof(T)));
    if (!v.empty()) HIP_TRY(ctx, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

}  // namespace

static uint32_t scene_features(const rtc::CompiledScene& cs) {
    uint32_t f = 0;
    if (!cs.moving_meta.empty()) f |= rtk::F_MOVING;
    if (!cs.rect_meta.empty()) f |= rtk::F_RECT;
    if (!cs.tri_meta.empty()) f |= rtk::F_TRI;
    if (!cs.media.empty()) f |= rtk::F_MEDIUM;
    if (cs.xforms.size() > 1 || cs.wraps.size() > 1) f |= rtk::F_XFORM;
    for (uint32_t mb : cs.mat_b) { const uint32_t kind = mb & 15u; if ((mb >> 4) != rtd::TEX_INLINE && kind != rtd::MK_METAL && kind != rtd::MK_DIELECTRIC) f |= rtk::F_TEX; }
    if (cs.has_lights) f |= rtk::F_LIGHTS;
    return f;
}

int rti::validate_params(RtCtx* ctx, const RtParams* p) {
    if (!p) return set_err(ctx, RT_ERR_INVALID, "params is null");
    if (p->width < 2 || p->height < 2) return set_err(ctx, RT_ERR_INVALID, "width and height must be >= 2 (u = (i+rnd)/(W-1), main.rs:752)");
    if (p->width > 65536 || p->height > 65536) return set_err(ctx, RT_ERR_INVALID, "image too large");
    if ((uint64_t)p->width * p->height * 3ull > 0xFFFFFFFFull) return set_err(ctx, RT_ERR_INVALID, "image too large (width * height * 3 must fit 32 bits; shard the render)");
    if (p->samples_per_pixel == 0) return set_err(ctx, RT_ERR_INVALID, "samples_per_pixel must be >= 1");
    if (p->max_depth == 0 || p->max_depth > 255) return set_err(ctx, RT_ERR_INVALID, "max_depth must be in 1..255");
    if (p->nan_policy > 1) return set_err(ctx, RT_ERR_INVALID, "bad nan_policy");
    Tiling t;
    if (make_tiling(*p, t) != RT_OK) return set_err(ctx, RT_ERR_INVALID, "bad tile_size / shard_index / shard_count");
    return RT_OK;
}

static int render_impl(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm, void* d_out, RtStats* stats);

// A render that fails half way (a HIP error, out of memory) must not leave work or recorded events in flight on the
// caller's stream: drain it before the error goes back.
int rti::render_checked(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm, void* d_out, RtStats* stats) {
    if (ctx->fail_renders != 0u) { --ctx->fail_renders; return set_err(ctx, RT_ERR_DEVICE, "injected failure (rt_test_fail_next_renders)"); }
    const int r = render_impl(ctx, scene, cam, prm, d_out, stats);
    if (r != RT_OK) { const std::string keep = ctx->err; (void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError(); ctx->err = keep; g_last_error = keep; }
    return r;
}

template <class T>
static int untile_host(const RtParams* p, const T* gathered, T* frame) {
    if (!p || !gathered || !frame) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    Tiling t; RtParams q = *p; q.shard_count = 1; q.shard_index = 0;
    if (make_tiling(q, t) != RT_OK) return set_err(nullptr, RT_ERR_INVALID, "bad tiling parameters");
    const uint32_t sc = p->shard_count <= 1u ? 1u : p->shard_count;
    const uint64_t ts2 = (uint64_t)t.ts * t.ts;
    // every shard buffer has the size of the largest shard (shard 0)
    const uint64_t per_shard = (uint64_t)((t.n_tiles + sc - 1) / sc) * ts2 * 3u;
    for (uint32_t tile = 0; tile < t.n_tiles; ++tile) {
        const uint32_t s = tile % sc, lt = tile / sc, tx = tile % t.tiles_x, ty = tile / t.tiles_x;
        const T* src = gathered + (uint64_t)s * per_shard + (uint64_t)lt * ts2 * 3u;
        for (uint32_t py = 0; py < t.ts; ++py) {
            const uint32_t y = ty * t.ts + py; if (y >= p->height) break;
            for (uint32_t px = 0; px < t.ts; ++px) {
                const uint32_t x = tx * t.ts + px; if (x >= p->width) break;
                const T* a = src + ((uint64_t)py * t.ts + px) * 3u;
                T* b = frame + ((uint64_t)y * p->width + x) * 3u;
                b[0] = a[0]; b[1] = a[1]; b[2] = a[2];
            }
        }
    }
    return RT_OK;
}
static int ctx_init(RtCtx* ctx, void* stream) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
    else { HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_count, 8 * rtk::kQueues * sizeof(uint32_t), hipHostMallocDefault));   // ring of 8 x (one pool size per queue)
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_counters, sizeof(unsigned long long) * 16, hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_words, 128 * sizeof(uint32_t), hipHostMallocDefault));
    return RT_OK;
}

// (every rt_* entry point below is declared extern "C" by include/rt_hip.h; the definitions take their linkage from there)
uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(const RtCtx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int rt_ctx_create(int device_id, void* stream, RtCtx** out_ctx) {
    if (!out_ctx) return set_err(nullptr, RT_ERR_INVALID, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return set_err(nullptr, RT_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return set_err(nullptr, RT_ERR_INVALID, "device_id out of range");
    {   // two HIP runtimes in one process end in heap corruption at exit ("the abort of round 2"): refuse early, with the reason
        std::string listing, why;
        if (!runtime_libraries_ok(listing, why)) return set_err(nullptr, RT_ERR_DEVICE, why);
    }
    RtCtx* ctx = new RtCtx();
    ctx->device = device_id;
    const int rc = ctx_init(ctx, stream);
    if (rc != RT_OK) { g_last_error = ctx->err; rt_ctx_destroy(ctx); return rc; }   // nothing of a half-built context survives
    *out_ctx = ctx;
    return RT_OK;
}

int rt_ctx_destroy(RtCtx* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& pl : ctx->pool) for (auto& b : pl) b.release();
    comm_release(ctx);
    ctx->blocksum.release(); ctx->counters.release(); ctx->out_tmp.release(); ctx->tile_prefix.release(); ctx->shard_tmp.release();
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->ev_gather) if (ev) (void)hipEventDestroy(ev);
    if (ctx->h_count) (void)hipHostFree(ctx->h_count);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->h_words) (void)hipHostFree(ctx->h_words);
    ctx->comm_words.release();
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

// ---- device layout of the threaded BVH (device_types.h: NodeDev and the text above it) -----------------------------------------
// Every record carries BOTH of its links, as byte addresses: `hit` (where a walk goes when the box is passed) and `skip` (when it
// is not). A lane of k_extend is then nothing but its address: one select per visit, no "am I walking" test. States that used to
// be lane flags are records of their own that lead back to themselves: DONE (the walk ran off the end), IDLE (the lane holds no
// ray), and one PARK TWIN per record with a leaf payload — the leaf's `hit` link. A lane that passes a leaf's box lands on the
// twin and stays (its box cannot be passed: negative half extents) until the primitive pass reads the payload and the resume
// address out of the twin and moves the lane on.
//
//   address space:  [0, top_bytes)            LDS copy of the top of the tree (scenes that do not fit LDS as a whole; else empty)
//                   top_bytes + 32 * i         record i of the array (pre-order), i < n
//                   special = top_bytes + 32n  DONE, then IDLE, then the twins in the order of their leaves
//
// The top = every record above a depth cut (depth = number of enclosing subtrees [i, skip_i) of the threaded array), the deepest cut
// with at most `max_top` records: closed under "parent of", so a walk starts in it and re-enters it whenever a skip leads back up.
struct DevNodes {
    std::vector<unsigned char> main, top;   // main: records + DONE + IDLE + twins; top: the LDS copies (M_TOP)
    uint32_t n = 0, n_top = 0, n_twins = 0;
    uint32_t walk_start = 0;                // address of the record a walk begins at: the root, or the park twin of the leaf tested first
    uint32_t unit = 32u, b_off = 16u;       // address step from a record to the next, and from a record's first half to its second (kernels.h SceneDev::rec_unit)
    uint32_t records() const { return n + 2u + n_twins; }
};
static void node_boxes(const std::vector<rtd::Node>& nodes, std::vector<rtd::NodeDev>& out) {
    // Boxes as centre c and half extent h with [c-h, c+h] containing the host box, plus the part of the device's rounding that grows with
    // the RECORD's own coordinates. The slab test (kernels.hip) computes per axis tc = fma(c, 1/d, -(o/d)), th = h*|1/d| (+ e), tc -+ th:
    // five roundings and a 1-ulp reciprocal, in all at most eps*|1/d|*(5|o| + 4(|c| + h)) with eps = 2^-24. The (|c| + h) part is paid here,
    // per record: h grows by 4 eps (|c| + h) — 1e-4 units for a box 500 units from the origin; the |o| part is the ray's (set_slab_ray: e).
    // Together they stay far below t_min * |d| (0.001), so a ray leaving a box face does not pass that box again, as in the reference.
    const float inf = std::numeric_limits<float>::infinity();
    constexpr double kEps = 5.9604644775390625e-8;
    out.resize(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i) {
        const rtd::Node& n = nodes[i];
        float c[3], h[3];
        for (int a = 0; a < 3; ++a) {
            if (!std::isfinite(n.mn[a]) || !std::isfinite(n.mx[a])) { c[a] = 0.f; h[a] = inf; continue; }   // a record without a box is always passed
            c[a] = (float)(0.5 * ((double)n.mn[a] + (double)n.mx[a]));
            double hd = std::max((double)n.mx[a] - (double)c[a], (double)c[a] - (double)n.mn[a]);
            hd += 4.5 * kEps * (std::fabs((double)c[a]) + hd);
            float hf = (float)hd; if ((double)hf < hd) hf = std::nextafterf(hf, inf);
            h[a] = hf;
        }
        out[i] = rtd::NodeDev{c[0], c[1], h[0], h[1], c[2], h[2], 0u, 0u};
    }
}
// `layout` (LDS-resident scenes only, max_top = 0; scripts/ only, RT_LDS_RECORDS): 0 = 32-byte records one after the other; 1 = the first
// halves of all records, then the second halves (a record's address counts in 16-byte steps); 2 = 32-byte records 48 bytes apart. In 0 every
// first half sits on an even 16-byte column of the LDS's 64 banks and every second half on an odd one: the sixteen lanes of a ds_read_b128
// group meet in 8 columns, not 16. Measured: the bank-conflict cycles drop, the kernel does not move.
// `first_leaf`: leaf word of primitives every walk tests before it enters the tree (0: none) — a park twin of its own that resumes at the root.
static bool device_nodes(const std::vector<rtd::Node>& nodes, uint32_t max_top, DevNodes& out, int layout = 0, uint32_t first_leaf = 0u) {
    const size_t n = nodes.size();
    out = DevNodes();
    out.n = (uint32_t)n;
    if (max_top != 0u) layout = 0;
    const uint32_t unit = layout == 1 ? 16u : layout == 2 ? 48u : 32u;
    // ---- the top (optional) ----
    std::vector<uint32_t> slot(n, 0xFFFFFFFFu);
    if (max_top != 0u && n != 0) {
        std::vector<uint32_t> depth(n), per_depth, ends;
        for (size_t i = 0; i < n; ++i) {
            while (!ends.empty() && ends.back() <= i) ends.pop_back();
            depth[i] = (uint32_t)ends.size();
            if (depth[i] >= per_depth.size()) per_depth.resize(depth[i] + 1, 0u);
            per_depth[depth[i]]++;
            if (nodes[i].skip > i + 1) ends.push_back(nodes[i].skip);
        }
        uint32_t cut = 0; uint64_t total = 0;
        while (cut < per_depth.size() && total + per_depth[cut] <= max_top) total += per_depth[cut++];
        if (cut != 0 && total < n) { uint32_t k = 0; for (size_t i = 0; i < n; ++i) if (depth[i] < cut) slot[i] = k++; out.n_top = k; }
    }
    for (const rtd::Node& nd : nodes) if (nd.leaf != 0u) out.n_twins++;
    if (first_leaf != 0u) out.n_twins++;
    const uint64_t top_bytes = (uint64_t)out.n_top * 32u, total_bytes = top_bytes + (uint64_t)out.records() * std::max(unit, 32u);
    if (total_bytes >= 0xFFFFFFF0ull) return false;                      // 32-bit byte addresses
    const uint32_t special = (uint32_t)top_bytes + (uint32_t)n * unit, done = special, idle = special + unit, twin0 = special + 2u * unit;
    auto U = [&](size_t i) -> uint32_t { return i >= n ? done : (slot[i] != 0xFFFFFFFFu ? slot[i] * 32u : (uint32_t)top_bytes + (uint32_t)i * unit); };
    std::vector<rtd::NodeDev> box;
    node_boxes(nodes, box);
    std::vector<rtd::NodeDev> recs((size_t)out.records());               // record k of the array, whatever its address
    out.top.assign((size_t)top_bytes, 0);
    auto put = [&](uint32_t address_in_main, const rtd::NodeDev& d) { recs[address_in_main / unit] = d; };
    auto self_loop = [&](uint32_t addr, uint32_t resume, uint32_t payload) {
        rtd::NodeDev d{0.f, 0.f, -1.f, -1.f, 0.f, -1.f, addr, addr};     // h < 0: lo > hi on every axis, the box is never passed
        std::memcpy(&d.cx, &resume, 4); std::memcpy(&d.cy, &payload, 4);
        return d;
    };
    uint32_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        rtd::NodeDev d = box[i];
        d.skip_bytes = U(nodes[i].skip);
        if (nodes[i].leaf != 0u) {
            const uint32_t twin = twin0 + unit * k++;
            d.leaf = twin;                                                // hit link -> its park twin
            put(twin - (uint32_t)top_bytes, self_loop(twin, U(nodes[i].skip), nodes[i].leaf));   // resume = first record after the leaf
        } else d.leaf = U(i + 1);                                         // hit link -> the next record in pre-order
        put((uint32_t)i * unit, d);
        if (slot[i] != 0xFFFFFFFFu) std::memcpy(out.top.data() + (size_t)slot[i] * 32, &d, 32);
    }
    if (first_leaf != 0u) {
        const uint32_t twin = twin0 + unit * k++;
        put(twin - (uint32_t)top_bytes, self_loop(twin, U(0), first_leaf));
        out.walk_start = twin;
    } else out.walk_start = U(0);
    put((uint32_t)n * unit, self_loop(done, done, rtd::LEAF_DONE));
    put((uint32_t)n * unit + unit, self_loop(idle, idle, rtd::LEAF_IDLE));
    // the bytes as they lie in memory (and, staged by a linear copy, in LDS)
    const size_t nr = recs.size();
    out.unit = unit; out.b_off = layout == 1 ? (uint32_t)nr * 16u : 16u;
    out.main.assign(nr * std::max(unit, 32u), 0);
    for (size_t r = 0; r < nr; ++r) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(&recs[r]);
        std::memcpy(out.main.data() + r * unit, src, 16);
        std::memcpy(out.main.data() + r * unit + out.b_off, src + 16, 16);
    }
    return true;
}
// Compressed layout (device_types.h Node16) for scenes that do not fit LDS. The grid spans the union of the finite boxes; a corner
// is rounded outwards to the grid and moved one step further out, which covers the decode's rounding (t = q * (scale/d) + (lo - o)/d
// carries ~1e-7 of the scene's extent in space, a grid step is 1.5e-5 of it). Returns false when the scene has no finite box.
// ---- near-first record orders for the HBM walk ---------------------------------------------------------------------------------------
// A threaded walk visits the children of a node in ONE fixed order. For a ray that runs against that order the far child comes first, its
// hit does not shrink t_max for the near child, and the walk visits most of both subtrees. With 288 GB of HBM the remedy is space: the
// record array once per direction OCTANT, each with the two children of every binary box node ordered near-first for that octant (the axis on
// which the children's centres differ most decides). A walk starts in the array of its ray's octant; the arrays share the primitive
// tables. Which child is visited first changes no closest hit (BVHNode::hit, bvh.rs:134-143, keeps the nearer of both whatever the order).
static void octant_order(const std::vector<rtd::Node>& in, uint32_t oct, std::vector<rtd::Node>& out) {
    const size_t n = in.size();
    out.clear(); out.reserve(n);
    auto boxed = [&](uint32_t i) { return std::isfinite(in[i].mn[0]) && std::isfinite(in[i].mx[0]) && std::isfinite(in[i].mn[1]) && std::isfinite(in[i].mx[1]) && std::isfinite(in[i].mn[2]) && std::isfinite(in[i].mx[2]); };
    auto sub_end = [&](uint32_t i) { return (uint32_t)std::min<size_t>(std::max<size_t>(in[i].skip, (size_t)i + 1), n); };
    std::vector<uint32_t> stack, kids;
    for (uint32_t c = 0; c < n; c = sub_end(c)) kids.push_back(c);
    for (size_t k = kids.size(); k-- > 0;) stack.push_back(kids[k]);
    while (!stack.empty()) {
        const uint32_t i = stack.back(); stack.pop_back();
        const uint32_t pos = (uint32_t)out.size(), end = sub_end(i);
        out.push_back(in[i]);
        out[pos].skip = pos + (end - i);
        if (in[i].leaf != 0u || end == i + 1) continue;
        kids.clear();
        for (uint32_t c = i + 1; c < end; c = sub_end(c)) kids.push_back(c);
        if (kids.size() == 2 && boxed(kids[0]) && boxed(kids[1])) {
            int axis = 0; double best = -1.0, ca[3], cb[3];
            for (int a = 0; a < 3; ++a) {
                ca[a] = 0.5 * ((double)in[kids[0]].mn[a] + in[kids[0]].mx[a]); cb[a] = 0.5 * ((double)in[kids[1]].mn[a] + in[kids[1]].mx[a]);
                if (std::fabs(ca[a] - cb[a]) > best) { best = std::fabs(ca[a] - cb[a]); axis = a; }
            }
            const bool negative = ((oct >> axis) & 1u) != 0u;                  // the ray runs towards lower coordinates on that axis
            if (ca[axis] != cb[axis] && (ca[axis] < cb[axis]) == negative) std::swap(kids[0], kids[1]);
        }
        for (size_t k = kids.size(); k-- > 0;) stack.push_back(kids[k]);
    }
}
// The axes worth an array of their own: bit a set when axis a separates the two children of at least 10 % of the binary box nodes. A
// scene spread over a plane (config 5: a million spheres on the ground; y decides 6 % of its nodes) gets the arrays of 4 quadrants,
// not 8 octants — half the cache footprint for the same visits (8 arrays 64.9 visits per segment, 1067 Msamples/s; x and z only 65.4, 1109).
static uint32_t deciding_axes(const std::vector<rtd::Node>& in) {
    const size_t n = in.size();
    auto boxed = [&](size_t i) { for (int a = 0; a < 3; ++a) if (!std::isfinite(in[i].mn[a]) || !std::isfinite(in[i].mx[a])) return false; return true; };
    auto sub_end = [&](size_t i) { return std::min<size_t>(std::max<size_t>(in[i].skip, i + 1), n); };
    uint64_t count[3] = {0, 0, 0}, total = 0;
    for (size_t i = 0; i < n; ++i) {
        const size_t end = sub_end(i);
        if (in[i].leaf != 0u || end == i + 1) continue;
        const size_t k0 = i + 1, k1 = sub_end(k0);
        if (k1 >= end || sub_end(k1) != end || !boxed(k0) || !boxed(k1)) continue;     // exactly two children, both with a box
        int axis = 0; double best = -1.0;
        for (int a = 0; a < 3; ++a) {
            const double dc = std::fabs(0.5 * ((double)in[k0].mn[a] + in[k0].mx[a]) - 0.5 * ((double)in[k1].mn[a] + in[k1].mx[a]));
            if (dc > best) { best = dc; axis = a; }
        }
        count[axis]++; total++;
    }
    uint32_t mask = 0u;
    for (int a = 0; a < 3; ++a) if (total != 0 && count[a] * 10u >= total) mask |= 1u << a;
    return mask;
}
static bool device_nodes16(const std::vector<rtd::Node>& nodes, std::vector<rtd::Node16>& out, float grid_lo[3], float grid_scale[3], uint32_t link_base = 0u,
                           bool keep_grid = false) {
    const size_t n = nodes.size();
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (const rtd::Node& nd : nodes) for (int a = 0; a < 3; ++a) {
        if (std::isfinite(nd.mn[a])) lo[a] = std::min(lo[a], (double)nd.mn[a]);
        if (std::isfinite(nd.mx[a])) hi[a] = std::max(hi[a], (double)nd.mx[a]);
    }
    for (int a = 0; a < 3 && !keep_grid; ++a) {
        if (!(hi[a] >= lo[a])) return false;
        const double pad = 1e-6 * (std::fabs(lo[a]) + std::fabs(hi[a])) + 1e-30;
        lo[a] -= pad; hi[a] += pad;
        grid_lo[a] = (float)lo[a]; if ((double)grid_lo[a] > lo[a]) grid_lo[a] = std::nextafterf(grid_lo[a], -std::numeric_limits<float>::infinity());
        double sc = (hi[a] - (double)grid_lo[a]) / 65533.0;               // corners use 1 .. 65534 before the extra step
        grid_scale[a] = (float)sc; if ((double)grid_scale[a] < sc) grid_scale[a] = std::nextafterf(grid_scale[a], std::numeric_limits<float>::infinity());
    }
    if ((uint64_t)(n + 2) * 16ull >= 0x7FFFFFF0ull) return false;          // links are 31-bit byte offsets
    out.assign(n + 2, rtd::Node16{});
    for (size_t i = 0; i < n; ++i) {
        const rtd::Node& nd = nodes[i];
        rtd::Node16 r{};
        const bool boxed = std::isfinite(nd.mn[0]) && std::isfinite(nd.mx[0]);
        if (!boxed) { r.lo[0] = 0xFFFFu; r.hi[0] = 0u; r.lo[1] = r.lo[2] = 0u; r.hi[1] = r.hi[2] = 0xFFFFu; }
        else for (int a = 0; a < 3; ++a) {
            const double ql = std::floor(((double)nd.mn[a] - (double)grid_lo[a]) / (double)grid_scale[a]) - 1.0;
            const double qh = std::ceil(((double)nd.mx[a] - (double)grid_lo[a]) / (double)grid_scale[a]) + 1.0;
            r.lo[a] = (uint16_t)std::min(65535.0, std::max(0.0, ql));
            r.hi[a] = (uint16_t)std::min(65535.0, std::max(0.0, qh));
        }
        r.link = nd.leaf != 0u ? (0x80000000u | nd.leaf) : link_base + (uint32_t)std::min<size_t>(nd.skip, n) * 16u;
        out[i] = r;
    }
    rtd::Node16 end{};                                                      // closing record: no box, "leaf" with the DONE payload (twice: a done lane reads one further)
    end.lo[0] = 0xFFFFu; end.hi[0] = 0u; end.hi[1] = end.hi[2] = 0xFFFFu; end.link = 0x80000000u | rtd::LEAF_DONE;
    out[n] = end; out[n + 1] = end;
    return true;
}
static size_t lds_scene_bytes(const rtc::CompiledScene& cs) {
    size_t twins = 0; for (const rtd::Node& nd : cs.nodes) if (nd.leaf != 0u) ++twins;
    return (cs.nodes.size() + 2 + twins) * 32 + cs.spheres.size() * 16;
}

// ---- layout options: RtUploadOptions of the ABI, resolved once per upload ---------------------------------------------------------------
// Environment variables are OVERRIDES for the experiment scripts under scripts/ only (they predate the ABI fields and keep the A/B
// scripts one-liners); no test and no host path sets them. A process that never sets them gets exactly what its RtUploadOptions say.
struct rti::UploadOpts {
    rtc::CompileOptions compile;
    bool lds_scene = true, node16 = true, octant_order = true, shade_lds = true, perlin_lds = true, extend_lds_tables = true, wide_nodes = false;
    uint32_t max_top = 1024u, octant_axes = 0u /* 0 = pick; else 8 | mask */;
};
// RtUploadOptions as the ABI promises them: unknown or contradictory switches are refused, not dropped
static int check_options(const RtUploadOptions* options, std::string& err) {
    if (options) {
        // a host built against a newer header must hear that this library does not know a switch, not have it dropped
        constexpr uint32_t kKnown = RT_LAYOUT_LISTS_AS_REFERENCE | RT_LAYOUT_LISTS_CULLED | RT_LAYOUT_NO_MEMBER_BOXES | RT_LAYOUT_MEMBER_BOXES | RT_LAYOUT_CHILD_ORDER_AS_REFERENCE |
                                    RT_LAYOUT_SCENE_IN_HBM | RT_LAYOUT_NODES_32B | RT_LAYOUT_NO_SHADE_TABLES_IN_LDS | RT_LAYOUT_NO_EXTEND_TABLES_IN_LDS | RT_LAYOUT_WIDE_NODES;
        if (options->struct_bytes < 8u || options->struct_bytes > 4096u) { err = "RtUploadOptions.struct_bytes is not set (sizeof(RtUploadOptions))"; return RT_ERR_INVALID; }
        const uint32_t f = options->layout_flags;
        if (f & ~kKnown) { err = "RtUploadOptions.layout_flags: unknown bit"; return RT_ERR_INVALID; }
        if (((f & RT_LAYOUT_LISTS_AS_REFERENCE) && (f & RT_LAYOUT_LISTS_CULLED)) || ((f & RT_LAYOUT_NO_MEMBER_BOXES) && (f & RT_LAYOUT_MEMBER_BOXES))) {
            err = "RtUploadOptions.layout_flags: a switch is set both ways"; return RT_ERR_INVALID;
        }
        if (options->struct_bytes >= 24u && !(options->list_park_cost >= 0.f)) { err = "RtUploadOptions.list_park_cost is negative or NaN"; return RT_ERR_INVALID; }
    }
    return RT_OK;
}

static rti::UploadOpts resolve_options(const RtUploadOptions* o) {
    rti::UploadOpts u;
    if (o && o->struct_bytes >= 8u) {
        const uint32_t f = o->layout_flags;
        if (f & RT_LAYOUT_LISTS_AS_REFERENCE) { u.compile.cull_lists = 0; u.compile.big_spheres_first = false; }   // "nothing tested at the start of a walk"
        else if (f & RT_LAYOUT_LISTS_CULLED) u.compile.cull_lists = 1;
        if (f & RT_LAYOUT_NO_MEMBER_BOXES) u.compile.member_boxes = 0; else if (f & RT_LAYOUT_MEMBER_BOXES) u.compile.member_boxes = 1;
        u.octant_order = !(f & RT_LAYOUT_CHILD_ORDER_AS_REFERENCE);
        u.lds_scene = !(f & RT_LAYOUT_SCENE_IN_HBM);
        u.node16 = !(f & RT_LAYOUT_NODES_32B);
        u.shade_lds = u.perlin_lds = !(f & RT_LAYOUT_NO_SHADE_TABLES_IN_LDS);
        u.extend_lds_tables = !(f & RT_LAYOUT_NO_EXTEND_TABLES_IN_LDS);
        u.wide_nodes = (f & RT_LAYOUT_WIDE_NODES) != 0u;
        if (u.wide_nodes) u.compile.big_spheres_first = false;      // the 8-wide walk tests nothing before the tree
        if (o->struct_bytes >= 12u && o->lds_top_records) u.max_top = o->lds_top_records;
        if (o->struct_bytes >= 16u && o->octant_axes) u.octant_axes = 8u | (o->octant_axes & 7u);
        if (o->struct_bytes >= 20u) u.compile.leaf_collapse = o->leaf_collapse;
        if (o->struct_bytes >= 24u && o->list_park_cost > 0.f) u.compile.park_cost = o->list_park_cost;
    }
    auto env = [](const char* n) -> const char* { const char* e = getenv(n); return e && e[0] ? e : nullptr; };
    if (const char* e = env("RT_LIST_CULL")) u.compile.cull_lists = e[0] == '2' ? 1 : (e[0] == '0' ? 0 : u.compile.cull_lists);
    if (const char* e = env("RT_BIG_SPHERES_FIRST")) u.compile.big_spheres_first = e[0] != '0';
    if (const char* e = env("RT_PAIR_BOXES")) u.compile.member_boxes = e[0] == '2' ? 1 : (e[0] == '0' ? 0 : u.compile.member_boxes);
    if (const char* e = env("RT_LIST_PARK_COST")) u.compile.park_cost = std::max(0.0, std::atof(e));
    if (const char* e = env("RT_LEAF_COLLAPSE")) u.compile.leaf_collapse = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env("RT_LDS_SCENE")) u.lds_scene = u.lds_scene && e[0] != '0';
    if (const char* e = env("RT_TOP_NODES")) u.max_top = (uint32_t)std::strtoul(e, nullptr, 10);
    if (const char* e = env("RT_NODE16")) u.node16 = u.node16 && e[0] != '0';
    if (const char* e = env("RT_OCTANT_ORDER")) u.octant_order = u.octant_order && e[0] != '0';
    if (const char* e = env("RT_OCTANT_AXES")) u.octant_axes = 8u | ((uint32_t)std::strtoul(e, nullptr, 10) & 7u);
    if (const char* e = env("RT_SHADE_LDS")) u.shade_lds = u.shade_lds && e[0] != '0';
    if (const char* e = env("RT_SHADE_PERLIN_LDS")) u.perlin_lds = e[0] != '0';
    if (const char* e = env("RT_EXTEND_LDS_TABLES")) u.extend_lds_tables = u.extend_lds_tables && e[0] != '0';
    if (const char* e = env("RT_WIDE_NODES")) { u.wide_nodes = e[0] != '0'; if (u.wide_nodes) u.compile.big_spheres_first = false; }
    u.max_top = std::min<uint32_t>(u.max_top, (128u * 1024u) / 32u);
    return u;
}

// ---- the host image of a scene: everything an upload copies to a device, built ONCE (rt_scene_upload_multi uploads it n times) ---------
struct rti::SceneImage {
    rtc::CompiledScene cs;
    DevNodes dn;
    std::vector<rtd::Node16> n16;
    float grid_lo[3] = {0, 0, 0}, grid_scale[3] = {1, 1, 1};
    bool in_lds = false, c16 = false, top = false;
    uint32_t oct_stride = 0u, oct_mask = 7u;
    std::vector<unsigned char> blob, eblob;
    uint32_t sb[12] = {0}, perlin_only = 0u, eb[5] = {0}, eb_rect_stride = 32u;
    uint32_t features = 0u;
    bool sort_rays = true;
    rtw::WideTree wide; bool use_wide = false;     // 8-wide nodes for k_extend_wide (a static BVH that does not fit LDS)
};
void rti::scene_image_free(SceneImage* im) { delete im; }

int rti::scene_image_build(const RtSceneDesc* desc, const RtUploadOptions* options, SceneImage** out, std::string& err) {
    *out = nullptr;
    { const int ok = check_options(options, err); if (ok != RT_OK) return ok; }
    const UploadOpts opt = resolve_options(options);
    std::unique_ptr<SceneImage> im(new SceneImage());
    rtc::CompiledScene& cs = im->cs;
    const int rc = rtc::compile_scene(*desc, opt.compile, cs);
    if (rc != RT_OK) { err = "scene: " + cs.error; return rc; }
    im->in_lds = opt.lds_scene && lds_scene_bytes(cs) <= kLdsSceneBudget;
    im->features = scene_features(cs);
    if (const char* e = getenv("RT_SORT_RAYS")) im->sort_rays = e[0] != '0';     // scripts/ only
    // RT_LAYOUT_WIDE_NODES: a static BVH in HBM (spheres / rects / triangles / boxes under box nodes only) walked 8 lanes to a ray over an
    // 8-wide tree (kernels.hip k_extend_wide) instead of the binary records below.
    if (!im->in_lds && opt.wide_nodes && opt.node16 && (im->features & ~(rtk::F_RECT | rtk::F_TRI)) == 0u && cs.prologue.empty() && cs.first_leaf == 0u && rtw::eligible(cs.nodes)) {   // (k_extend_wide tests nothing before the tree)
        double extent = 0.0;
        for (int a = 0; a < 3; ++a) extent = std::max(extent, std::max(std::fabs((double)cs.nodes[0].mn[a]), std::fabs((double)cs.nodes[0].mx[a])));
        std::string werr;
        im->use_wide = rtw::build(cs.nodes, (float)(6e-7 * extent), im->wide, werr) && im->wide.depth < rtd::WIDE_MAX_DEPTH;
        if (!im->use_wide) im->wide = rtw::WideTree();
    }
    // 16-byte compressed records for a scene that does not fit LDS (RT_LAYOUT_NODES_32B: 32-byte records with the top of the tree in LDS)
    im->c16 = !im->in_lds && opt.node16 && device_nodes16(cs.nodes, im->n16, im->grid_lo, im->grid_scale);
    // the record array once per direction octant (octant_order above), as long as 31-bit links reach; RT_LAYOUT_CHILD_ORDER_AS_REFERENCE: one
    // array, the reference's order (left, then right) — the order whose visit counts the tests compare with the CPU restatement's
    {
        bool octants = im->c16 && !im->use_wide && opt.octant_order && 8ull * (cs.nodes.size() + 2) * 16ull < 0x7FFFFFF0ull;   // (the wide walk orders near-first by itself)
        if (octants) {
            const uint32_t stride = (uint32_t)((cs.nodes.size() + 2) * 16);
            std::vector<rtd::Node16> all; all.reserve(8 * (cs.nodes.size() + 2));
            std::vector<rtd::Node> ordered; std::vector<rtd::Node16> one;
            uint32_t mask = (opt.octant_axes & 8u) ? (opt.octant_axes & 7u) : deciding_axes(cs.nodes);
            for (uint32_t oct = 0; oct < 8 && octants; ++oct) {
                if ((oct & ~mask) != 0u) { all.resize(all.size() + cs.nodes.size() + 2); continue; }      // never selected (kernels.hip go_root): left empty, never touched
                octant_order(cs.nodes, oct, ordered);
                octants = ordered.size() == cs.nodes.size() && device_nodes16(ordered, one, im->grid_lo, im->grid_scale, oct * stride, true);
                all.insert(all.end(), one.begin(), one.end());
            }
            if (octants) { im->n16.swap(all); im->oct_stride = stride; im->oct_mask = mask; }
        }
    }
    int lds_layout = 0;
    if (const char* e = getenv("RT_LDS_RECORDS")) lds_layout = e[0] == '1' ? 1 : e[0] == '2' ? 2 : 0;     // scripts/ only: 1 = halves apart, 2 = 48 bytes apart
    if (lds_layout == 2 && lds_scene_bytes(cs) + (lds_scene_bytes(cs) - cs.spheres.size() * 16) / 2 > 78 * 1024) lds_layout = 1;   // (two workgroups per CU or not at all)
    if (!im->c16 && !device_nodes(cs.nodes, im->in_lds ? 0u : opt.max_top, im->dn, im->in_lds ? lds_layout : 0, cs.first_leaf)) { err = "scene: node array beyond 4 GB"; return RT_ERR_UNSUPPORTED; }
    im->top = !im->c16 && im->dn.n_top != 0u;
    // k_shade's small tables as one blob for LDS staging (kernels.h SceneDev::shade_blob): only when it is small
    {
        std::vector<unsigned char>& blob = im->blob; uint32_t* sb = im->sb;
        auto put = [&](const void* p, size_t bytes) { const uint32_t at = (uint32_t)blob.size(); blob.resize((blob.size() + bytes + 15) & ~(size_t)15, 0); if (bytes) std::memcpy(blob.data() + at, p, bytes); return at; };
        sb[0] = put(cs.spheres.data(), cs.spheres.size() * 16); sb[1] = put(cs.sphere_meta.data(), cs.sphere_meta.size() * 4);
        sb[2] = put(cs.rects.data(), cs.rects.size() * 16); sb[3] = put(cs.rect_meta.data(), cs.rect_meta.size() * 4);
        sb[4] = put(cs.moving.data(), cs.moving.size() * 16); sb[5] = put(cs.moving_meta.data(), cs.moving_meta.size() * 4);
        sb[6] = put(cs.mat_a.data(), cs.mat_a.size() * 16); sb[7] = put(cs.mat_b.data(), cs.mat_b.size() * 4);
        sb[8] = put(cs.xforms.data(), cs.xforms.size() * sizeof(rtd::Xform)); sb[9] = put(cs.wraps.data(), cs.wraps.size() * sizeof(rtd::Wrap));
        sb[10] = put(cs.lights.data(), cs.lights.size() * sizeof(rtd::Light)); sb[11] = put(cs.textures.data(), cs.textures.size() * sizeof(rtd::Texture));
        // every 512-path workgroup pays for the copy: measured on Cornell (1.6 KB) k_shade 87.4 -> 75.2 ms, on book-1 (19 KB) 36.8 -> 40.2 ms
        if (!(opt.shade_lds && blob.size() <= 8 * 1024)) blob.clear();
        // a scene with noise textures whose other tables are too big to stage: the Perlin tables alone (7 KB each). A turbulence is a chain of
        // 7 x 8 x 4 dependent look-ups, and a wave waits for the one lane that makes it
        if (blob.empty() && opt.perlin_lds && !cs.perlins.empty() && cs.perlins.size() * sizeof(rtd::PerlinTable) <= 16 * 1024) {
            put(cs.perlins.data(), cs.perlins.size() * sizeof(rtd::PerlinTable)); im->perlin_only = 1u;
            // ... and the tables that do not grow with the primitive count, if they are small: material, texture, transform, wrapper, light
            const size_t small = cs.mat_a.size() * 20 + cs.xforms.size() * sizeof(rtd::Xform) + cs.wraps.size() * sizeof(rtd::Wrap) + cs.lights.size() * sizeof(rtd::Light) +
                                 cs.textures.size() * sizeof(rtd::Texture);
            if (small <= 4 * 1024) {
                sb[6] = put(cs.mat_a.data(), cs.mat_a.size() * 16); sb[7] = put(cs.mat_b.data(), cs.mat_b.size() * 4);
                sb[8] = put(cs.xforms.data(), cs.xforms.size() * sizeof(rtd::Xform)); sb[9] = put(cs.wraps.data(), cs.wraps.size() * sizeof(rtd::Wrap));
                sb[10] = put(cs.lights.data(), cs.lights.size() * sizeof(rtd::Light)); sb[11] = put(cs.textures.data(), cs.textures.size() * sizeof(rtd::Texture));
            } else im->perlin_only = 2u;
        }
    }
    // k_extend's primitive pass tables for LDS-resident scenes (kernels.h SceneDev::ext_blob): small ones only, inside the LDS budget
    if (im->in_lds && opt.extend_lds_tables && !(cs.rects.empty() && cs.moving.empty() && cs.media.empty())) {
        // staged once per workgroup and launch (persistent waves), so size only matters against the 160 KB of the CU: the box records
        // (32 B per Box) always; the rect table as it is (2 x 16 B per rect) where that fits too, else without its two padding words, else
        // not at all — in a scene whose rects are the sides of boxes (book-2 final: 2400 of 2401) the walk never reads them, and the one
        // lone rect comes from HBM/L2
        std::vector<unsigned char>& eblob = im->eblob; uint32_t* eb = im->eb;
        const size_t room = 160 * 1024 - lds_scene_bytes(cs);
        const size_t n_rects = cs.rects.size() / 2;
        for (uint32_t stride : {32u, 24u, 0u}) {
            eblob.clear();
            auto put = [&](const void* p, size_t bytes) { const uint32_t at = (uint32_t)eblob.size(); eblob.resize((eblob.size() + bytes + 15) & ~(size_t)15, 0); if (bytes) std::memcpy(eblob.data() + at, p, bytes); return at; };
            eb[4] = put(cs.boxes.data(), cs.boxes.size() * 16);
            if (stride == 0u) eb[0] = 0u;
            else if (stride == 32u) eb[0] = put(cs.rects.data(), n_rects * 32);
            else {
                std::vector<float> packed(n_rects * 6);
                for (size_t i = 0; i < n_rects; ++i) { std::memcpy(&packed[6 * i], &cs.rects[2 * i], 16); std::memcpy(&packed[6 * i + 4], &cs.rects[2 * i + 1], 8); }
                eb[0] = put(packed.data(), packed.size() * 4);
            }
            eb[1] = put(cs.moving.data(), cs.moving.size() * 16);
            eb[2] = put(cs.xforms.data(), cs.xforms.size() * sizeof(rtd::Xform)); eb[3] = put(cs.media.data(), cs.media.size() * sizeof(rtd::Medium));
            im->eb_rect_stride = stride;
            if (eblob.size() <= room) break;
            eblob.clear();
        }
    }
    *out = im.release();
    return RT_OK;
}

int rti::scene_image_upload(RtCtx* ctx, const SceneImage& im, RtScene** out_scene) {
    *out_scene = nullptr;
    const rtc::CompiledScene& cs = im.cs;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RtScene* s = new RtScene();
    int r = RT_OK;
    auto up = [&](auto& buf, const auto& vec) { if (r == RT_OK) r = upload(ctx, buf, vec); };
    if (im.top) up(s->top_nodes, im.dn.top);
    if (im.c16) up(s->nodes, im.n16); else up(s->nodes, im.dn.main);
    up(s->spheres, cs.spheres); up(s->sphere_meta, cs.sphere_meta); up(s->moving, cs.moving); up(s->moving_meta, cs.moving_meta);
    up(s->rects, cs.rects); up(s->rect_meta, cs.rect_meta); up(s->tris, cs.tris); up(s->tri_meta, cs.tri_meta); up(s->boxes, cs.boxes); up(s->media, cs.media);
    up(s->xforms, cs.xforms); up(s->wraps, cs.wraps); up(s->mat_a, cs.mat_a); up(s->mat_b, cs.mat_b); up(s->textures, cs.textures); up(s->perlins, cs.perlins);
    up(s->images, cs.images); up(s->image_bytes, cs.image_bytes); up(s->lights, cs.lights);
    // a small scene whose shading tables are not staged in LDS: the material record of every sphere by SPHERE index (one dependent load fewer in
    // k_shade: book-1 waits ~700 cycles of a wave's ~20 000 for the material's record after the sphere's)
    std::vector<rtd::Float4> sphere_ma; std::vector<uint32_t> sphere_mb;
    if (im.blob.empty() && !cs.spheres.empty() && cs.spheres.size() <= 65536 && !(getenv("RT_SPHERE_MATS") && getenv("RT_SPHERE_MATS")[0] == '0')) {
        sphere_ma.resize(cs.spheres.size()); sphere_mb.resize(cs.spheres.size());
        for (size_t k = 0; k < cs.spheres.size(); ++k) { const uint32_t m = cs.sphere_meta[k] & rtd::META_MAT_MASK; sphere_ma[k] = cs.mat_a[m]; sphere_mb[k] = cs.mat_b[m]; }
        up(s->sphere_mat_a, sphere_ma); up(s->sphere_mat_b, sphere_mb);
    }
    if (im.use_wide) up(s->wide, im.wide.words);
    if (!im.blob.empty()) up(s->shade_blob, im.blob);
    if (!im.eblob.empty()) up(s->ext_blob, im.eblob);
    if (r == RT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) r = set_err(ctx, RT_ERR_DEVICE, "scene upload failed");   // the image is the caller's: its vectors outlive this wait
    if (r != RT_OK) { rt_scene_destroy(ctx, s); return r; }
    rtk::SceneDev& d = s->dev;
    const uint32_t* sb = im.sb; const uint32_t* eb = im.eb;
    d.nodes = (const rtd::Float4*)s->nodes.p; d.n_nodes = (uint32_t)cs.nodes.size();
    d.top_nodes = im.top ? (const rtd::Float4*)s->top_nodes.p : nullptr; d.n_top = im.top ? im.dn.n_top : 0u; d.n_records = im.c16 ? (uint32_t)im.n16.size() : im.dn.records();
    d.rec_unit = im.c16 ? 16u : im.dn.unit; d.rec_b = im.c16 ? 0u : im.dn.b_off;
    d.walk_start = im.c16 ? 0u : im.dn.walk_start; d.first_leaf = cs.first_leaf;
    d.oct_stride = im.oct_stride; d.oct_mask = im.oct_mask;
    d.sort_rays = (im.c16 && im.sort_rays) ? 1u : 0u;
    d.wide = im.use_wide ? (const uint4*)s->wide.p : nullptr; d.n_wide = im.use_wide ? im.wide.n_nodes : 0u;
    d.nodes16 = im.c16 ? 1u : 0u; for (int a = 0; a < 3; ++a) { d.grid_lo[a] = im.grid_lo[a]; d.grid_scale[a] = im.grid_scale[a]; }
    d.n_prologue = (uint32_t)cs.prologue.size();
    for (uint32_t k = 0; k < rtd::MAX_PROLOGUE; ++k) d.prologue[k] = k < cs.prologue.size() ? cs.prologue[k] : 0u;
    d.n_prim_kinds = (cs.sphere_meta.empty() ? 0u : 1u) + (cs.moving_meta.empty() ? 0u : 1u) + (cs.rect_meta.empty() ? 0u : 1u) +
                     (cs.tri_meta.empty() ? 0u : 1u) + (cs.media.empty() ? 0u : 1u);   // (a Box counts with the rects: their sides)
    d.spheres = (const rtd::Float4*)s->spheres.p; d.sphere_meta = (const uint32_t*)s->sphere_meta.p; d.n_spheres = (uint32_t)cs.spheres.size();
    if (!sphere_ma.empty()) { d.sphere_mat_a = (const rtd::Float4*)s->sphere_mat_a.p; d.sphere_mat_b = (const uint32_t*)s->sphere_mat_b.p; }
    d.moving = (const rtd::Float4*)s->moving.p; d.moving_meta = (const uint32_t*)s->moving_meta.p;
    d.rects = (const rtd::Float4*)s->rects.p; d.rect_meta = (const uint32_t*)s->rect_meta.p;
    d.tris = (const rtd::Float4*)s->tris.p; d.tri_meta = (const uint32_t*)s->tri_meta.p; d.boxes = (const rtd::Float4*)s->boxes.p;
    d.media = (const rtd::Medium*)s->media.p; d.xforms = (const rtd::Xform*)s->xforms.p; d.wraps = (const rtd::Wrap*)s->wraps.p;
    d.mat_a = (const rtd::Float4*)s->mat_a.p; d.mat_b = (const uint32_t*)s->mat_b.p;
    d.textures = (const rtd::Texture*)s->textures.p; d.perlins = (const rtd::PerlinTable*)s->perlins.p;
    d.images = (const rtd::Image*)s->images.p; d.image_bytes = (const uint8_t*)s->image_bytes.p;
    d.lights = (const rtd::Light*)s->lights.p; d.n_lights = (uint32_t)cs.lights.size();
    d.shade_blob = im.blob.empty() ? nullptr : (const rtd::Float4*)s->shade_blob.p; d.shade_blob_bytes = (uint32_t)im.blob.size();
    d.sb_spheres = sb[0]; d.sb_sphere_meta = sb[1]; d.sb_rects = sb[2]; d.sb_rect_meta = sb[3]; d.sb_moving = sb[4]; d.sb_moving_meta = sb[5];
    d.ext_blob = im.eblob.empty() ? nullptr : (const rtd::Float4*)s->ext_blob.p; d.ext_blob_bytes = (uint32_t)im.eblob.size();
    d.eb_rect_stride = im.eb_rect_stride; d.eb_rects = eb[0]; d.eb_moving = eb[1]; d.eb_xforms = eb[2]; d.eb_media = eb[3]; d.eb_boxes = eb[4];
    d.sb_perlin_only = im.perlin_only;
    d.sb_mat_a = sb[6]; d.sb_mat_b = sb[7]; d.sb_xforms = sb[8]; d.sb_wraps = sb[9]; d.sb_lights = sb[10]; d.sb_textures = sb[11];
    s->features = im.features;
    s->in_lds = im.in_lds; s->lds_bytes = lds_scene_bytes(cs);
    s->bg_mode = cs.background_mode; for (int i = 0; i < 3; ++i) s->bg[i] = cs.background[i];
    if (cs.first_leaf != 0u && ((cs.first_leaf >> 24) & 15u) == 1u) {
        const uint32_t k = cs.first_leaf & rtd::LEAF_MAX_FIRST, type = cs.first_leaf >> 28;
        if (type == rtd::LT_SPHERE) {
            s->first_id = (rtd::LT_SPHERE << 28) | k;
            s->first_prim[0] = cs.spheres[k].x; s->first_prim[1] = cs.spheres[k].y; s->first_prim[2] = cs.spheres[k].z; s->first_prim[3] = cs.spheres[k].w;
        } else if (type == rtd::LT_RECT) {
            s->first_id = (rtd::LT_RECT << 28) | k;
            std::memcpy(s->first_prim, &cs.rects[2 * k], 32);
        }
    }
    s->n_nodes = cs.nodes.size();
    s->n_prims = cs.sphere_meta.size() + cs.moving_meta.size() + cs.rect_meta.size() + cs.tri_meta.size() + cs.media.size();
    s->bytes = cs.nodes.size() * 32 + cs.spheres.size() * 16 + cs.moving.size() * 16 + cs.rects.size() * 16 + cs.tris.size() * 16 +
               4 * (cs.sphere_meta.size() + cs.moving_meta.size() + cs.rect_meta.size() + cs.tri_meta.size());
    *out_scene = s;
    return RT_OK;
}

int rt_scene_upload_ex(RtCtx* ctx, const RtSceneDesc* desc, const RtUploadOptions* options, RtScene** out_scene) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!desc || !out_scene) return set_err(ctx, RT_ERR_INVALID, "desc / out_scene is null");
    *out_scene = nullptr;
    SceneImage* im = nullptr; std::string err;
    const int rc = scene_image_build(desc, options, &im, err);
    if (rc != RT_OK) return set_err(ctx, rc, err);
    const int r = scene_image_upload(ctx, *im, out_scene);
    scene_image_free(im);
    return r;
}
int rt_scene_upload(RtCtx* ctx, const RtSceneDesc* desc, RtScene** out_scene) { return rt_scene_upload_ex(ctx, desc, nullptr, out_scene); }

int rt_scene_destroy(RtCtx* ctx, RtScene* s) {
    if (!s) return RT_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    DevBuf* all[] = {&s->nodes, &s->spheres, &s->sphere_meta, &s->moving, &s->moving_meta, &s->rects, &s->rect_meta, &s->tris, &s->tri_meta, &s->boxes, &s->media,
                     &s->xforms, &s->wraps, &s->mat_a, &s->mat_b, &s->textures, &s->perlins, &s->images, &s->image_bytes, &s->lights, &s->top_nodes, &s->shade_blob, &s->ext_blob, &s->wide, &s->sphere_mat_a, &s->sphere_mat_b};
    for (DevBuf* b : all) b->release();
    delete s;
    return RT_OK;
}

int rt_output_floats(const RtParams* p, uint64_t* out_n) {
    if (!p || !out_n) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    Tiling t;
    if (make_tiling(*p, t) != RT_OK) return set_err(nullptr, RT_ERR_INVALID, "bad tiling parameters");
    if (p->shard_count <= 1u) *out_n = (uint64_t)p->width * p->height * 3u;
    else *out_n = (uint64_t)t.n_local * t.ts * t.ts * 3u;
    return RT_OK;
}

// a launcher's own explanation (kernels.h launch_note) in front of the HIP error string
#define LAUNCH_TRY(call)                                                                                                        \
    do {                                                                                                                        \
        hipError_t e_ = (call);                                                                                                 \
        if (e_ != hipSuccess) return set_err(ctx, RT_ERR_DEVICE, std::string(rtk::launch_note() ? rtk::launch_note() : #call) + ": " + hipGetErrorString(e_)); \
    } while (0)

static int render_impl(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm, void* d_out, RtStats* stats) {
    using clk = std::chrono::steady_clock;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Tiling tl; make_tiling(*prm, tl);
    const uint32_t sc = prm->shard_count <= 1u ? 1u : prm->shard_count, si = prm->shard_count <= 1u ? 0u : prm->shard_index;
    rtk::RenderDev rd{};
    auto cp3 = [](float* d, const RtVec3& v) { d[0] = (float)v.x; d[1] = (float)v.y; d[2] = (float)v.z; };
    cp3(rd.cam_origin, cam->origin); cp3(rd.cam_llc, cam->lower_left_corner); cp3(rd.cam_horizontal, cam->horizontal); cp3(rd.cam_vertical, cam->vertical);
    cp3(rd.cam_u, cam->u); cp3(rd.cam_v, cam->v);
    rd.cam_lens_radius = (float)cam->lens_radius; rd.cam_time0 = (float)cam->time0; rd.cam_time1 = (float)cam->time1;
    rd.width = prm->width; rd.height = prm->height; rd.spp = prm->samples_per_pixel; rd.max_depth = prm->max_depth; rd.seed = prm->seed;
    rd.nan_policy = prm->nan_policy; rd.bg_mode = scene->bg_mode; for (int i = 0; i < 3; ++i) rd.bg[i] = scene->bg[i];
    rd.tile_size = tl.ts; rd.tiles_x = tl.tiles_x; rd.tiles_y = tl.tiles_y; rd.shard_index = si; rd.shard_count = sc;
    // in-image pixels per local tile (edge tiles are clipped) -> prefix table
    std::vector<uint32_t> prefix(tl.n_local + 1, 0u);
    uint64_t valid_pixels = 0;
    for (uint32_t lt = 0; lt < tl.n_local; ++lt) {
        const uint32_t tile = si + lt * sc, tx = tile % tl.tiles_x, ty = tile / tl.tiles_x;
        const uint32_t w = std::min(tl.ts, prm->width - tx * tl.ts), h = std::min(tl.ts, prm->height - ty * tl.ts);
        valid_pixels += (uint64_t)w * h;
        if (valid_pixels > 0xFFFFFFFFull) return set_err(ctx, RT_ERR_INVALID, "more than 2^32 - 1 pixels in one shard (shard the image further)");
        prefix[lt + 1] = (uint32_t)valid_pixels;
    }
    // samples per work item: 1 whenever the whole IMAGE (all shards, so that every shard sums the same way) has fewer
    // than 2^32 - 2^28 samples (what a u32 item index can address) — a path is then one sample, nothing is regenerated
    // mid-flight and the radiance of every sample is stored on its own (16 B each: 64 GB for an unsharded render at the
    // limit; this is a 288 GB device, and a sharded render holds its own shard's samples only). Larger renders group
    // 2, 4, ... consecutive samples of a pixel into one item.
    uint32_t block_shift = (prm->flags & RT_FLAG_SAMPLE_BLOCKS) ? 4u : 0u;
    while ((1u << block_shift) > rd.spp && block_shift > 0) --block_shift;
    {
        const uint64_t image_pixels = (uint64_t)prm->width * prm->height;
        while ((1u << block_shift) < rd.spp && image_pixels * ((rd.spp + (1u << block_shift) - 1) >> block_shift) >= kMaxItems) ++block_shift;
    }
    rd.block_shift = block_shift; rd.n_blocks = (rd.spp + (1u << block_shift) - 1) >> block_shift;
    const uint64_t total_items = valid_pixels * rd.n_blocks;
    if (total_items >= kMaxItems) return set_err(ctx, RT_ERR_INVALID, "too many work items for one shard (image too large)");
    rd.total_items = (uint32_t)total_items;
    rd.n_local_tiles = tl.n_local;
    {   // launch-invariant divisors of the item -> (tile, pixel, sample) decode, and how far clipped edge tiles can move a tile index
        const uint64_t ts2 = (uint64_t)tl.ts * tl.ts, item_tile = ts2 * rd.n_blocks;
        rd.div_nblocks = rtk::make_fastdiv(rd.n_blocks); rd.div_width = rtk::make_fastdiv(prm->width); rd.div_ts = rtk::make_fastdiv(tl.ts); rd.div_shards = rtk::make_fastdiv(sc);
        rd.div_ts2 = rtk::make_fastdiv((uint32_t)ts2); rd.div_tiles_x = rtk::make_fastdiv(tl.tiles_x); rd.div_sq_row = rtk::make_fastdiv(std::max(1u, tl.ts >> 3));
        const bool fits = item_tile <= 0xFFFFFFFFull;
        rd.div_item_tile = rtk::make_fastdiv(fits ? (uint32_t)item_tile : 0xFFFFFFFFu);
        const uint64_t clipped = (uint64_t)tl.n_local * ts2 - valid_pixels;
        rd.tile_slack = fits ? (uint32_t)std::min<uint64_t>(tl.n_local, clipped / ts2 + 1) : tl.n_local;
        // one device, every tile: item -> tile by arithmetic (kernels.h RenderDev::row_items)
        const uint64_t row_items = (uint64_t)tl.ts * prm->width * rd.n_blocks;
        const bool by_rows = sc == 1u && fits && row_items <= 0xFFFFFFFFull && !(getenv("RT_TILE_SEARCH") && getenv("RT_TILE_SEARCH")[0] == '1');   // (scripts/ only: the search, for A/B)
        rd.row_items = by_rows ? (uint32_t)row_items : 0u;
        // any shard whose tiles all lie inside the image: the table is a multiplication (the 4096^2 strong-scaling frame at every shard count)
        rd.tiles_all_full = fits && clipped == 0 && !(getenv("RT_TILE_SEARCH") && getenv("RT_TILE_SEARCH")[0] == '1') ? 1u : 0u;
        rd.div_row_items = rtk::make_fastdiv(by_rows ? (uint32_t)row_items : 1u);
    }

    if (stats) { std::memset(stats, 0, sizeof(*stats)); }
    const auto t_begin = clk::now();
    if (total_items == 0) { if (stats) stats->render_ms = 0.0; return RT_OK; }

    // Pool: as many paths in flight as there are work items, up to 2^28 (45 GB for the two pools) and to what the
    // device has free. Launches then carry hundreds of millions of rays: few launches, short tails.
    const size_t rec[6] = {16, 16, 8, 16, 4, block_shift ? (size_t)16 : (size_t)0};   // ray_o ray_d hit s0 sd [s1 = acc, sample]
    size_t slot_bytes = 0; for (int a = 0; a < 6; ++a) slot_bytes += 2 * rec[a];
    uint32_t P = prm->pool_slots ? prm->pool_slots : (1u << 28);
    P = (uint32_t)std::min<uint64_t>(P, total_items);
    if (!prm->pool_slots) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            size_t held = ctx->blocksum.bytes; for (int k = 0; k < 2; ++k) for (int a = 0; a < 6; ++a) held += ctx->pool[k][a].bytes;
            const size_t avail = free_b + held, need_sum = (size_t)total_items * 16;
            const size_t budget = avail * 7 / 10 > need_sum ? avail * 7 / 10 - need_sum : 0;
            while (P > (1u << 20) && (size_t)P * slot_bytes > budget) P >>= 1;
        }
    }
    // Every slot of the first fill owns a LINEAGE of work items — w, w + P, w + 2P, ... — which its path works through one after the other
    // (kernels.h RenderDev::lineage). So that all lineages are equally long (and the pool stays full until the last generation), P is
    // not the cap itself but total / k for the smallest k that fits the cap: 480 M items under a cap of 2^28 run as 2 x 240 M.
    if (!prm->pool_slots && P < total_items) { const uint64_t k = (total_items + P - 1u) / P; P = (uint32_t)((total_items + k - 1u) / k); }
    // kQueues queues of queue_cap slots, filled 512 at a time in turn (k_generate): P is a multiple of 512 * kQueues
    constexpr uint32_t kGrain = 512u * rtk::kQueues;
    P = std::max<uint32_t>(kGrain, (uint32_t)std::min<uint64_t>(((uint64_t)P + kGrain - 1u) / kGrain * kGrain, 0xFFFFF000ull));
    rd.queue_cap = P / rtk::kQueues;
    rtk::PoolDev pd[2];
    for (int k = 0; k < 2; ++k) {
        for (int a = 0; a < 6; ++a) if (rec[a]) HIP_TRY(ctx, ctx->pool[k][a].ensure((size_t)P * rec[a]));
        pd[k].ray_o = (rtd::Float4*)ctx->pool[k][0].p; pd[k].ray_d = (rtd::Float4*)ctx->pool[k][1].p; pd[k].hit = (uint2*)ctx->pool[k][2].p;
        pd[k].s0 = (rtd::Float4*)ctx->pool[k][3].p; pd[k].sd = (uint32_t*)ctx->pool[k][4].p; pd[k].s1 = rec[5] ? (rtd::Float4*)ctx->pool[k][5].p : nullptr;
    }
    HIP_TRY(ctx, ctx->blocksum.ensure((size_t)total_items * 16));
    rd.blocksum = (rtd::Float4*)ctx->blocksum.p;
    // diagnostic: RT_DEBUG_POISON=1 fills the per-item sums with NaN first, so an item no kernel ever finished shows up in the frame
    if (const char* e = getenv("RT_DEBUG_POISON")) if (e[0] == '1') HIP_TRY(ctx, hipMemsetAsync(ctx->blocksum.p, 0xFF, (size_t)total_items * 16, ctx->stream));
    HIP_TRY(ctx, ctx->tile_prefix.ensure(prefix.size() * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->tile_prefix.p, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // `prefix` is a stack vector
    rd.tile_prefix = (const uint32_t*)ctx->tile_prefix.p;
    // counters, one 128-byte line each (they are hit by atomics from every workgroup):
    // line 0 unused since round 3 (it held the queues' next-work-item counters; a path now finds its next item by itself, kernels.h RenderDev::lineage),
    // line 1 queue head, lines 2,3 pool counts; 64-bit statistics from line 4
    // (four counters per queue: next work item, queue head, size of either pool; each on a line of its own)
    constexpr size_t kLine = 128, kQ = rtk::kQueues;
    static_assert(rtk::kQStride * sizeof(uint32_t) == kLine, "queue counters are one line apart");
    HIP_TRY(ctx, ctx->counters.ensure(4 * kQ * kLine + sizeof(unsigned long long) * 16));
    char* cbase = (char*)ctx->counters.p;
    uint32_t* c_head = (uint32_t*)(cbase + 1 * kQ * kLine);
    uint32_t* c_count[2] = {(uint32_t*)(cbase + 2 * kQ * kLine), (uint32_t*)(cbase + 3 * kQ * kLine)};
    unsigned long long* c64 = (unsigned long long*)(cbase + 4 * kQ * kLine);
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters.p, 0, 4 * kQ * kLine + sizeof(unsigned long long) * 16, ctx->stream));

    const bool counting = (prm->flags & RT_FLAG_COUNTERS) != 0, timing = (prm->flags & RT_FLAG_TIMING) != 0;
    rtk::LaunchCfg cfg{};
    uint32_t extend_geometry[2] = {0u, 0u};
    cfg.n_cu = (uint32_t)ctx->n_cu; cfg.extend_geometry = extend_geometry; cfg.features = scene->features; cfg.scene_in_lds = scene->in_lds;

    size_t ev_used = 0;
    auto next_event_on = [&](hipEvent_t& ev, hipStream_t st) -> hipError_t {
        if (ev_used == ctx->events.size()) { hipEvent_t e; hipError_t r = hipEventCreate(&e); if (r != hipSuccess) return r; ctx->events.push_back(e); }
        ev = ctx->events[ev_used++];
        return hipEventRecord(ev, st);
    };
    auto next_event = [&](hipEvent_t& ev) -> hipError_t { return next_event_on(ev, ctx->stream); };
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timing) HIP_TRY(ctx, next_event(e0));
    const uint32_t n_init = (uint32_t)std::min<uint64_t>(P, total_items);
    rd.n_init = n_init; rd.lineage = P;
    rd.q_lo = 0u; rd.q_n = rtk::kQueues; rd.q_shift = rtk::kQShift;
    // the ground sphere tested where a ray is made (kernels.h RenderDev::first_in_shade): a scene without motion, one such sphere, no counting
    rd.first_in_shade = scene->first_id != 0u && rtk::can_test_first_in_shade(scene->features) && !counting && !(getenv("RT_FIRST_IN_SHADE") && getenv("RT_FIRST_IN_SHADE")[0] == '0') ? 1u : 0u;
    rd.first_id = scene->first_id; for (int k = 0; k < 8; ++k) rd.first_prim[k] = scene->first_prim[k];
    HIP_TRY(ctx, rtk::launch_generate(pd[0], rd, n_init, c_count[0], ctx->stream));
    if (timing) { HIP_TRY(ctx, next_event(e1)); spans.push_back({e0, e1, 2}); }
    // The host never waits for an iteration it has just enqueued: the kernels read the pool size from device memory and size-check
    // themselves, so the host only needs (a) an UPPER BOUND of the pool size to size k_shade's grid and (b) to learn that it has
    // reached 0. After every iteration the size is copied to a pinned ring slot behind an event; before enqueuing the next
    // iteration the host takes whatever copies have landed (the size never grows, so an older value is a valid bound) and blocks
    // only when it is kAhead iterations ahead. (Batches of 4, 8, 16, 32 iterations with a blocking read in between sized eight
    // launches of k_shade by the 268 M paths of the start while 5 M were alive — 0.6 ms each for empty workgroups.)
    constexpr uint32_t kAhead = 3, kRing = 4;
    // The tail goes to the drain kernel: once at most `drain_at` paths are alive, ONE launch carries each of them to its end
    // (kernels.hip DRAIN). RT_FLAG_FUSED hands the whole render to it (a diagnostic: bit-identical frame, slower).
    // Measured on the bench workload (LDS-resident scene): hand-over at 2^18 paths 103.2 ms, never 104.1, at 2^21 106.5, at 2^24 121. Scene in
    // HBM (config-5 stand-in, 2048^2 x 32, near-first record orders): 2^18 126.0 ms, 2^20 126.2, 2^21 131.7, 2^22 143.4, 2^24 193.8. (With ONE
    // record order a k_extend launch of that scene did not get shorter than 1.5-2 ms however few rays it carried — the longest far-first walk
    // of the launch — and 2^21 was the best hand-over: 185.9 ms against 211.5 at 2^18.)
    uint32_t drain_at = prm->tail_paths ? prm->tail_paths : (1u << 18);          // RtParams.tail_paths (1 = never: a pool holds 8 queues of >= 1 path)
    if (const char* e = getenv("RT_DRAIN_AT")) drain_at = (uint32_t)std::strtoul(e, nullptr, 10);   // scripts/ only
    // the 8-wide walk resolves hits that tie within rounding in its own way; handing a tail to the per-path kernel (binary records) would
    // make such pixels depend on WHEN the hand-over happens, so a wide scene's wavefront loop runs to its end
    if (scene->dev.wide != nullptr) drain_at = 0u;
    if (prm->flags & RT_FLAG_FUSED) drain_at = 0xFFFFFFFFu;

    // ---- lanes: the pool's 8 queues as ONE wavefront loop, or as TWO halves of 4 queues on two streams ------------------------------------
    // k_extend (an LDS-resident scene) is bound by the LDS array and VALU issue and touches HBM for 40 bytes a segment; k_shade waits on
    // memory latency for 58 % of its wave time and leaves the LDS idle. Run one after the other each has the chip to itself and leaves
    // the other's unit unused. With two halves in flight, half A's k_shade runs WHILE half B's k_extend does: k_extend is launched with
    // half of a CU's resident slots (cfg.extend_share = 2) and the shading workgroups of the other half fill the rest. The extends of the
    // two lanes are chained by events (A1, B1, A2, B2, ...) so that at any time one extend and one shade are in flight; a lane's shade
    // follows its extend in stream order. The picture does not depend on any of this (per-item sums; RNG keyed by pixel and sample).
    struct Pending { hipEvent_t ev; uint32_t ring; };
    struct Lane {
        hipStream_t st; uint32_t q_lo, q_n, q_shift;
        uint32_t live; int cur; uint32_t launched, drained; bool done;
        std::vector<Pending> pending; size_t head;
    };
    bool overlap = false;
    if (const char* e = getenv("RT_OVERLAP")) overlap = e[0] == '1';
    if (prm->flags & RT_FLAG_FUSED) overlap = false;
    if (overlap && !ctx->stream2) { if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) overlap = false; }
    const uint32_t n_lanes = overlap ? 2u : 1u, per_lane = rtk::kQueues / n_lanes;
    Lane lanes[2];
    // upper bound of the size of the LARGEST queue from here on (a queue never grows): it sizes the grids, and 0 ends the render
    const uint32_t live0 = std::min<uint32_t>(rd.queue_cap, (n_init + rtk::kQueues - 1u) / rtk::kQueues + 512u);
    for (uint32_t l = 0; l < n_lanes; ++l) lanes[l] = Lane{l == 0 ? ctx->stream : ctx->stream2, l * per_lane, per_lane, rtk::kQShift - (overlap ? 1u : 0u), live0, 0, 0u, 0u, false, {}, 0};
    if (overlap) {   // the second stream starts behind k_generate
        hipEvent_t eg = nullptr; HIP_TRY(ctx, next_event(eg));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, eg, 0));
    }
    cfg.extend_share = overlap ? 2u : 1u;
    std::vector<uint32_t> iter_live;     // RT_DEBUG_ITER=1 (with RT_FLAG_TIMING): the host's bound of the largest queue at every iteration
    hipEvent_t last_extend = nullptr;    // overlap: end of the most recent k_extend of either lane
    auto take = [&](Lane& L, const Pending& pd_) {
        uint32_t m = 0; for (uint32_t k = 0; k < L.q_n; ++k) m = std::max(m, ctx->h_count[pd_.ring * rtk::kQueues + L.q_lo + k]);
        L.live = std::min(L.live, m);
    };
    // one step of a lane: nothing (done), the drain launch, or one wavefront iteration. Returns a HIP/RT status through `rc`.
    auto step = [&](Lane& L, uint32_t lane_index) -> int {
        while (L.head < L.pending.size() && hipEventQuery(L.pending[L.head].ev) == hipSuccess) take(L, L.pending[L.head++]);
        if (L.live != 0u && L.pending.size() - L.head >= kAhead) {
            HIP_TRY(ctx, hipEventSynchronize(L.pending[L.head].ev));
            take(L, L.pending[L.head++]);
        }
        if (L.live == 0u) { L.done = true; return RT_OK; }
        rtk::RenderDev r = rd; r.q_lo = L.q_lo; r.q_n = L.q_n; r.q_shift = L.q_shift;
        if ((uint64_t)L.live * rtk::kQueues <= drain_at) {
            hipEvent_t ea = nullptr, eb = nullptr;
            if (timing) HIP_TRY(ctx, next_event_on(ea, L.st));
            LAUNCH_TRY(rtk::launch_drain(cfg, scene->dev, pd[L.cur], r, L.live, c_count[L.cur], c_head, c_count[1 - L.cur], c64, counting, L.st));
            if (timing) { HIP_TRY(ctx, next_event_on(eb, L.st)); spans.push_back({ea, eb, 3}); }
            L.drained = L.live * L.q_n;
            L.done = true;
            return RT_OK;
        }
        hipEvent_t ea = nullptr, eb = nullptr, ec = nullptr;
        if (overlap && last_extend) HIP_TRY(ctx, hipStreamWaitEvent(L.st, last_extend, 0));     // extends alternate between the lanes
        if (timing) HIP_TRY(ctx, next_event_on(ea, L.st));
        cfg.max_rays = (uint32_t)std::min<uint64_t>((uint64_t)L.live * L.q_n, 0xFFFFFFFFull);
        LAUNCH_TRY(rtk::launch_extend(cfg, scene->dev, pd[L.cur], r, c_count[L.cur], c_head, c_count[1 - L.cur], c64, counting, L.st));
        if (timing || overlap) HIP_TRY(ctx, next_event_on(eb, L.st));
        if (overlap) last_extend = eb;
        HIP_TRY(ctx, rtk::launch_shade(cfg, scene->dev, pd[L.cur], pd[1 - L.cur], r, L.live, c_count[L.cur], c_count[1 - L.cur], c_head, c64, counting, L.st));
        if (timing) { HIP_TRY(ctx, next_event_on(ec, L.st)); spans.push_back({ea, eb, 0}); spans.push_back({eb, ec, 1}); if (lane_index == 0u) iter_live.push_back(L.live); }
        L.cur = 1 - L.cur;
        const uint32_t ring = (L.launched % kRing) + lane_index * kRing;
        HIP_TRY(ctx, hipMemcpy2DAsync(ctx->h_count + ring * rtk::kQueues, sizeof(uint32_t), c_count[L.cur], kLine, sizeof(uint32_t), rtk::kQueues, hipMemcpyDeviceToHost,
                                      L.st));   // the kQueues pool sizes, one per line (a lane reads its own queues' entries)
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, next_event_on(ev, L.st));
        L.pending.push_back({ev, ring});
        if (++L.launched > 100000000u) return set_err(ctx, RT_ERR_DEVICE, "render loop did not terminate");
        return RT_OK;
    };
    for (;;) {
        bool any = false;
        for (uint32_t l = 0; l < n_lanes; ++l) if (!lanes[l].done) { const int rc = step(lanes[l], l); if (rc != RT_OK) return rc; any = any || !lanes[l].done; }
        if (!any) break;
    }
    if (overlap) {   // the resolve (first stream) waits for the second lane
        hipEvent_t ej = nullptr; HIP_TRY(ctx, next_event_on(ej, ctx->stream2));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ej, 0));
    }
    const uint32_t launched = lanes[0].launched + (n_lanes > 1 ? lanes[1].launched : 0u), drained = lanes[0].drained + (n_lanes > 1 ? lanes[1].drained : 0u);
    hipEvent_t r0 = nullptr, r1 = nullptr;
    if (timing) HIP_TRY(ctx, next_event(r0));
    if (sc > 1) HIP_TRY(ctx, hipMemsetAsync(d_out, 0, (size_t)tl.n_local * tl.ts * tl.ts * 3 * sizeof(float), ctx->stream));   // clipped pixels of edge tiles stay 0
    HIP_TRY(ctx, rtk::launch_resolve(rd, (float*)d_out, (uint32_t)valid_pixels, ctx->stream));
    if (timing) { HIP_TRY(ctx, next_event(r1)); spans.push_back({r0, r1, 2}); }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, c64, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        stats->render_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
        for (const Span& s : spans) {
            float ms = 0.f; if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
            if (s.kind == 0) stats->extend_ms += ms; else if (s.kind == 1) stats->shade_ms += ms; else if (s.kind == 3) stats->drain_ms += ms; else stats->other_ms += ms;
        }
        if (const char* e = getenv("RT_DEBUG_ITER")) if (e[0] == '1' && timing) {
            size_t it = 0;
            for (size_t k = 0; k + 1 < spans.size(); ++k) {
                if (spans[k].kind != 0 || spans[k + 1].kind != 1 || it >= iter_live.size()) continue;
                float a = 0.f, b = 0.f; hipEventElapsedTime(&a, spans[k].a, spans[k].b); hipEventElapsedTime(&b, spans[k + 1].a, spans[k + 1].b);
                std::fprintf(stderr, "iter %3zu  paths <= %10llu  k_extend %9.1f us  k_shade %8.1f us\n", it, (unsigned long long)iter_live[it] * rtk::kQueues, a * 1e3, b * 1e3);
                ++it;
            }
        }
        stats->samples = valid_pixels * rd.spp;
        stats->segments = ctx->h_counters[rtk::CTR_SEGMENTS];
#if defined(RT_STAMPS) || defined(RT_SHADE_STAMPS)
        const bool copy_counts = true;    // k_extend's pass statistics travel in these slots (scripts/gpu_stamps.py)
#else
        const bool copy_counts = counting;
#endif
        if (copy_counts) {
            stats->node_tests = ctx->h_counters[rtk::CTR_NODE_TESTS];
            for (int k = 0; k < RT_N_PRIM_TYPES; ++k) stats->prim_tests[k] = ctx->h_counters[rtk::CTR_PRIM_TESTS + k];
        }
        for (int k = 0; k < 5; ++k) stats->debug[k] = ctx->h_counters[rtk::CTR_DEBUG + k];
#ifdef RT_STAMPS
        stats->debug[5] = ctx->h_counters[rtk::CTR_SAMPLES];       // node steps of all waves (scripts/gpu_stamps.py)
#endif
        stats->debug[6] = extend_geometry[0]; stats->debug[7] = extend_geometry[1];   // k_extend: threads per workgroup, resident workgroups per CU
        stats->iterations = (uint32_t)ctx->h_counters[rtk::CTR_ITERATIONS]; stats->extend_launches = launched; stats->shade_launches = launched; stats->pool_slots = P;
        stats->n_devices = 1u; stats->lds_top_nodes = scene->dev.n_top; stats->drain_paths = drained;
        stats->scene_nodes = scene->n_nodes; stats->scene_prims = scene->n_prims; stats->scene_bytes = scene->bytes; stats->bvh_in_lds = scene->in_lds ? 1u : 0u;
    }
    return RT_OK;
}

int rt_render_device(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm, void* rgb_sum_device, RtStats* stats) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!scene || !cam || !rgb_sum_device) return set_err(ctx, RT_ERR_INVALID, "scene / cam / output is null");
    const int v = validate_params(ctx, prm); if (v != RT_OK) return v;
    return render_checked(ctx, scene, cam, prm, rgb_sum_device, stats);
}

int rt_render(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm, float* rgb_sum_host, RtStats* stats) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!scene || !cam || !rgb_sum_host) return set_err(ctx, RT_ERR_INVALID, "scene / cam / output is null");
    const int v = validate_params(ctx, prm); if (v != RT_OK) return v;
    uint64_t n = 0; rt_output_floats(prm, &n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->out_tmp.ensure(n * sizeof(float)));
    const auto t0 = std::chrono::steady_clock::now();
    const int r = render_checked(ctx, scene, cam, prm, ctx->out_tmp.p, stats);
    if (r != RT_OK) return r;
    HIP_TRY(ctx, hipMemcpyAsync(rgb_sum_host, ctx->out_tmp.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_untile(const RtParams* p, const float* gathered, float* rgb_sum) { return untile_host<float>(p, gathered, rgb_sum); }
int rt_untile_rgb8(const RtParams* p, const uint8_t* gathered, uint8_t* rgb8) { return untile_host<uint8_t>(p, gathered, rgb8); }

int rt_scene_compile_info(const RtSceneDesc* desc, RtCompileInfo* out) { return rt_scene_compile_info_ex(desc, nullptr, out); }
int rt_scene_compile_info_ex(const RtSceneDesc* desc, const RtUploadOptions* options, RtCompileInfo* out) {
    if (!desc || !out) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    { std::string why; const int ok = check_options(options, why); if (ok != RT_OK) return set_err(nullptr, ok, why); }
    rtc::CompiledScene cs;
    const int rc = rtc::compile_scene(*desc, resolve_options(options).compile, cs);
    if (rc != RT_OK) return set_err(nullptr, rc, "scene: " + cs.error);
    out->n_nodes = cs.nodes.size(); out->n_box_nodes = cs.n_box_nodes; out->n_spheres = cs.sphere_meta.size(); out->n_moving = cs.moving_meta.size();
    out->n_rects = cs.rect_meta.size(); out->n_tris = cs.tri_meta.size(); out->n_media = cs.media.size(); out->n_xforms = cs.xforms.size();
    out->n_lights = cs.lights.size(); out->n_materials = cs.mat_b.size();
    out->features = scene_features(cs);
    out->fits_lds = lds_scene_bytes(cs) <= kLdsSceneBudget ? 1u : 0u;
    {   // what a walk tests before it enters the tree: the root list's every-ray members, then the root BVH's scene-sized spheres
        std::vector<uint32_t> first = cs.prologue;
        if (cs.first_leaf != 0u) first.push_back(cs.first_leaf);
        out->n_first = (uint32_t)std::min<size_t>(first.size(), 4); out->_pad = 0u;
        for (uint32_t k = 0; k < 4u; ++k) out->first[k] = k < first.size() ? first[k] : 0u;
    }
    return RT_OK;
}

int rt_scene_top_layout_check(const RtSceneDesc* desc, uint32_t max_top, uint64_t* out_n_top) {
    if (!desc) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    rtc::CompiledScene cs;
    const int rc = rtc::compile_scene(*desc, cs);
    if (rc != RT_OK) return set_err(nullptr, rc, "scene: " + cs.error);
    DevNodes dn;
    if (out_n_top) *out_n_top = 0;
    if (!device_nodes(cs.nodes, max_top, dn)) return set_err(nullptr, RT_ERR_UNSUPPORTED, "node array beyond 4 GB");
    if (out_n_top) *out_n_top = dn.n_top;
    const size_t n = cs.nodes.size();
    const uint32_t top_bytes = dn.n_top * 32u, special = top_bytes + (uint32_t)n * 32u, done = special, idle = special + 32u, end = top_bytes + dn.records() * 32u;
    auto rec = [&](uint32_t addr) -> rtd::NodeDev {
        rtd::NodeDev d;
        std::memcpy(&d, addr < top_bytes ? dn.top.data() + addr : dn.main.data() + (addr - top_bytes), 32);
        return d;
    };
    auto word = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto parks = [&](uint32_t addr, uint32_t resume, uint32_t payload) {
        if (addr < special || addr + 32u > end) return false;
        const rtd::NodeDev t = rec(addr);
        return t.skip_bytes == addr && t.leaf == addr && t.hx < 0.f && t.hy < 0.f && t.hz < 0.f && word(t.cx) == resume && word(t.cy) == payload;
    };
    // the walk that passes every box (and is moved on from every twin) must enumerate the records in pre-order
    std::vector<uint32_t> addr(n + 1);
    uint32_t a = 0;
    for (size_t i = 0; i < n; ++i) {
        addr[i] = a;
        if (a >= special) return set_err(nullptr, RT_ERR_DEVICE, "walk left the records early");
        const rtd::NodeDev d = rec(a), h = rec(top_bytes + (uint32_t)i * 32u);
        if (std::memcmp(&d, &h, 32) != 0) return set_err(nullptr, RT_ERR_DEVICE, "top copy differs from the array's record");
        if (cs.nodes[i].leaf != 0u) {
            if (!parks(d.leaf, d.skip_bytes, cs.nodes[i].leaf)) return set_err(nullptr, RT_ERR_DEVICE, "park twin of record " + std::to_string(i));
            a = d.skip_bytes;                                         // the twin's resume address
        } else a = d.leaf;
    }
    addr[n] = a;
    if (a != done || !parks(done, done, rtd::LEAF_DONE) || !parks(idle, idle, rtd::LEAF_IDLE)) return set_err(nullptr, RT_ERR_DEVICE, "closing records");
    for (size_t i = 0; i < n; ++i)
        if (rec(addr[i]).skip_bytes != addr[std::min<size_t>(cs.nodes[i].skip, n)]) return set_err(nullptr, RT_ERR_DEVICE, "skip link of record " + std::to_string(i));
    return RT_OK;
}

int rt_scene_compile_dump(const RtSceneDesc* desc, void* nodes, uint64_t cap_nodes, float* spheres, uint32_t* sphere_meta, uint64_t cap_spheres) {
    return rt_scene_compile_dump_ex(desc, nullptr, nodes, cap_nodes, spheres, sphere_meta, cap_spheres);
}
int rt_scene_compile_dump_ex(const RtSceneDesc* desc, const RtUploadOptions* options, void* nodes, uint64_t cap_nodes, float* spheres, uint32_t* sphere_meta,
                             uint64_t cap_spheres) {
    if (!desc) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    { std::string why; const int ok = check_options(options, why); if (ok != RT_OK) return set_err(nullptr, ok, why); }
    rtc::CompiledScene cs;
    const int rc = rtc::compile_scene(*desc, resolve_options(options).compile, cs);
    if (rc != RT_OK) return set_err(nullptr, rc, "scene: " + cs.error);
    if ((nodes && cap_nodes < cs.nodes.size()) || ((spheres || sphere_meta) && cap_spheres < cs.sphere_meta.size())) return set_err(nullptr, RT_ERR_INVALID, "capacity too small");
    if (nodes) std::memcpy(nodes, cs.nodes.data(), cs.nodes.size() * sizeof(rtd::Node));
    if (spheres) std::memcpy(spheres, cs.spheres.data(), cs.spheres.size() * sizeof(rtd::Float4));
    if (sphere_meta) std::memcpy(sphere_meta, cs.sphere_meta.data(), cs.sphere_meta.size() * 4);
    return RT_OK;
}

int rt_scene_wide_layout_check(const RtSceneDesc* desc, RtWideInfo* out) {
    if (!desc || !out) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    rtc::CompiledScene cs;
    rtc::CompileOptions copt; copt.big_spheres_first = false;      // (the 8-wide walk tests nothing before the tree)
    const int rc = rtc::compile_scene(*desc, copt, cs);
    if (rc != RT_OK) return set_err(nullptr, rc, "scene: " + cs.error);
    if (!rtw::eligible(cs.nodes)) return set_err(nullptr, RT_ERR_UNSUPPORTED, "not a static BVH: the scene keeps the binary walk");
    double extent = 0.0;
    for (int a = 0; a < 3; ++a) extent = std::max(extent, std::max(std::fabs((double)cs.nodes[0].mn[a]), std::fabs((double)cs.nodes[0].mx[a])));
    const float margin = (float)(6e-7 * extent);
    rtw::WideTree wt; std::string err;
    if (!rtw::build(cs.nodes, margin, wt, err)) return set_err(nullptr, RT_ERR_UNSUPPORTED, err);
    // true box of one primitive (f64 from the compiled f32 data: what the kernel's tests see)
    struct B { double mn[3], mx[3]; };
    auto prim_box = [&](uint32_t type, uint32_t idx, B& b) {
        auto set = [&](int a, double lo, double hi) { b.mn[a] = lo; b.mx[a] = hi; };
        if (type == rtd::LT_SPHERE) { const rtd::Float4 s = cs.spheres[idx]; const double r = std::fabs((double)s.w); set(0, s.x - r, s.x + r); set(1, s.y - r, s.y + r); set(2, s.z - r, s.z + r); }
        else if (type == rtd::LT_TRI) {
            for (int a = 0; a < 3; ++a) { double lo = 1e300, hi = -1e300; for (int v = 0; v < 3; ++v) { const rtd::Float4 p = cs.tris[3 * idx + v]; const double c = a == 0 ? p.x : a == 1 ? p.y : p.z; lo = std::min(lo, c); hi = std::max(hi, c); } set(a, lo, hi); }
        } else if (type == rtd::LT_RECT) {
            const rtd::Float4 r0 = cs.rects[2 * idx], r1 = cs.rects[2 * idx + 1]; const int k = (int)r1.y, ia = k == 0 ? 1 : 0, ib = k == 2 ? 1 : 2;
            set(k, r1.x, r1.x); set(ia, r0.x, r0.y); set(ib, r0.z, r0.w);
        } else { const rtd::Float4 b0 = cs.boxes[2 * idx], b1 = cs.boxes[2 * idx + 1]; set(0, b0.x, b0.y); set(1, b0.z, b0.w); set(2, b1.x, b1.y); }
    };
    std::vector<uint8_t> seen[8];
    seen[rtd::LT_SPHERE].assign(cs.sphere_meta.size(), 0); seen[rtd::LT_TRI].assign(cs.tri_meta.size(), 0);
    seen[rtd::LT_RECT].assign(cs.rect_meta.size(), 0); seen[rtd::LT_BOX].assign(cs.boxes.size() / 2, 0);
    std::string bad;
    uint64_t used_slots = 0;
    // content box of a wide node, bottom-up; every entry's decoded box must contain the content below it
    std::vector<int> state(wt.n_nodes, 0);
    std::function<bool(uint32_t, B&, uint32_t)> walk = [&](uint32_t w, B& content, uint32_t depth) -> bool {
        if (w >= wt.n_nodes || state[w] != 0) { bad = "a wide node is referenced twice or out of range"; return false; }
        state[w] = 1;
        if (depth >= rtd::WIDE_MAX_DEPTH) { bad = "deeper than the walk's stack"; return false; }
        const uint32_t* W = wt.words.data() + (size_t)w * 32u;
        float org[3]; std::memcpy(&org[0], &W[3], 4); std::memcpy(&org[1], &W[7], 4); std::memcpy(&org[2], &W[11], 4);
        float scl[3]; for (int a = 0; a < 3; ++a) { const uint32_t bits = ((W[15] >> (8 * a)) & 0xFFu) << 23; std::memcpy(&scl[a], &bits, 4); }
        for (int a = 0; a < 3; ++a) { content.mn[a] = 1e300; content.mx[a] = -1e300; }
        for (int k = 0; k < 8; ++k) {
            const uint32_t* c = W + 4 * k;
            if (c[0] == 0u) continue;
            ++used_slots;
            const uint32_t q[6] = {c[1] & 0xFFu, (c[1] >> 8) & 0xFFu, (c[1] >> 16) & 0xFFu, c[1] >> 24, c[2] & 0xFFu, (c[2] >> 8) & 0xFFu};
            B below;
            if (c[0] >> 31) {
                const uint32_t type = (c[0] >> 28) & 7u, cnt = (c[0] >> 24) & 15u, first = c[0] & rtd::LEAF_MAX_FIRST;
                if (cnt == 0u || cnt > 8u || seen[type].empty()) { bad = "leaf entry with a bad kind or count"; return false; }
                for (int a = 0; a < 3; ++a) { below.mn[a] = 1e300; below.mx[a] = -1e300; }
                for (uint32_t m = 0; m < cnt; ++m) {
                    if (first + m >= seen[type].size() || seen[type][first + m]++) { bad = "a primitive is in two leaf entries or out of range"; return false; }
                    B pb; prim_box(type, first + m, pb);
                    for (int a = 0; a < 3; ++a) { below.mn[a] = std::min(below.mn[a], pb.mn[a]); below.mx[a] = std::max(below.mx[a], pb.mx[a]); }
                }
                out->n_prims += cnt;
            } else if (!walk(c[0] - 1u, below, depth + 1)) return false;
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)std::fmaf((float)q[a], scl[a], org[a]), hi = (double)std::fmaf((float)q[3 + a], scl[a], org[a]);
                if (!(lo <= below.mn[a] - margin && hi >= below.mx[a] + margin)) { bad = "an entry's decoded box does not contain what is below it"; return false; }
                content.mn[a] = std::min(content.mn[a], below.mn[a]); content.mx[a] = std::max(content.mx[a], below.mx[a]);
            }
        }
        return true;
    };
    B all;
    if (!walk(0u, all, 0u)) return set_err(nullptr, RT_ERR_DEVICE, "wide tree: " + bad);
    for (uint32_t t : {(uint32_t)rtd::LT_SPHERE, (uint32_t)rtd::LT_TRI, (uint32_t)rtd::LT_RECT, (uint32_t)rtd::LT_BOX})
        for (size_t i = 0; i < seen[t].size(); ++i) {
            // (spheres that only bound a medium, and the rects that are a Box's sides, are not in any leaf: a static BVH has no media, and box sides are reached through their Box)
            if (seen[t][i] == 0 && !(t == rtd::LT_RECT && !cs.boxes.empty())) return set_err(nullptr, RT_ERR_DEVICE, "wide tree: a primitive is in no leaf entry");
        }
    for (uint32_t w = 0; w < wt.n_nodes; ++w) if (!state[w]) return set_err(nullptr, RT_ERR_DEVICE, "wide tree: an unreachable node");
    out->n_nodes = wt.n_nodes; out->n_leaf_entries = wt.n_leaf_entries; out->n_inner_entries = wt.n_inner_entries; out->depth = wt.depth;
    out->mean_children = wt.n_nodes ? (double)used_slots / wt.n_nodes : 0.0;
    out->mean_leaf_members = wt.n_leaf_entries ? (double)out->n_prims / wt.n_leaf_entries : 0.0;
    return RT_OK;
}

int rt_test_fail_next_renders(RtCtx* ctx, uint32_t n) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    ctx->fail_renders = n;
    return RT_OK;
}

int rt_resolve_device(RtCtx* ctx, const void* rgb_sum_device, uint32_t width, uint32_t height, uint32_t spp, void* rgb8_device) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!rgb_sum_device || !rgb8_device || spp == 0) return set_err(ctx, RT_ERR_INVALID, "bad argument");
    if (width == 0 || height == 0 || (uint64_t)width * height * 3u > 0xFFFFFFFFull) return set_err(ctx, RT_ERR_INVALID, "bad image size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, rtk::launch_write_color((const float*)rgb_sum_device, width * height, spp, (uint8_t*)rgb8_device, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

