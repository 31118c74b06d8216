// rt_multi.cpp — multi-GPU entry points of include/rt_hip.h: the framebuffer tile-sharded over the GPUs of one node and
// gathered on the root with ONE grouped ncclSend/ncclRecv exchange (RCCL over xGMI), then put in place by a kernel on the root.
//
// Replaces the reference's pixel loops main.rs:730-784 for a whole node. The path has no other exchange step: pixels are
// independent and the scene is replicated, so there is no data-path collective besides this gather. xGMI is point to point
// (each peer has its own link to the root), so the exchange is n-1 direct sends, not a ring; with RT_OUT_RGB8 the shards are
// tone-mapped (write_color, main.rs:141-169) before they travel: 3 bytes per pixel instead of 12.
//
// RCCL is loaded with dlopen at the first multi-GPU call (librccl.so.1 is a 570 MB library the single-GPU path never needs; a
// process that already holds it — PyTorch does — shares that copy).
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "rt_internal.hpp"

using namespace rti;

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

// Mapped objects of the process whose file name starts with `stem` (dl_iterate_phdr), each path once.
struct MapScan { const char* stem; std::vector<std::string> paths; };
int map_scan_cb(struct dl_phdr_info* info, size_t, void* data) {
    MapScan* m = (MapScan*)data;
    if (!info->dlpi_name || !info->dlpi_name[0]) return 0;
    const char* base = std::strrchr(info->dlpi_name, '/');
    base = base ? base + 1 : info->dlpi_name;
    if (std::strncmp(base, m->stem, std::strlen(m->stem)) != 0) return 0;
    for (const std::string& p : m->paths) if (p == info->dlpi_name) return 0;
    m->paths.push_back(info->dlpi_name);
    return 0;
}
std::vector<std::string> mapped(const char* stem) { MapScan m{stem, {}}; dl_iterate_phdr(map_scan_cb, &m); return m.paths; }

const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        // ONE librccl per process. PyTorch bundles its own (torch/lib/librccl.so, soname librccl.so.1) and a process that ends up with that
        // copy AND /opt/rocm's holds two RCCLs over two HIP runtimes — the round-2 abort ("double free or corruption" at exit, DESIGN.md
        // section 6). So: a copy that is already mapped is the copy (opened again by its own path, which only takes a reference); two mapped
        // copies are refused; only a process without any gets one loaded by name. RT_RCCL_LIB names a file for the last case.
        const std::vector<std::string> have = mapped("librccl.so");
        if (have.size() > 1) { g_rccl.error = "two copies of librccl are mapped in this process (" + have[0] + ", " + have[1] + "): refusing to use either"; return; }
        if (have.size() == 1) g_rccl.handle = dlopen(have[0].c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        else {
            const char* env = getenv("RT_RCCL_LIB");
            if (env && env[0]) g_rccl.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : names) { if (g_rccl.handle) break; g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
            // the copy just loaded must sit on the HIP runtime this library runs on, not bring a second one
            std::string listing, why;
            if (g_rccl.handle && !runtime_libraries_ok(listing, why)) { g_rccl.error = "loading librccl mapped a second ROCm runtime: " + why; return; }
        }
        if (!g_rccl.handle) { const char* de = dlerror(); g_rccl.error = std::string("cannot load librccl: ") + (de ? de : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(g_rccl.handle, n); if (!p && g_rccl.error.empty()) g_rccl.error = std::string("librccl lacks ") + n; return p; };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
        g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
        g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
        g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
        g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    });
    return g_rccl.error.empty() ? &g_rccl : nullptr;
}
#define NCCL_TRY(ctx, call)                                                                                            \
    do {                                                                                                               \
        ncclResult_t r_ = (call);                                                                                      \
        if (r_ != ncclSuccess) return set_err(ctx, RT_ERR_DEVICE, std::string(#call) + ": " + R->GetErrorString(r_));  \
    } while (0)

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RT_COMM_ID_BYTES is the size of an ncclUniqueId");

struct ShardGeom { Tiling tl; uint64_t per_shard_px; };   // tiles of shard 0 (the largest) * ts^2
int shard_geom(const RtParams& prm, uint32_t world, ShardGeom& g) {
    RtParams q = prm; q.shard_index = 0; q.shard_count = world;
    if (make_tiling(q, g.tl) != RT_OK) return RT_ERR_INVALID;
    g.per_shard_px = (uint64_t)g.tl.n_local * g.tl.ts * g.tl.ts;
    return RT_OK;
}

// ---- failure agreement: before any shard moves, every rank learns whether EVERY rank has a shard to give -------------------------------
// Each peer sends one status word to the root and gets the verdict word back: two exchanges of 4 bytes per peer on the context's stream
// (tens of microseconds beside a frame of tens of milliseconds). Verdict 0 = go; else 1 + the lowest rank that failed. Without it a
// rank whose render failed (out of memory, a HIP error) returned early while the root had already posted ncclRecv for its shard and then
// waited in hipStreamSynchronize for ever (round-2 ADVICE; the ABI promises error codes, never hangs). Now every rank returns: the one
// that failed its own error, the others RT_ERR_PEER naming it. Only a failure of the COMMUNICATOR itself (an RCCL error inside this
// function) is beyond agreement; it comes back as RT_ERR_DEVICE.
constexpr int kMaxWorld = 64;
int agree_on_status(RtCtx* ctx, const Rccl* R, int my_status, int* failed_rank) {
    *failed_rank = -1;
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    if (world < 2) { if (my_status != RT_OK) *failed_rank = rank; return RT_OK; }
    if (world > kMaxWorld || !ctx->comm_words.p || !ctx->h_words) return set_err(ctx, RT_ERR_INVALID, "communicator without its status words (rt_comm_init_rank)");
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    uint32_t* dev = (uint32_t*)ctx->comm_words.p;   // [0] this rank's word, [1] the verdict, [2 + r] rank r's word on the root
    uint32_t* host = ctx->h_words;
    host[0] = my_status != RT_OK ? 1u : 0u;
    if (rank == 0) {
        NCCL_TRY(ctx, R->GroupStart());
        for (int r = 1; r < world; ++r) {
            const ncclResult_t q = R->Recv(dev + 2 + r, 1, ncclUint32, r, comm, ctx->stream);
            if (q != ncclSuccess) { (void)R->GroupEnd(); return set_err(ctx, RT_ERR_DEVICE, std::string("ncclRecv (status): ") + R->GetErrorString(q)); }
        }
        NCCL_TRY(ctx, R->GroupEnd());
        HIP_TRY(ctx, hipMemcpyAsync(host + 2, dev + 2, sizeof(uint32_t) * (size_t)world, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        host[2] = host[0];
        uint32_t verdict = 0u;
        for (int r = world - 1; r >= 0; --r) if (host[2 + r] != 0u) verdict = (uint32_t)r + 1u;
        host[1] = verdict;
        HIP_TRY(ctx, hipMemcpyAsync(dev + 1, host + 1, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        NCCL_TRY(ctx, R->GroupStart());
        for (int r = 1; r < world; ++r) {
            const ncclResult_t q = R->Send(dev + 1, 1, ncclUint32, r, comm, ctx->stream);
            if (q != ncclSuccess) { (void)R->GroupEnd(); return set_err(ctx, RT_ERR_DEVICE, std::string("ncclSend (verdict): ") + R->GetErrorString(q)); }
        }
        NCCL_TRY(ctx, R->GroupEnd());
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (verdict != 0u) *failed_rank = (int)verdict - 1;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(dev, host, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        NCCL_TRY(ctx, R->Send(dev, 1, ncclUint32, 0, comm, ctx->stream));
        NCCL_TRY(ctx, R->Recv(dev + 1, 1, ncclUint32, 0, comm, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(host + 1, dev + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (host[1] != 0u) *failed_rank = (int)host[1] - 1;
    }
    return RT_OK;
}
// what the verdict means for this rank: its own error if it failed, RT_ERR_PEER naming the first failed rank otherwise
int called_off(RtCtx* ctx, int my_status, const std::string& my_error, int failed_rank) {
    if (my_status != RT_OK) { ctx->err = my_error; g_last_error = my_error; return my_status; }
    return set_err(ctx, RT_ERR_PEER, "rank " + std::to_string(failed_rank) + " failed its part of the render: the exchange was called off on every rank");
}
int attach_comm(RtCtx* ctx, void* comm, int rank, int world) {
    HIP_TRY(ctx, ctx->comm_words.ensure(sizeof(uint32_t) * (size_t)(kMaxWorld + 2)));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return RT_OK;
}

// One rank's part of a sharded render: render shard `rank` of `world`, optional write_color on the shard, agreement, exchange, and on the
// root the untile kernel into `frame_device`. The same code serves one-process-per-GPU (rt_render_gather) and the threads of
// rt_render_multi (one communicator per device from ncclCommInitAll).
int render_gather_rank(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* prm_in, uint32_t kind, void* frame_device, RtStats* stats) {
    const Rccl* R = nullptr;
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    if (world > 1) {
        R = rccl();
        if (!R) return set_err(ctx, RT_ERR_DEVICE, g_rccl.error);
        if (!ctx->comm) return set_err(ctx, RT_ERR_INVALID, "no communicator on this context (rt_comm_init_rank)");
    }
    // argument errors are the same on every rank (same params, same call): they return before anything is posted
    if (kind > RT_OUT_RGB8) return set_err(ctx, RT_ERR_INVALID, "bad output kind");
    RtParams prm = *prm_in;
    prm.shard_index = (uint32_t)rank; prm.shard_count = (uint32_t)world;
    if (prm.tile_size == 0) prm.tile_size = 32;
    const int v = validate_params(ctx, &prm); if (v != RT_OK) return v;
    ShardGeom g; if (shard_geom(prm, (uint32_t)world, g) != RT_OK) return set_err(ctx, RT_ERR_INVALID, "bad tiling parameters");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t elem = kind == RT_OUT_RGB8 ? 1 : 4;
    const size_t shard_elems = (size_t)g.per_shard_px * 3;
    const auto t0 = std::chrono::steady_clock::now();
    if (world == 1) {
        // one GPU: no tiles to move. f32: render straight into the caller's frame; u8: render, then write_color into it
        if (!frame_device) return set_err(ctx, RT_ERR_INVALID, "rank 0 needs a frame buffer");
        prm.shard_count = 1; prm.shard_index = 0;
        if (kind == RT_OUT_RGB_SUM_F32) return render_checked(ctx, scene, cam, &prm, frame_device, stats);
        uint64_t n = 0; rt_output_floats(&prm, &n);
        HIP_TRY(ctx, ctx->shard_tmp.ensure(n * 4));
        const int r = render_checked(ctx, scene, cam, &prm, ctx->shard_tmp.p, stats);
        if (r != RT_OK) return r;
        HIP_TRY(ctx, rtk::launch_write_color((const float*)ctx->shard_tmp.p, prm.width * prm.height, prm.samples_per_pixel, (uint8_t*)frame_device, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (stats) { stats->n_devices = 1; stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
        return RT_OK;
    }
    // ---- this rank's part; a failure from here on is NOT returned before the other ranks know of it ----
    int my = RT_OK; std::string my_error;
    auto fail = [&](int code, const std::string& msg) { if (my == RT_OK) { my = code; my_error = msg; } };
    if (rank == 0 && !frame_device) fail(RT_ERR_INVALID, "rank 0 needs a frame buffer");
    // f32 shard of this rank; the root's lives at slot 0 of the gather buffer when the exchange is f32
    if (ctx->shard_tmp.ensure(shard_elems * 4 + (kind == RT_OUT_RGB8 ? shard_elems : 0)) != hipSuccess) fail(RT_ERR_OOM, "out of device memory for this rank's shard");
    float* shard_f32 = (float*)ctx->shard_tmp.p;
    uint8_t* shard_u8 = (uint8_t*)ctx->shard_tmp.p + shard_elems * 4;
    void* gather = nullptr;
    if (rank == 0) { if (ctx->out_tmp.ensure(shard_elems * elem * (size_t)world) != hipSuccess) fail(RT_ERR_OOM, "out of device memory for the gathered shards"); gather = ctx->out_tmp.p; }
    void* mine = shard_f32;                                      // what this rank contributes, in the exchange's element type
    if (rank == 0 && kind == RT_OUT_RGB_SUM_F32) mine = gather;  // slot 0
    if (my == RT_OK) {
        const int r = render_checked(ctx, scene, cam, &prm, kind == RT_OUT_RGB_SUM_F32 ? mine : (void*)shard_f32, stats);
        if (r != RT_OK) fail(r, ctx->err);
    }
    {
        int failed = -1;
        const int a = agree_on_status(ctx, R, my, &failed);
        if (a != RT_OK) return a;
        if (failed >= 0) return called_off(ctx, my, my_error, failed);
    }
    if (!ctx->ev_gather[0]) { HIP_TRY(ctx, hipEventCreate(&ctx->ev_gather[0])); HIP_TRY(ctx, hipEventCreate(&ctx->ev_gather[1])); }
    hipEvent_t e0 = ctx->ev_gather[0], e1 = ctx->ev_gather[1];
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    if (kind == RT_OUT_RGB8) {
        // write_color on the shard (an elementwise map over its compact tile buffer; clipped pixels are 0 and stay 0)
        uint8_t* dst = rank == 0 ? (uint8_t*)gather : shard_u8;
        HIP_TRY(ctx, rtk::launch_write_color(shard_f32, (uint32_t)g.per_shard_px, prm.samples_per_pixel, dst, ctx->stream));
        mine = dst;
    }
    // ---- the one exchange of the path: every peer sends its shard straight to the root (its own xGMI link) ----
    const ncclDataType_t dt = kind == RT_OUT_RGB8 ? ncclUint8 : ncclFloat32;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    if (rank == 0) {
        NCCL_TRY(ctx, R->GroupStart());
        for (int r = 1; r < world; ++r) {
            const ncclResult_t q = R->Recv((char*)gather + (size_t)r * shard_elems * elem, shard_elems, dt, r, comm, ctx->stream);
            if (q != ncclSuccess) { (void)R->GroupEnd(); return set_err(ctx, RT_ERR_DEVICE, std::string("ncclRecv: ") + R->GetErrorString(q)); }
        }
        NCCL_TRY(ctx, R->GroupEnd());
        if (kind == RT_OUT_RGB8) HIP_TRY(ctx, rtk::launch_untile_u8((const uint8_t*)gather, (uint8_t*)frame_device, prm.width, prm.height, g.tl.ts, g.tl.tiles_x,
                                                                    (uint32_t)world, (uint64_t)shard_elems, ctx->stream));
        else HIP_TRY(ctx, rtk::launch_untile_f32((const float*)gather, (float*)frame_device, prm.width, prm.height, g.tl.ts, g.tl.tiles_x, (uint32_t)world,
                                                 (uint64_t)shard_elems, ctx->stream));
    } else {
        NCCL_TRY(ctx, R->Send(mine, shard_elems, dt, 0, comm, ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
        float ms = 0.f; if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) stats->gather_ms = ms;
        stats->n_devices = (uint32_t)world;
        stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

}  // namespace

bool rti::runtime_libraries_ok(std::string& listing, std::string& why) {
    listing.clear(); why.clear();
    bool ok = true;
    for (const char* stem : {"libamdhip64.so", "libhsa-runtime64.so", "librccl.so"}) {
        const std::vector<std::string> have = mapped(stem);
        for (const std::string& p : have) listing += p + "\n";
        // libhsa-runtime64 is listed but not judged: under rocprofv3 the profiler's tool library brings the system's copy next to the one
        // the HIP runtime in use was linked against, and only the latter is ever initialised. Two HIP runtimes (or two RCCLs) are two
        // sets of device state in use at once.
        if (have.size() > 1 && why.empty() && std::strcmp(stem, "libhsa-runtime64.so") != 0) {
            ok = false;
            why = std::string("two copies of ") + stem + " are mapped in this process (" + have[0] + ", " + have[1] + "): two ROCm runtimes with separate device state, "
                  "which ends in heap corruption at exit. PyTorch bundles its own copies under other file names; load it BEFORE this library (its copies carry the "
                  "system's sonames and are then shared)";
        }
    }
    return ok;
}

void rti::comm_release(RtCtx* ctx) {
    if (ctx && ctx->comm) { const Rccl* R = rccl(); if (R) (void)R->CommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; ctx->comm_world = 1; ctx->comm_rank = 0; }
}

// The host threads of the peer devices (one per device beyond the root), started with the context and parked on a condition variable
// between frames: a frame hands worker i the job of device i + 1 and waits for all of them. (Round 2 spawned and joined n - 1 threads
// per frame.)
class DeviceWorkers {
  public:
    void start(int n) {
        job_.assign((size_t)n, nullptr); seen_.assign((size_t)n, 0);
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
    }
    // hand jobs[i] to worker i (the vector must stay alive until wait() has returned)
    void post(const std::vector<std::function<void()>>& jobs) {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t i = 0; i < jobs.size() && i < job_.size(); ++i) job_[i] = &jobs[i];
        pending_ = (int)std::min(jobs.size(), job_.size());
        ++generation_;
        cv_job_.notify_all();
    }
    // ... and return when every one of them has finished
    void wait() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this] { return pending_ == 0; });
    }
    void stop() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_job_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
        th_.clear();
    }
    ~DeviceWorkers() { stop(); }
  private:
    void loop(int i) {
        for (;;) {
            const std::function<void()>* f = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_job_.wait(lk, [&] { return stop_ || (generation_ != seen_[(size_t)i] && job_[(size_t)i] != nullptr); });
                if (stop_) return;
                seen_[(size_t)i] = generation_; f = job_[(size_t)i]; job_[(size_t)i] = nullptr;
            }
            (*f)();
            { std::lock_guard<std::mutex> lk(mu_); if (--pending_ == 0) cv_done_.notify_all(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_; std::condition_variable cv_job_, cv_done_;
    std::vector<const std::function<void()>*> job_; std::vector<uint64_t> seen_;
    uint64_t generation_ = 0; int pending_ = 0; bool stop_ = false;
};

struct RtMultiCtx {
    DeviceWorkers workers;
    std::vector<RtCtx*> ctx;
    std::string err;
    void* host_pinned = nullptr; size_t host_pinned_bytes = 0;
    DevBuf frame;   // full frame on the root device before it goes to the host
};
struct RtMultiScene { std::vector<RtScene*> scene; };

extern "C" {

int rt_runtime_libraries(char* out, uint64_t cap) {
    std::string listing, why;
    const bool ok = runtime_libraries_ok(listing, why);
    if (out && cap) { const size_t n = std::min<size_t>(listing.size(), (size_t)cap - 1); std::memcpy(out, listing.data(), n); out[n] = 0; }
    return ok ? RT_OK : set_err(nullptr, RT_ERR_DEVICE, why);
}

int rt_test_device_workers(int n_workers, int rounds) {
    if (n_workers < 0 || n_workers > 64 || rounds < 0) return set_err(nullptr, RT_ERR_INVALID, "n_workers / rounds");
    DeviceWorkers w;
    w.start(n_workers);
    std::vector<int> ran((size_t)n_workers, 0);
    int root = 0;
    for (int r = 0; r < rounds; ++r) {
        std::vector<std::function<void()>> jobs;
        for (int i = 0; i < n_workers; ++i) jobs.emplace_back([&ran, i, r] { if (ran[(size_t)i] == r) ++ran[(size_t)i]; });
        w.post(jobs);
        ++root;                   // (the calling thread's own part of a frame)
        w.wait();
    }
    w.stop();
    for (int i = 0; i < n_workers; ++i) if (ran[(size_t)i] != rounds) return set_err(nullptr, RT_ERR_DEVICE, "a worker missed or repeated a frame");
    return root == rounds ? RT_OK : RT_ERR_DEVICE;
}

int rt_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return set_err(nullptr, RT_ERR_INVALID, "id_out is null");
    const Rccl* R = rccl();
    if (!R) return set_err(nullptr, RT_ERR_DEVICE, g_rccl.error);
    ncclUniqueId id;
    NCCL_TRY(nullptr, R->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_init_rank(RtCtx* ctx, const uint8_t* id_in, int rank, int world) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!id_in || world < 1 || rank < 0 || rank >= world) return set_err(ctx, RT_ERR_INVALID, "bad id / rank / world");
    const Rccl* R = rccl();
    if (!R) return set_err(ctx, RT_ERR_DEVICE, g_rccl.error);
    comm_release(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id; std::memcpy(id.internal, id_in, RT_COMM_ID_BYTES);
    if (world > kMaxWorld) return set_err(ctx, RT_ERR_INVALID, "world larger than 64");
    ncclComm_t comm = nullptr;
    NCCL_TRY(ctx, R->CommInitRank(&comm, world, id, rank));
    return attach_comm(ctx, comm, rank, world);
}

int rt_comm_selftest(RtCtx* ctx) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!ctx->comm) return set_err(ctx, RT_ERR_INVALID, "no communicator on this context (rt_comm_init_rank)");
    const Rccl* R = rccl();
    if (!R) return set_err(ctx, RT_ERR_DEVICE, g_rccl.error);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    constexpr size_t n = 1 << 16;
    std::vector<uint8_t> src(n), dst(n, 0);
    for (size_t i = 0; i < n; ++i) src[i] = (uint8_t)((i * 2654435761u) >> 13);
    DevBuf a, b;
    HIP_TRY(ctx, a.ensure(n)); HIP_TRY(ctx, b.ensure(n));
    struct Free { DevBuf &x, &y; ~Free() { x.release(); y.release(); } } fr{a, b};
    HIP_TRY(ctx, hipMemcpyAsync(a.p, src.data(), n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(b.p, 0, n, ctx->stream));
    NCCL_TRY(ctx, R->GroupStart());
    const ncclResult_t s = R->Send(a.p, n, ncclUint8, ctx->comm_rank, (ncclComm_t)ctx->comm, ctx->stream);
    const ncclResult_t r = R->Recv(b.p, n, ncclUint8, ctx->comm_rank, (ncclComm_t)ctx->comm, ctx->stream);
    NCCL_TRY(ctx, R->GroupEnd());
    if (s != ncclSuccess || r != ncclSuccess) return set_err(ctx, RT_ERR_DEVICE, "self send/recv was refused");
    HIP_TRY(ctx, hipMemcpyAsync(dst.data(), b.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (std::memcmp(src.data(), dst.data(), n) != 0) return set_err(ctx, RT_ERR_DEVICE, "self send/recv returned other bytes");
    return RT_OK;
}

int rt_render_gather(RtCtx* ctx, const RtScene* scene, const RtCamera* cam, const RtParams* params, uint32_t output_kind, void* frame_device, RtStats* stats) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!scene || !cam || !params) return set_err(ctx, RT_ERR_INVALID, "scene / cam / params is null");
    return render_gather_rank(ctx, scene, cam, params, output_kind, frame_device, stats);
}

int rt_untile_device(RtCtx* ctx, const RtParams* params, uint32_t kind, const void* gathered_device, void* frame_device) {
    if (!ctx) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!params || !gathered_device || !frame_device || kind > RT_OUT_RGB8) return set_err(ctx, RT_ERR_INVALID, "bad argument");
    const uint32_t world = params->shard_count <= 1u ? 1u : params->shard_count;
    ShardGeom g; if (shard_geom(*params, world, g) != RT_OK) return set_err(ctx, RT_ERR_INVALID, "bad tiling parameters");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t shard_elems = g.per_shard_px * 3;
    if (kind == RT_OUT_RGB8) HIP_TRY(ctx, rtk::launch_untile_u8((const uint8_t*)gathered_device, (uint8_t*)frame_device, params->width, params->height, g.tl.ts, g.tl.tiles_x,
                                                                world, shard_elems, ctx->stream));
    else HIP_TRY(ctx, rtk::launch_untile_f32((const float*)gathered_device, (float*)frame_device, params->width, params->height, g.tl.ts, g.tl.tiles_x, world, shard_elems,
                                             ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

const char* rt_last_error_multi(const RtMultiCtx* m) { return m ? m->err.c_str() : g_last_error.c_str(); }

int rt_ctx_destroy_multi(RtMultiCtx* m) {
    if (!m) return RT_OK;
    m->workers.stop();
    if (!m->ctx.empty() && m->ctx[0]) { (void)hipSetDevice(m->ctx[0]->device); m->frame.release(); if (m->host_pinned) (void)hipHostFree(m->host_pinned); }
    for (RtCtx* c : m->ctx) rt_ctx_destroy(c);
    delete m;
    return RT_OK;
}

int rt_ctx_create_multi(const int* device_ids, int n, RtMultiCtx** out) {
    if (!out) return set_err(nullptr, RT_ERR_INVALID, "out_ctx is null");
    *out = nullptr;
    if (!device_ids || n < 1 || n > 64) return set_err(nullptr, RT_ERR_INVALID, "device_ids / n_devices");
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (device_ids[i] == device_ids[j]) return set_err(nullptr, RT_ERR_INVALID, "a device is listed twice");
    RtMultiCtx* m = new RtMultiCtx();
    for (int i = 0; i < n; ++i) {
        RtCtx* c = nullptr;
        const int r = rt_ctx_create(device_ids[i], nullptr, &c);
        if (r != RT_OK) { rt_ctx_destroy_multi(m); return r; }
        m->ctx.push_back(c);
    }
    const Rccl* R = rccl();
    if (!R) { rt_ctx_destroy_multi(m); return set_err(nullptr, RT_ERR_DEVICE, g_rccl.error); }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    const ncclResult_t q = R->CommInitAll(comms.data(), n, device_ids);   // single process: one communicator per device
    if (q != ncclSuccess) { rt_ctx_destroy_multi(m); return set_err(nullptr, RT_ERR_DEVICE, std::string("ncclCommInitAll: ") + R->GetErrorString(q)); }
    for (int i = 0; i < n; ++i) {
        const int r = attach_comm(m->ctx[i], comms[i], i, n);
        if (r != RT_OK) { const std::string why = m->ctx[i]->err; for (int k = i + 1; k < n; ++k) (void)R->CommDestroy(comms[k]); rt_ctx_destroy_multi(m); return set_err(nullptr, r, why); }
    }
    m->workers.start(n - 1);
    *out = m;
    return RT_OK;
}

int rt_scene_destroy_multi(RtMultiCtx* m, RtMultiScene* s) {
    if (!s) return RT_OK;
    for (size_t i = 0; i < s->scene.size(); ++i) rt_scene_destroy(m && i < m->ctx.size() ? m->ctx[i] : nullptr, s->scene[i]);
    delete s;
    return RT_OK;
}

int rt_scene_upload_multi_ex(RtMultiCtx* m, const RtSceneDesc* desc, const RtUploadOptions* options, RtMultiScene** out) {
    if (!m || !desc || !out) return set_err(nullptr, RT_ERR_INVALID, "null argument");
    *out = nullptr;
    // compiled and laid out ONCE (the 1.26 M-primitive tree of config 5 takes seconds to build), then copied to every device
    SceneImage* im = nullptr; std::string err;
    const int rc = scene_image_build(desc, options, &im, err);
    if (rc != RT_OK) { m->err = err; return set_err(nullptr, rc, err); }
    RtMultiScene* s = new RtMultiScene();
    for (RtCtx* c : m->ctx) {
        RtScene* sc = nullptr;
        const int r = scene_image_upload(c, *im, &sc);
        if (r != RT_OK) { m->err = c->err; scene_image_free(im); rt_scene_destroy_multi(m, s); return set_err(nullptr, r, m->err); }
        s->scene.push_back(sc);
    }
    scene_image_free(im);
    *out = s;
    return RT_OK;
}
int rt_scene_upload_multi(RtMultiCtx* m, const RtSceneDesc* desc, RtMultiScene** out) { return rt_scene_upload_multi_ex(m, desc, nullptr, out); }

static int render_multi(RtMultiCtx* m, const RtMultiScene* s, const RtCamera* cam, const RtParams* prm, uint32_t kind, void* host_out, RtStats* stats) {
    if (!m) return set_err(nullptr, RT_ERR_INVALID, "ctx is null");
    if (!s || !cam || !prm || !host_out || s->scene.size() != m->ctx.size()) { m->err = "scene / cam / params / output"; return set_err(nullptr, RT_ERR_INVALID, m->err); }
    const int n = (int)m->ctx.size();
    const auto t0 = std::chrono::steady_clock::now();
    RtCtx* root = m->ctx[0];
    if (hipSetDevice(root->device) != hipSuccess) { m->err = "hipSetDevice"; return RT_ERR_DEVICE; }
    const size_t elem = kind == RT_OUT_RGB8 ? 1 : 4, frame_bytes = (size_t)prm->width * prm->height * 3 * elem;
    if (m->frame.ensure(frame_bytes) != hipSuccess) { m->err = "out of device memory for the frame"; return set_err(nullptr, RT_ERR_OOM, m->err); }
    if (m->host_pinned_bytes < frame_bytes) {
        if (m->host_pinned) (void)hipHostFree(m->host_pinned);
        m->host_pinned = nullptr; m->host_pinned_bytes = 0;
        if (hipHostMalloc(&m->host_pinned, frame_bytes, hipHostMallocDefault) == hipSuccess) m->host_pinned_bytes = frame_bytes;
    }
    // one host thread per device: each runs its own wavefront loop (it has host round trips) and its side of the exchange
    std::vector<int> rc((size_t)n, RT_OK);
    std::vector<RtStats> st((size_t)n);
    std::vector<std::function<void()>> jobs;
    for (int i = 1; i < n; ++i) jobs.emplace_back([&, i] { rc[i] = render_gather_rank(m->ctx[i], s->scene[i], cam, prm, kind, nullptr, &st[i]); });
    m->workers.post(jobs);          // the peers' host threads were started with the context
    rc[0] = render_gather_rank(root, s->scene[0], cam, prm, kind, m->frame.p, &st[0]);   // the root's part on the calling thread
    m->workers.wait();
    // the device that failed its own part speaks first (the others only report RT_ERR_PEER, "rank k failed")
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < n; ++i)
            if (rc[i] != RT_OK && (pass == 1 || rc[i] != RT_ERR_PEER)) { m->err = "device " + std::to_string(m->ctx[i]->device) + ": " + m->ctx[i]->err; return set_err(nullptr, rc[i], m->err); }
    // frame -> host (through pinned memory when it could be had)
    (void)hipSetDevice(root->device);
    hipError_t e = hipSuccess;
    if (m->host_pinned) {
        e = hipMemcpyAsync(m->host_pinned, m->frame.p, frame_bytes, hipMemcpyDeviceToHost, root->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(root->stream);
        if (e == hipSuccess) std::memcpy(host_out, m->host_pinned, frame_bytes);
    } else {
        e = hipMemcpyAsync(host_out, m->frame.p, frame_bytes, hipMemcpyDeviceToHost, root->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(root->stream);
    }
    if (e != hipSuccess) { m->err = std::string("frame copy: ") + hipGetErrorString(e); return set_err(nullptr, RT_ERR_DEVICE, m->err); }
    if (stats) {
        *stats = st[0];
        for (int i = 1; i < n; ++i) {
            stats->samples += st[i].samples; stats->segments += st[i].segments; stats->node_tests += st[i].node_tests;
            for (int k = 0; k < RT_N_PRIM_TYPES; ++k) stats->prim_tests[k] += st[i].prim_tests[k];
            stats->extend_ms = std::max(stats->extend_ms, st[i].extend_ms); stats->shade_ms = std::max(stats->shade_ms, st[i].shade_ms);
            stats->other_ms = std::max(stats->other_ms, st[i].other_ms);
        }
        stats->n_devices = (uint32_t)n;
        stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RT_OK;
}

int rt_render_multi(RtMultiCtx* m, const RtMultiScene* s, const RtCamera* cam, const RtParams* prm, float* rgb_sum_host, RtStats* stats) {
    return render_multi(m, s, cam, prm, RT_OUT_RGB_SUM_F32, rgb_sum_host, stats);
}
int rt_render_multi_rgb8(RtMultiCtx* m, const RtMultiScene* s, const RtCamera* cam, const RtParams* prm, uint8_t* rgb8_host, RtStats* stats) {
    return render_multi(m, s, cam, prm, RT_OUT_RGB8, rgb8_host, stats);
}

}  // extern "C"
