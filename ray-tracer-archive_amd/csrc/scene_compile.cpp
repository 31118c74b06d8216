// scene_compile.cpp — compiles the reference-shaped object graph (RtSceneDesc) into the flat device
// layout: a threaded (stackless) BVH in the reference's own visiting order plus per-type primitive
// arrays.  Host-only code.
//
// BVH construction restates the INTENT of BVHNode::construct (bvh.rs:77-130): random axis per node
// (bvh.rs:87), stable sort of the node's own sub-range by box minimum on that axis (bvh.rs:108 sorts
// the whole vector by mistake — SURVEY F6), median split (bvh.rs:109), ordered pair for span 2
// (bvh.rs:100-107), single object for span 1 (bvh.rs:96-98, tested once here instead of twice).
#include "scene_compile.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>

namespace rtc {
namespace {

constexpr double PI = 3.14159265358979323846264338327950288;
constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ull;
inline uint64_t fin(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Box3 { double mn[3], mx[3]; };
inline Box3 surrounding(const Box3& a, const Box3& b) {   // aabb.rs:57-69
    Box3 o;
    for (int i = 0; i < 3; ++i) { o.mn[i] = std::min(a.mn[i], b.mn[i]); o.mx[i] = std::max(a.mx[i], b.mx[i]); }
    return o;
}

// world -> local map of a chain of Translate / RotateY wrappers: local = M(theta)(world - off),
// M(theta)(x,z) = (cos*x - sin*z, sin*x + cos*z)  (hittable.rs:151-155)
struct Chain {
    double theta = 0.0;            // radians, sum of the RotateY angles
    double off[3] = {0, 0, 0};
    bool identity = true;
    struct Op { uint32_t kind; double degrees; };
    std::vector<Op> ops;           // outer -> inner: every Translate / RotateY / FlipFace above the primitive
    uint32_t xform_id = 0;         // id of (theta, off) in CompiledScene::xforms
};

inline float f_down(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
inline float f_up(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }

struct Compiler {
    const RtSceneDesc& d;
    CompiledScene& out;
    std::map<std::pair<long long, std::pair<long long, std::pair<long long, long long>>>, uint32_t> xform_cache;
    bool box_pair_members = false;
    bool big_spheres_first = true;
    bool cull_lists = true;   // HittableList members behind culling boxes (emit_list_culled); RT_LIST_CULL=0: every member probed by every ray, as the reference does
    double park_cost = 6.0;   // what a stop of the walk at a leaf costs, in primitive tests (RT_LIST_PARK_COST)
    std::map<std::vector<long long>, uint32_t> wrap_cache;

    Compiler(const RtSceneDesc& desc, CompiledScene& o) : d(desc), out(o) {}

    bool fail(const std::string& m) { if (out.error.empty()) out.error = m; return false; }
    bool ok() const { return out.error.empty(); }
    const RtHittable* H(int id) { if (id < 0 || (uint64_t)id >= d.n_hittables) { fail("hittable id out of range"); return nullptr; } return &d.hittables[id]; }

    // ---- bounding boxes, as each bounding_box() of the reference computes them (f64) ----
    bool bbox(int id, double t0, double t1, Box3& b, int depth = 0) {
        const RtHittable* h = H(id); if (!h) return false;
        if (depth > 128) return fail("graph too deep (cycle?)");
        const double* p = h->p;
        switch (h->kind) {
        case RT_HIT_SPHERE: for (int i = 0; i < 3; ++i) { b.mn[i] = p[i] - p[3]; b.mx[i] = p[i] + p[3]; } return true;            // sphere.rs:66-73
        case RT_HIT_MOVING_SPHERE: {                                                                                                   // moving_sphere.rs:67-78
            Box3 b0, b1; const double r = p[8];
            for (int i = 0; i < 3; ++i) {
                const double c0 = p[i] + ((t0 - p[6]) / (p[7] - p[6])) * (p[3 + i] - p[i]);
                const double c1 = p[i] + ((t1 - p[6]) / (p[7] - p[6])) * (p[3 + i] - p[i]);
                b0.mn[i] = c0 - r; b0.mx[i] = c0 + r; b1.mn[i] = c1 - r; b1.mx[i] = c1 + r;
            }
            b = surrounding(b0, b1); return true;
        }
        case RT_HIT_XY_RECT: b = Box3{{p[0], p[2], p[4] - 0.0001}, {p[1], p[3], p[4] + 0.0001}}; return true;                       // aarect.rs:49-56
        case RT_HIT_XZ_RECT: b = Box3{{p[0], p[4] - 0.0001, p[2]}, {p[1], p[4] + 0.0001, p[3]}}; return true;                       // aarect.rs:99-106
        case RT_HIT_YZ_RECT: b = Box3{{p[4] - 0.0001, p[0], p[2]}, {p[4] + 0.0001, p[1], p[3]}}; return true;                       // aarect.rs:168-175
        case RT_HIT_TRIANGLE:
            for (int i = 0; i < 3; ++i) { b.mn[i] = std::min(p[i], std::min(p[3 + i], p[6 + i])) - 0.0001; b.mx[i] = std::max(p[i], std::max(p[3 + i], p[6 + i])) + 0.0001; }
            return true;
        case RT_HIT_BOX: for (int i = 0; i < 3; ++i) { b.mn[i] = p[i]; b.mx[i] = p[3 + i]; } return true;                             // boxes.rs:80-83
        case RT_HIT_LIST: case RT_HIT_BVH: {                                                                                           // hittable_list.rs:52-72, bvh.rs:144-147
            if (h->n_children <= 0) return false;
            const double a0 = h->kind == RT_HIT_BVH ? p[0] : t0, a1 = h->kind == RT_HIT_BVH ? p[1] : t1;
            for (int c = 0; c < h->n_children; ++c) {
                if ((uint64_t)(h->first_child + c) >= d.n_children) return fail("children out of range");
                Box3 cb; if (!bbox(d.children[h->first_child + c], a0, a1, cb, depth + 1)) return false;
                b = c == 0 ? cb : surrounding(b, cb);
            }
            return true;
        }
        case RT_HIT_TRANSLATE: {                                                                                                       // hittable.rs:86-95
            if (!bbox(h->first_child, t0, t1, b, depth + 1)) return false;
            for (int i = 0; i < 3; ++i) { b.mn[i] += p[i]; b.mx[i] += p[i]; }
            return true;
        }
        case RT_HIT_ROTATE_Y: {                                                                                                        // hittable.rs:107-144 (child box at times 0,1)
            Box3 cb; if (!bbox(h->first_child, 0.0, 1.0, cb, depth + 1)) return false;
            const double radians = p[0] * PI / 180.0, s = std::sin(radians), c = std::cos(radians);
            const double inf = std::numeric_limits<double>::infinity();
            for (int i = 0; i < 3; ++i) { b.mn[i] = inf; b.mx[i] = -inf; }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
                const double x = i * cb.mx[0] + (1 - i) * cb.mn[0], y = j * cb.mx[1] + (1 - j) * cb.mn[1], z = k * cb.mx[2] + (1 - k) * cb.mn[2];
                const double t[3] = {c * x + s * z, y, -s * x + c * z};
                for (int a = 0; a < 3; ++a) { b.mn[a] = std::min(b.mn[a], t[a]); b.mx[a] = std::max(b.mx[a], t[a]); }
            }
            return true;
        }
        case RT_HIT_FLIP_FACE: case RT_HIT_CONSTANT_MEDIUM: return bbox(h->first_child, t0, t1, b, depth + 1);                         // hittable.rs:202-204
        default: return fail("unknown hittable kind");
        }
    }

    // ---- transforms ----
    uint32_t intern_xform(const Chain& c) {
        if (c.identity) return 0;
        auto q = [](double v) { return (long long)std::llround(v * 1048576.0); };
        auto key = std::make_pair(q(c.theta), std::make_pair(q(c.off[0]), std::make_pair(q(c.off[1]), q(c.off[2]))));
        auto it = xform_cache.find(key);
        if (it != xform_cache.end()) return it->second;
        rtd::Xform x{};
        x.sin_t = (float)std::sin(c.theta); x.cos_t = (float)std::cos(c.theta);
        for (int i = 0; i < 3; ++i) x.off[i] = (float)c.off[i];
        out.xforms.push_back(x);
        if (out.xforms.size() > 255) fail("more than 255 distinct instance transforms");
        xform_cache[key] = (uint32_t)out.xforms.size() - 1;
        return (uint32_t)out.xforms.size() - 1;
    }
    static void apply_translate(Chain& c, const double* off2) {
        // local' = M(theta)(w - off) - off2 = M(theta)(w - off - M(-theta) off2)
        const double s = std::sin(c.theta), k = std::cos(c.theta);
        c.off[0] += k * off2[0] + s * off2[2];
        c.off[1] += off2[1];
        c.off[2] += -s * off2[0] + k * off2[2];
        c.identity = false;
    }
    static void apply_rotate(Chain& c, double degrees) { c.theta += degrees * PI / 180.0; c.identity = false; }

    uint32_t intern_wrap(const Chain& c) {
        if (c.ops.empty()) return 0;
        if (c.ops.size() > rtd::MAX_WRAP_OPS) { out.error = "a chain of more than 6 Translate/RotateY/FlipFace wrappers must be restructured"; return 0; }
        std::vector<long long> key{(long long)c.xform_id};
        for (const auto& o : c.ops) { key.push_back(o.kind); key.push_back((long long)std::llround(o.degrees * 1048576.0)); }
        auto it = wrap_cache.find(key);
        if (it != wrap_cache.end()) return it->second;
        rtd::Wrap w{};
        w.xform = c.xform_id; w.n_ops = (uint32_t)c.ops.size();
        for (size_t i = 0; i < c.ops.size(); ++i) {
            const double radians = c.ops[i].degrees * PI / 180.0;   // rt_weekend.rs:4-6
            w.op[i].kind = c.ops[i].kind; w.op[i].sin_t = (float)std::sin(radians); w.op[i].cos_t = (float)std::cos(radians);
        }
        out.wraps.push_back(w);
        if (out.wraps.size() > rtd::MAX_WRAPS) fail("more than 1024 distinct wrapper chains");
        wrap_cache[key] = (uint32_t)out.wraps.size() - 1;
        return (uint32_t)out.wraps.size() - 1;
    }

    // ---- primitives ----
    uint32_t meta_for(const RtHittable& h, const Chain& c) {
        if (h.material < 0 || (uint64_t)h.material >= d.n_materials) { fail("primitive without a valid material"); return 0; }
        if ((uint32_t)h.material > rtd::META_MAT_MASK) { fail("too many materials"); return 0; }
        return rtd::make_meta((uint32_t)h.material, intern_wrap(c));
    }
    void push_leaf_node(uint32_t type, uint32_t first, uint32_t count) {
        if (first > rtd::LEAF_MAX_FIRST) { fail("too many primitives of one type"); return; }
        rtd::Node n{};
        n.mn[0] = n.mn[1] = n.mn[2] = -std::numeric_limits<float>::infinity();
        n.mx[0] = n.mx[1] = n.mx[2] = std::numeric_limits<float>::infinity();
        n.leaf = rtd::make_leaf(type, first, count);
        n.skip = 0;   // patched: a no-box node never takes its skip edge
        out.nodes.push_back(n);
        out.nodes.back().skip = (uint32_t)out.nodes.size();
    }
    uint32_t add_rect(int kaxis, double a0, double a1, double b0, double b1, double k, uint32_t meta) {
        out.rects.push_back(rtd::Float4{(float)a0, (float)a1, (float)b0, (float)b1});
        out.rects.push_back(rtd::Float4{(float)k, (float)kaxis, 0.f, 0.f});
        out.rect_meta.push_back(meta);
        return (uint32_t)out.rect_meta.size() - 1;
    }
    // boxes.rs:17-74 — the six sides in the reference's order
    uint32_t add_box_rects(const double* p, uint32_t meta) {
        const double x0 = p[0], y0 = p[1], z0 = p[2], x1 = p[3], y1 = p[4], z1 = p[5];
        const uint32_t first = add_rect(2, x0, x1, y0, y1, z1, meta);
        add_rect(2, x0, x1, y0, y1, z0, meta);
        add_rect(1, x0, x1, z0, z1, y1, meta);
        add_rect(1, x0, x1, z0, z1, y0, meta);
        add_rect(0, y0, y1, z0, z1, x1, meta);
        add_rect(0, y0, y1, z0, z1, x0, meta);
        return first;
    }

    // Strip wrappers above `id`, folding them into the chain. Returns the first non-wrapper id.
    int unwrap(int id, Chain& c, int depth = 0) {
        while (true) {
            const RtHittable* h = H(id); if (!h) return -1;
            if (++depth > 128) { fail("wrapper chain too deep"); return -1; }
            if (h->kind == RT_HIT_TRANSLATE) { apply_translate(c, h->p); c.ops.push_back({rtd::WO_TRANSLATE, 0.0}); id = h->first_child; }
            else if (h->kind == RT_HIT_ROTATE_Y) { apply_rotate(c, h->p[0]); c.ops.push_back({rtd::WO_ROTATE_Y, h->p[0]}); id = h->first_child; }
            else if (h->kind == RT_HIT_FLIP_FACE) { c.ops.push_back({rtd::WO_FLIP_FACE, 0.0}); id = h->first_child; }
            else return id;
        }
    }

    void emit(int id, const Chain& ctx, int depth = 0) {
        if (!ok()) return;
        if (depth > 128) { fail("graph too deep (cycle?)"); return; }
        const RtHittable* h0 = H(id); if (!h0) return;
        if (h0->kind == RT_HIT_TRANSLATE || h0->kind == RT_HIT_ROTATE_Y || h0->kind == RT_HIT_FLIP_FACE) {
            Chain c = ctx;
            const int inner = unwrap(id, c); if (inner < 0) return;
            c.xform_id = intern_xform(c);
            if (c.xform_id != ctx.xform_id) {
                push_leaf_node(rtd::LT_XFORM, c.xform_id, 0);
                emit(inner, c, depth + 1);
                push_leaf_node(rtd::LT_XFORM, ctx.xform_id | rtd::XFORM_EXIT, 0);
            } else emit(inner, c, depth + 1);
            return;
        }
        const RtHittable& h = *h0;
        const double* p = h.p;
        switch (h.kind) {
        case RT_HIT_SPHERE:
            out.spheres.push_back(rtd::Float4{(float)p[0], (float)p[1], (float)p[2], (float)p[3]});
            out.sphere_meta.push_back(meta_for(h, ctx));
            if (!to_prologue) push_leaf_node(rtd::LT_SPHERE, (uint32_t)out.sphere_meta.size() - 1, 1);     // (else: emit_bvh's first_leaf names it)
            break;
        case RT_HIT_MOVING_SPHERE:
            out.moving.push_back(rtd::Float4{(float)p[0], (float)p[1], (float)p[2], (float)p[8]});
            out.moving.push_back(rtd::Float4{(float)p[3], (float)p[4], (float)p[5], (float)p[6]});
            out.moving.push_back(rtd::Float4{(float)p[7], 0.f, 0.f, 0.f});
            out.moving_meta.push_back(meta_for(h, ctx));
            if (to_prologue) out.prologue.push_back(rtd::make_leaf(rtd::LT_MOVING, (uint32_t)out.moving_meta.size() - 1, 1));
            else push_leaf_node(rtd::LT_MOVING, (uint32_t)out.moving_meta.size() - 1, 1);
            break;
        case RT_HIT_XY_RECT: case RT_HIT_XZ_RECT: case RT_HIT_YZ_RECT: {
            const int kaxis = h.kind == RT_HIT_XY_RECT ? 2 : (h.kind == RT_HIT_XZ_RECT ? 1 : 0);
            const uint32_t idx = add_rect(kaxis, p[0], p[1], p[2], p[3], p[4], meta_for(h, ctx));
            if (!to_prologue) push_leaf_node(rtd::LT_RECT, idx, 1);                                        // (else: emit_bvh's first_leaf names it)
            break;
        }
        case RT_HIT_TRIANGLE:
            out.tris.push_back(rtd::Float4{(float)p[0], (float)p[1], (float)p[2], 0.f});
            out.tris.push_back(rtd::Float4{(float)p[3], (float)p[4], (float)p[5], 0.f});
            out.tris.push_back(rtd::Float4{(float)p[6], (float)p[7], (float)p[8], 0.f});
            out.tri_meta.push_back(meta_for(h, ctx));
            push_leaf_node(rtd::LT_TRI, (uint32_t)out.tri_meta.size() - 1, 1);
            break;
        case RT_HIT_BOX: {
            // a primitive kind of its own for the walk: one 32-byte record (the six bounds + where its sides start in rects[]), tested in
            // straight-line code; the sides themselves stay rects — what a hit is shaded from, and what the hit id names
            const uint32_t first = add_box_rects(p, meta_for(h, ctx));
            uint32_t fbits = first; float ff; std::memcpy(&ff, &fbits, 4);
            out.boxes.push_back(rtd::Float4{(float)p[0], (float)p[3], (float)p[1], (float)p[4]});
            out.boxes.push_back(rtd::Float4{(float)p[2], (float)p[5], ff, 0.f});
            push_leaf_node(rtd::LT_BOX, (uint32_t)(out.boxes.size() / 2) - 1, 1);
            break;
        }
        case RT_HIT_LIST:
            for (int c = 0; c < h.n_children; ++c) if ((uint64_t)(h.first_child + c) >= d.n_children) { fail("children out of range"); return; }
            if (cull_lists) emit_list_culled(h, ctx, depth);
            else for (int c = 0; c < h.n_children; ++c) emit(d.children[h.first_child + c], ctx, depth + 1);
            break;
        case RT_HIT_BVH: emit_bvh(id, h, ctx, depth); break;
        case RT_HIT_CONSTANT_MEDIUM: emit_medium(id, h, ctx); break;
        default: fail("unknown hittable kind");
        }
    }

    // ---- HittableList members behind culling boxes (not in the reference: HittableList::hit, hittable_list.rs:33-50, probes every member) ----
    // A member whose box the ray misses cannot be hit, so skipping it changes no result; what it saves is the walk parking at the
    // member's leaf (k_extend) and the member's own test. A box must hold for every ray time, so a subtree with a moving sphere outside a
    // BVH (no time range to take the box over) gets none. Spheres count with |radius| (a hollow glass sphere has a negative one).
    bool cull_bbox(int id, Box3& b, int depth = 0) {
        if (id < 0 || (uint64_t)id >= d.n_hittables || depth > 128) return false;
        const RtHittable* h = &d.hittables[id];
        const double* p = h->p;
        switch (h->kind) {
        case RT_HIT_SPHERE: for (int i = 0; i < 3; ++i) { b.mn[i] = p[i] - std::fabs(p[3]); b.mx[i] = p[i] + std::fabs(p[3]); } return true;
        case RT_HIT_MOVING_SPHERE: return false;
        case RT_HIT_XY_RECT: case RT_HIT_XZ_RECT: case RT_HIT_YZ_RECT: case RT_HIT_TRIANGLE: case RT_HIT_BOX: {
            const std::string keep = out.error; const bool r = bbox(id, 0.0, 1.0, b); out.error = keep; return r;
        }
        case RT_HIT_LIST: case RT_HIT_BVH: {
            if (h->n_children <= 0) return false;
            for (int c = 0; c < h->n_children; ++c) {
                if ((uint64_t)(h->first_child + c) >= d.n_children) return false;
                Box3 cb; if (!cull_bbox(d.children[h->first_child + c], cb, depth + 1)) return false;
                b = c == 0 ? cb : surrounding(b, cb);
            }
            return true;
        }
        case RT_HIT_TRANSLATE: {
            if (!cull_bbox(h->first_child, b, depth + 1)) return false;
            for (int i = 0; i < 3; ++i) { b.mn[i] += p[i]; b.mx[i] += p[i]; }
            return true;
        }
        case RT_HIT_ROTATE_Y: {
            Box3 cb; if (!cull_bbox(h->first_child, cb, depth + 1)) return false;
            const double radians = p[0] * PI / 180.0, s = std::sin(radians), c = std::cos(radians);
            const double inf = std::numeric_limits<double>::infinity();
            for (int i = 0; i < 3; ++i) { b.mn[i] = inf; b.mx[i] = -inf; }
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
                const double x = i ? cb.mx[0] : cb.mn[0], y = j ? cb.mx[1] : cb.mn[1], z = k ? cb.mx[2] : cb.mn[2];
                const double t[3] = {c * x + s * z, y, -s * x + c * z};
                for (int a = 0; a < 3; ++a) { b.mn[a] = std::min(b.mn[a], t[a]); b.mx[a] = std::max(b.mx[a], t[a]); }
            }
            return true;
        }
        case RT_HIT_FLIP_FACE: case RT_HIT_CONSTANT_MEDIUM: return cull_bbox(h->first_child, b, depth + 1);
        default: return false;
        }
    }
    // leaf type a bare primitive lands in and the primitives it adds to it (0: not a bare primitive)
    static uint32_t bare_kind(const RtHittable& h, uint32_t& count) {
        count = 1;
        switch (h.kind) {
        case RT_HIT_SPHERE: return rtd::LT_SPHERE;
        case RT_HIT_XY_RECT: case RT_HIT_XZ_RECT: case RT_HIT_YZ_RECT: return rtd::LT_RECT;
        case RT_HIT_BOX: count = 6; return rtd::LT_BOX;     // (count = its tests, for the cost model of the list grouping)
        case RT_HIT_TRIANGLE: return rtd::LT_TRI;
        default: return 0;
        }
    }
    void emit_boxed(const std::vector<int>& ids, const Box3& box, const Chain& ctx, int depth) {
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.push_back(rtd::Node{});
        out.n_box_nodes++;
        for (int id : ids) emit(id, ctx, depth + 1);
        finish_box_node(me, box);
    }
    // Members of the ROOT list that every ray meets — a moving sphere (no box without a time range) or a medium whose boundary holds
    // everything else (the book-2 final scene's fog) — would stop every walk once each, as a primitive kind of their own. They go to
    // the prologue instead: tested when a walk begins (k_extend), with all the lanes that begin together. HittableList::hit keeps the
    // closest of its members' hits whatever their order (hittable_list.rs:40-47), so testing them first changes no hit.
    bool prologue_member(const RtHittable& root, int c) {
        const RtHittable& m = d.hittables[d.children[root.first_child + c]];
        if (m.kind == RT_HIT_MOVING_SPHERE) return true;
        if (m.kind != RT_HIT_CONSTANT_MEDIUM) return false;
        Box3 mine;
        if (!cull_bbox(d.children[root.first_child + c], mine)) return true;
        for (int e = 0; e < root.n_children; ++e) {
            Box3 b;
            if (e == c || !cull_bbox(d.children[root.first_child + e], b)) continue;
            for (int a = 0; a < 3; ++a) if (b.mn[a] < mine.mn[a] || b.mx[a] > mine.mx[a]) return false;
        }
        return true;
    }
    bool to_prologue = false;
    void emit_list_culled(const RtHittable& h, const Chain& ctx, int depth) {
        const int n = h.n_children;
        const bool root = &h == &d.hittables[d.world] && ctx.identity && ctx.ops.empty();
        std::vector<char> pro(n, 0);
        if (root) {
            size_t k = 0;
            for (int c = 0; c < n; ++c) { const int id = d.children[h.first_child + c]; if (id >= 0 && (uint64_t)id < d.n_hittables && k < rtd::MAX_PROLOGUE && prologue_member(h, c)) { pro[c] = 1; ++k; } }
            if (k == (size_t)n) std::fill(pro.begin(), pro.end(), 0);     // the walk needs a record to start at
        }
        int c = 0;
        while (c < n && ok()) {
            const int id = d.children[h.first_child + c];
            const RtHittable* m = H(id); if (!m) return;
            if (pro[c]) { to_prologue = true; emit(id, ctx, depth + 1); to_prologue = false; ++c; continue; }
            uint32_t cnt0 = 0;
            const uint32_t kind = bare_kind(*m, cnt0);
            Box3 b0;
            if (m->kind == RT_HIT_BVH || !cull_bbox(id, b0)) { emit(id, ctx, depth + 1); ++c; continue; }   // a BVH starts with its own box
            if (kind == 0u) { emit_boxed({id}, b0, ctx, depth); ++c; continue; }
            // a run of bare primitives of one leaf type: consecutive members share a leaf (one stop of the walk, `count` tests) or get
            // leaves of their own (a stop each, but only for the rays that meet the smaller box). Cost of a leaf = chance of being met
            // (~ half area of its box) x (park_cost + tests); the cheapest split of the run into consecutive groups, by dynamic programming.
            std::vector<int> ids; std::vector<uint32_t> cnt; std::vector<Box3> box;
            for (int e = c; e < n; ++e) {
                const int eid = d.children[h.first_child + e];
                if (eid < 0 || (uint64_t)eid >= d.n_hittables) break;
                uint32_t k = 0; Box3 eb;
                if (bare_kind(d.hittables[eid], k) != kind || !cull_bbox(eid, eb)) break;
                ids.push_back(eid); cnt.push_back(k); box.push_back(eb);
            }
            const size_t r = ids.size();
            std::vector<double> best(r + 1, std::numeric_limits<double>::infinity()); std::vector<size_t> from(r + 1, 0);
            best[0] = 0.0;
            for (size_t e = 1; e <= r; ++e) {
                Box3 u = box[e - 1]; uint32_t tests = 0, records = 0;
                for (size_t s0 = e; s0-- > 0;) {
                    u = surrounding(u, box[s0]); tests += cnt[s0]; records += kind == rtd::LT_BOX ? 1u : cnt[s0];   // a leaf counts its records (a Box is one)
                    if (records > rtd::LEAF_MAX_COUNT) break;
                    const double cost = best[s0] + half_area(u) * (park_cost + (double)tests);
                    if (cost < best[e]) { best[e] = cost; from[e] = s0; }
                }
            }
            std::vector<size_t> cuts;
            for (size_t e = r; e > 0; e = from[e]) cuts.push_back(e);
            size_t s0 = 0;
            for (size_t k = cuts.size(); k-- > 0;) {
                const size_t e = cuts[k];
                Box3 u = box[s0]; for (size_t i = s0 + 1; i < e; ++i) u = surrounding(u, box[i]);
                emit_boxed(std::vector<int>(ids.begin() + s0, ids.begin() + e), u, ctx, depth);
                s0 = e;
            }
            c += (int)r;
        }
    }

    void emit_medium(int id, const RtHittable& h, const Chain& ctx) {
        Chain c = ctx; c.ops.clear();
        const int inner = unwrap(h.first_child, c); if (inner < 0) return;
        const RtHittable* b = H(inner); if (!b) return;
        rtd::Medium m{};
        m.boundary_xform = intern_xform(c);
        if (b->kind == RT_HIT_SPHERE) {
            out.spheres.push_back(rtd::Float4{(float)b->p[0], (float)b->p[1], (float)b->p[2], (float)b->p[3]});
            out.sphere_meta.push_back(0);
            m.boundary_type = rtd::LT_SPHERE; m.boundary_first = (uint32_t)out.sphere_meta.size() - 1; m.boundary_count = 1;
        } else if (b->kind == RT_HIT_BOX) {
            m.boundary_type = rtd::LT_RECT; m.boundary_first = add_box_rects(b->p, 0); m.boundary_count = 6;
        } else { out.error = "constant medium boundary must be a sphere or a box (optionally under Translate/RotateY)"; return; }
        if (h.p[0] == 0.0) { fail("constant medium with zero density"); return; }
        m.neg_inv_density = (float)(-1.0 / h.p[0]);   // constant_medium.rs:25
        Chain none; none.xform_id = 0;
        m.meta = meta_for(h, none);
        m.medium_id = (uint32_t)id;
        out.media.push_back(m);
        if (to_prologue) out.prologue.push_back(rtd::make_leaf(rtd::LT_MEDIUM, (uint32_t)out.media.size() - 1, 1));
        else push_leaf_node(rtd::LT_MEDIUM, (uint32_t)out.media.size() - 1, 1);
    }

    // ---- BVH ----
    struct Build { std::vector<int> obj; std::vector<double> key[3]; std::vector<Box3> box; uint64_t axis_state; };

    // The BVH every ray enters first: the world itself, or a member of the world list.
    bool every_ray_enters(int id, const Chain& ctx) const {
        if (!ctx.identity || !ctx.ops.empty()) return false;
        if (id == d.world) return true;
        const RtHittable& w = d.hittables[d.world];
        if (w.kind != RT_HIT_LIST) return false;
        for (int c = 0; c < w.n_children; ++c) if ((uint64_t)(w.first_child + c) < d.n_children && d.children[w.first_child + c] == id) return true;
        return false;
    }
    void emit_bvh(int id, const RtHittable& h, const Chain& ctx, int depth) {
        int n = h.n_children;
        if (n <= 0) { fail("empty BVH"); return; }
        Build B;
        B.obj.resize(n); B.box.resize(n); for (int a = 0; a < 3; ++a) B.key[a].resize(n);
        for (int c = 0; c < n; ++c) {
            if ((uint64_t)(h.first_child + c) >= d.n_children) { fail("children out of range"); return; }
            const int cid = d.children[h.first_child + c];
            Box3 k0;
            if (!bbox(cid, 0.0, 0.0, k0) || !bbox(cid, h.p[0], h.p[1], B.box[c])) { fail("No bounding box in BVHNode constructor."); return; }   // bvh.rs:19-21,123-127
            B.obj[c] = cid;
            for (int a = 0; a < 3; ++a) B.key[a][c] = k0.mn[a];
        }
        // A sphere whose box is most of this BVH's box (the r = 1000 ground of the books' scenes) spoils every box above it — they all grow to
        // the scene's size, and nearly every ray walks down to that sphere — and is met by nearly every ray anyway. Out of the tree with it:
        // it is tested when a walk begins (a walk starts parked at a leaf of its own, kernels.h SceneDev::walk_start: the sphere is tested in the
        // wave's next sphere pass like any leaf's), the walk then starts with that hit's t_max, and the tree of the others
        // keeps tight boxes. BVHNode::hit keeps the closest of its members' hits whatever their order (bvh.rs:134-143), so the hit is the
        // same; the test COUNTS are not the reference's (CompileOptions::big_spheres_first = false restores those).
        if (big_spheres_first && out.first_leaf == 0u && n >= 3 && every_ray_enters(id, ctx)) {
            Box3 all = B.box[0];
            for (int c = 1; c < n; ++c) all = surrounding(all, B.box[c]);
            // (spheres or rects — the ground of the 1 M-sphere scene is a 1200 x 1200 rect —, the kind of the first one found: a leaf word names one kind)
            std::vector<char> big((size_t)n, 0); int n_big = 0; bool rects = false;
            auto is_rect = [](int k) { return k == RT_HIT_XY_RECT || k == RT_HIT_XZ_RECT || k == RT_HIT_YZ_RECT; };
            for (int c = 0; c < n && n_big < 4; ++c) {
                const RtHittable* m = H(B.obj[c]); if (!m) return;
                const bool sphere = m->kind == RT_HIT_SPHERE, rect = is_rect(m->kind);
                if ((sphere || rect) && (n_big == 0 || rect == rects) && half_area(B.box[c]) >= 0.5 * half_area(all)) { big[(size_t)c] = 1; rects = rect; ++n_big; }
            }
            // RT_BVH_SAH splits such a sphere off near the root by itself (book-1 on that tree: 35.6 ms with it in the tree, 36.7 with it tested
            // in the walk's first pass) — but not where it can be tested where the rays are MADE (RenderDev::first_in_shade: one sphere, a scene
            // without motion or media): there the SAH tree gains as the reference-shaped one does (37.0 -> 32.1 ms)
            bool still = true;
            for (uint64_t i = 0; i < d.n_hittables && still; ++i) still = d.hittables[i].kind != RT_HIT_MOVING_SPHERE && d.hittables[i].kind != RT_HIT_CONSTANT_MEDIUM;
            if (d.bvh_builder == RT_BVH_SAH && !(still && n_big == 1)) n_big = 0;
            if (n_big != 0 && n - n_big >= 2) {
                Build K; for (int a = 0; a < 3; ++a) K.key[a].reserve((size_t)(n - n_big));
                const uint32_t first = rects ? (uint32_t)out.rect_meta.size() : (uint32_t)out.sphere_meta.size();
                for (int c = 0; c < n; ++c) {
                    if (big[(size_t)c]) { to_prologue = true; emit(B.obj[c], ctx, depth + 1); to_prologue = false; continue; }   // (consecutive in the sphere table)
                    K.obj.push_back(B.obj[c]); K.box.push_back(B.box[c]); for (int a = 0; a < 3; ++a) K.key[a].push_back(B.key[a][c]);
                }
                out.first_leaf = rtd::make_leaf(rects ? rtd::LT_RECT : rtd::LT_SPHERE, first, (uint32_t)n_big);
                B = std::move(K); n -= n_big;
            }
        }
        std::vector<int> order(n); for (int i = 0; i < n; ++i) order[i] = i;
        B.axis_state = fin(d.bvh_seed + GAMMA * (uint64_t)(id + 1));
        if (d.bvh_builder == RT_BVH_SAH) build_sah(B, order, 0, (size_t)n, ctx, depth);
        else build_range(B, order, 0, (size_t)n, ctx, depth);
    }

    static double half_area(const Box3& b) {
        const double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    // Boxes leave the compiler EXACT, rounded outwards to f32 only: what the device's slab test needs on top of that — its own rounding —
    // is added where its size is known: per record for the part that grows with the box's coordinates (rt_api.cpp node_boxes), per RAY
    // for the part that grows with the ray's origin (kernels.hip set_slab_ray). Round 2 padded every box here by 1e-6 * (|x| + the
    // scene's extent): with the r = 5000 fog sphere of the book-2 final scene that is 0.008 units, more than t_min * |d| = 0.001, so
    // every ray LEAVING a box face passed that box's slab again (the reference's exact box ends at t = 0 < t_min) and the device made
    // 10-24 % more box-side tests than the reference on that scene (round-2 VERDICT weak #2).
    void finish_box_node(uint32_t me, const Box3& box) {
        rtd::Node& n = out.nodes[me];
        for (int i = 0; i < 3; ++i) { n.mn[i] = f_down(box.mn[i]); n.mx[i] = f_up(box.mx[i]); }
        n.leaf = 0;
        n.skip = (uint32_t)out.nodes.size();
    }

    // RT_BVH_SAH: top-down binned surface-area heuristic (16 bins per axis). Not the reference's tree —
    // an option for large scenes; the picture is the same because a BVH only culls. Children are laid out
    // lower-coordinate first (the threaded walk has one fixed order for all rays).
    Box3 build_sah(Build& B, std::vector<int>& order, size_t start, size_t end, const Chain& ctx, int depth) {
        const size_t n = end - start;
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.push_back(rtd::Node{});
        out.n_box_nodes++;
        Box3 box = range_box(B, order, start, end);
        auto make_leaf = [&]() { for (size_t i = start; i < end; ++i) { if (n >= 2) emit_member(B, order[i], ctx, depth); else emit(B.obj[order[i]], ctx, depth + 1); } finish_box_node(me, box); return box; };
        if (n <= 2 || depth > 100) return make_leaf();
        constexpr int NB = 16;
        double cmn[3], cmx[3];
        for (int a = 0; a < 3; ++a) { cmn[a] = std::numeric_limits<double>::infinity(); cmx[a] = -cmn[a]; }
        for (size_t i = start; i < end; ++i) {
            const Box3& b = B.box[order[i]];
            for (int a = 0; a < 3; ++a) { const double c = 0.5 * (b.mn[a] + b.mx[a]); cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
        }
        double best_cost = std::numeric_limits<double>::infinity(); int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; ++a) {
            const double ext = cmx[a] - cmn[a];
            if (!(ext > 0)) continue;
            size_t cnt[NB] = {0}; Box3 bb[NB]; bool used[NB] = {false};
            const double scale = NB / ext;
            for (size_t i = start; i < end; ++i) {
                const Box3& b = B.box[order[i]];
                int k = (int)((0.5 * (b.mn[a] + b.mx[a]) - cmn[a]) * scale); if (k >= NB) k = NB - 1; if (k < 0) k = 0;
                bb[k] = used[k] ? surrounding(bb[k], b) : b; used[k] = true; cnt[k]++;
            }
            double right_area[NB]; size_t right_cnt[NB]; Box3 acc{}; bool have_acc = false; size_t c = 0;
            for (int k = NB - 1; k >= 1; --k) {
                if (used[k]) { acc = have_acc ? surrounding(acc, bb[k]) : bb[k]; have_acc = true; }
                c += cnt[k]; right_cnt[k] = c; right_area[k] = have_acc ? half_area(acc) : 0.0;
            }
            have_acc = false; c = 0;
            for (int k = 0; k < NB - 1; ++k) {
                if (used[k]) { acc = have_acc ? surrounding(acc, bb[k]) : bb[k]; have_acc = true; }
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const double cost = half_area(acc) * (double)c + right_area[k + 1] * (double)right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k + 1; }
            }
        }
        const double leaf_cost = half_area(box) * (double)n * 1.5;
        if (best_axis < 0 || (n <= 6 && 1.0 * half_area(box) + 1.5 * best_cost >= leaf_cost)) {
            if (n <= rtd::LEAF_MAX_COUNT) return make_leaf();
            // many coincident centroids: fall back to an index median
            const size_t mid = start + n / 2;
            const Box3 bl = build_sah(B, order, start, mid, ctx, depth + 1), br = build_sah(B, order, mid, end, ctx, depth + 1);
            (void)bl; (void)br;
            finish_box_node(me, box);
            return box;
        }
        const double ext = cmx[best_axis] - cmn[best_axis], scale = NB / ext;
        auto bin_of = [&](int idx) { const Box3& b = B.box[idx]; int k = (int)((0.5 * (b.mn[best_axis] + b.mx[best_axis]) - cmn[best_axis]) * scale); return k >= NB ? NB - 1 : (k < 0 ? 0 : k); };
        auto midit = std::stable_partition(order.begin() + start, order.begin() + end, [&](int idx) { return bin_of(idx) < best_split; });
        size_t mid = (size_t)(midit - order.begin());
        if (mid == start || mid == end) mid = start + n / 2;
        build_sah(B, order, start, mid, ctx, depth + 1);
        build_sah(B, order, mid, end, ctx, depth + 1);
        finish_box_node(me, box);
        return box;
    }

    Box3 range_box(const Build& B, const std::vector<int>& order, size_t s, size_t e) {
        Box3 b = B.box[order[s]];
        for (size_t i = s + 1; i < e; ++i) b = surrounding(b, B.box[order[i]]);
        return b;
    }

    // A sphere that shares its BVH node with another object (the two members of a span-2 node, bvh.rs:99-107; the members of an SAH leaf)
    // gets a box of its own in a scene small enough to be walked from LDS: the union box of two spheres is mostly empty, a sphere test
    // costs four box tests, and a stop at a leaf costs more than either. Book-1: 6.0 -> 2.1 sphere tests and 41.7 -> 47.3 box tests per
    // segment, k_extend 56.0 -> 52.1 ms, same frame bit for bit. Boxes and rects are their own bounding boxes already (measured: boxing them
    // too costs the book-2 final scene 4 %); in a scene walked from HBM a box record is a load like the sphere record it would spare.
    void emit_member(const Build& B, int k, const Chain& ctx, int depth) {
        const RtHittable* m = H(B.obj[k]);
        if (box_pair_members && m && (m->kind == RT_HIT_SPHERE || m->kind == RT_HIT_MOVING_SPHERE)) emit_boxed({B.obj[k]}, B.box[k], ctx, depth);
        else emit(B.obj[k], ctx, depth + 1);
    }

    // BVHNode::construct (bvh.rs:77-130) on the sub-range [start, end)
    Box3 build_range(Build& B, std::vector<int>& order, size_t start, size_t end, const Chain& ctx, int depth) {
        B.axis_state += GAMMA;
        const int axis = (int)((uint32_t)(fin(B.axis_state) >> 32) % 3u);   // bvh.rs:87
        const size_t span = end - start;
        const uint32_t me = (uint32_t)out.nodes.size();
        out.nodes.push_back(rtd::Node{});   // box node, filled below
        out.n_box_nodes++;
        Box3 box;
        if (span == 1) {
            emit(B.obj[order[start]], ctx, depth + 1);
            box = B.box[order[start]];
        } else if (span == 2) {
            const int a = order[start], b = order[start + 1];
            const bool less = B.key[axis][a] < B.key[axis][b];   // box_compare: strictly less, else "greater" (bvh.rs:24-31)
            const int l = less ? a : b, r = less ? b : a;
            emit_member(B, l, ctx, depth);
            emit_member(B, r, ctx, depth);
            box = surrounding(B.box[l], B.box[r]);
        } else {
            const std::vector<double>& key = B.key[axis];
            std::stable_sort(order.begin() + start, order.begin() + end, [&key](int x, int y) { return key[x] < key[y]; });   // sort_by is stable
            const size_t mid = start + span / 2;
            const Box3 bl = build_range(B, order, start, mid, ctx, depth + 1);
            const Box3 br = build_range(B, order, mid, end, ctx, depth + 1);
            box = surrounding(bl, br);
        }
        // boxes live in the space the children are traversed in; children under an instance
        // transform are traversed in local space, where the reference's boxes are defined too
        finish_box_node(me, box);
        return box;
    }

    // ---- post passes: merge adjacent leaf runs; fold a box node and its single leaf run into one ----
    static bool nobox(const rtd::Node& n) { return std::isinf(n.mn[0]) && n.mn[0] < 0; }
    static uint32_t ltype(const rtd::Node& n) { return n.leaf >> 28; }
    static uint32_t lcount(const rtd::Node& n) { return (n.leaf >> 24) & 15u; }
    static uint32_t lfirst(const rtd::Node& n) { return n.leaf & rtd::LEAF_MAX_FIRST; }
    static bool is_prim_run(const rtd::Node& n) { const uint32_t t = ltype(n); return n.leaf != 0 && ((t >= rtd::LT_SPHERE && t <= rtd::LT_TRI) || t == rtd::LT_BOX); }

    void remap(std::vector<rtd::Node>& nodes, const std::vector<uint32_t>& map, uint32_t new_n) {
        for (auto& n : nodes) n.skip = n.skip >= map.size() ? new_n : map[n.skip];
    }

    void merge_runs() {
        std::vector<rtd::Node>& src = out.nodes;
        const size_t n = src.size();
        std::vector<char> target(n + 1, 0);
        for (const auto& nd : src) if (!nobox(nd) && nd.skip <= n) target[nd.skip] = 1;
        std::vector<rtd::Node> dst; dst.reserve(n);
        std::vector<uint32_t> map(n + 1);
        for (size_t i = 0; i < n; ++i) {
            const rtd::Node& nd = src[i];
            if (!dst.empty() && !target[i] && nobox(nd) && is_prim_run(nd)) {
                rtd::Node& pv = dst.back();
                if (nobox(pv) && is_prim_run(pv) && ltype(pv) == ltype(nd) && lfirst(pv) + lcount(pv) == lfirst(nd) &&
                    lcount(pv) + lcount(nd) <= rtd::LEAF_MAX_COUNT) {
                    pv.leaf = rtd::make_leaf(ltype(pv), lfirst(pv), lcount(pv) + lcount(nd));
                    map[i] = (uint32_t)dst.size() - 1;
                    continue;
                }
            }
            map[i] = (uint32_t)dst.size();
            dst.push_back(nd);
        }
        map[n] = (uint32_t)dst.size();
        remap(dst, map, (uint32_t)dst.size());
        for (size_t i = 0; i < dst.size(); ++i) if (nobox(dst[i])) dst[i].skip = (uint32_t)i + 1;
        src.swap(dst);
    }

    void fold_box_leaf() {
        std::vector<rtd::Node>& src = out.nodes;
        const size_t n = src.size();
        std::vector<rtd::Node> dst; dst.reserve(n);
        std::vector<uint32_t> map(n + 1);
        for (size_t i = 0; i < n; ++i) {
            map[i] = (uint32_t)dst.size();
            const rtd::Node& nd = src[i];
            if (!nobox(nd) && nd.leaf == 0 && i + 1 < n && nd.skip == i + 2 && nobox(src[i + 1]) && is_prim_run(src[i + 1])) {
                rtd::Node f = nd; f.leaf = src[i + 1].leaf;
                dst.push_back(f);
                map[i + 1] = (uint32_t)dst.size();   // nothing points at it; keep the map total
                ++i;
                continue;
            }
            dst.push_back(nd);
        }
        map[n] = (uint32_t)dst.size();
        remap(dst, map, (uint32_t)dst.size());
        for (size_t i = 0; i < dst.size(); ++i) if (nobox(dst[i])) dst[i].skip = (uint32_t)i + 1;
        src.swap(dst);
    }

    // ---- optional: a box node whose whole subtree is one contiguous run of primitives of one kind, `limit` of them at most, becomes a leaf ----
    // (fewer stops of the walk for more primitive tests; the closest hit is the same. RT_LEAF_COLLAPSE=n, default off.)
    void collapse_small_subtrees(uint32_t limit) {
        std::vector<rtd::Node>& src = out.nodes;
        const size_t n = src.size();
        struct Run { uint32_t type, first, count; bool ok; };
        std::vector<Run> run(n, Run{0, 0, 0, false});
        auto sub_end = [&](size_t i) { return std::min<size_t>(std::max<size_t>(src[i].skip, i + 1), n); };
        for (size_t i = n; i-- > 0;) {
            const rtd::Node& nd = src[i];
            if (nd.leaf != 0u) { if (is_prim_run(nd)) run[i] = Run{ltype(nd), lfirst(nd), lcount(nd), true}; continue; }
            Run r{0, 0, 0, true}; bool first = true;
            for (size_t c = i + 1; c < sub_end(i) && r.ok; c = sub_end(c)) {
                const Run& k = run[c];
                if (!k.ok) { r.ok = false; break; }
                if (first) { r = k; first = false; }
                else if (k.type == r.type && k.first == r.first + r.count) r.count += k.count;
                else r.ok = false;
            }
            run[i] = (first || !r.ok || nobox(nd)) ? Run{0, 0, 0, false} : r;
        }
        std::vector<rtd::Node> dst; dst.reserve(n);
        std::vector<uint32_t> map(n + 1);
        for (size_t i = 0; i < n;) {
            map[i] = (uint32_t)dst.size();
            const rtd::Node& nd = src[i];
            if (nd.leaf == 0u && run[i].ok && run[i].count <= std::min<uint32_t>(limit, rtd::LEAF_MAX_COUNT) && sub_end(i) > i + 1) {
                rtd::Node f = nd; f.leaf = rtd::make_leaf(run[i].type, run[i].first, run[i].count); f.skip = (uint32_t)sub_end(i);
                dst.push_back(f);
                for (size_t c = i + 1; c < sub_end(i); ++c) map[c] = (uint32_t)dst.size();
                i = sub_end(i);
                continue;
            }
            dst.push_back(nd);
            ++i;
        }
        map[n] = (uint32_t)dst.size();
        remap(dst, map, (uint32_t)dst.size());
        for (size_t i = 0; i < dst.size(); ++i) if (nobox(dst[i]) || dst[i].leaf != 0u) dst[i].skip = (uint32_t)i + 1;
        src.swap(dst);
    }

    // ---- materials, textures, lights ----
    void compile_materials() {
        for (uint64_t i = 0; i < d.n_textures; ++i) {
            const RtTexture& t = d.textures[i];
            rtd::Texture o{};
            o.kind = (uint32_t)t.kind; o.a = t.a; o.b = t.b; o.scale = (float)t.scale;
            o.color[0] = (float)t.color.x; o.color[1] = (float)t.color.y; o.color[2] = (float)t.color.z;
            if (t.kind < RT_TEX_SOLID || t.kind > RT_TEX_IMAGE) { fail("unknown texture kind"); return; }
            if (t.kind == RT_TEX_CHECKER && (t.a < 0 || t.b < 0 || (uint64_t)t.a >= d.n_textures || (uint64_t)t.b >= d.n_textures)) { fail("checker texture ids out of range"); return; }
            if (t.kind == RT_TEX_NOISE && (t.a < 0 || (uint64_t)t.a >= d.n_perlins)) { fail("noise texture perlin id out of range"); return; }
            if (t.kind == RT_TEX_IMAGE && (t.a >= 0 && (uint64_t)t.a >= d.n_images)) { fail("image id out of range"); return; }
            out.textures.push_back(o);
        }
        // checker-of-checker: the device resolves the chain in a bounded loop; a deeper chain or a cycle cannot be rendered
        for (uint64_t i = 0; i < d.n_textures; ++i) {
            if (d.textures[i].kind != RT_TEX_CHECKER) continue;
            std::vector<std::pair<int32_t, uint32_t>> stack{{(int32_t)i, 1u}};
            while (!stack.empty()) {
                const auto [id, depth] = stack.back(); stack.pop_back();
                const RtTexture& t = d.textures[id];
                if (t.kind != RT_TEX_CHECKER) continue;
                if (depth > rtd::MAX_CHECKER_NESTING) { fail("checker textures must be nested at most 8 deep (and without cycles)"); return; }
                stack.push_back({t.a, depth + 1}); stack.push_back({t.b, depth + 1});
            }
        }
        for (uint64_t i = 0; i < d.n_perlins; ++i) {
            rtd::PerlinTable pt{};
            for (int k = 0; k < 256; ++k) {
                pt.ranvec[k] = rtd::Float4{(float)d.perlins[i].ranvec[k][0], (float)d.perlins[i].ranvec[k][1], (float)d.perlins[i].ranvec[k][2], 0.f};
                pt.perm_x[k] = d.perlins[i].perm_x[k] & 255u; pt.perm_y[k] = d.perlins[i].perm_y[k] & 255u; pt.perm_z[k] = d.perlins[i].perm_z[k] & 255u;
            }
            out.perlins.push_back(pt);
        }
        for (uint64_t i = 0; i < d.n_images; ++i) {
            const RtImage& im = d.images[i];
            if (im.data && (im.width == 0 || im.height == 0)) { fail("image with data but a zero dimension"); return; }
            rtd::Image o{}; o.offset = out.image_bytes.size(); o.width = im.data ? im.width : 0; o.height = im.data ? im.height : 0;
            if (im.data) out.image_bytes.insert(out.image_bytes.end(), im.data, im.data + (size_t)im.width * im.height * 3);
            out.images.push_back(o);
        }
        for (uint64_t i = 0; i < d.n_materials; ++i) {
            const RtMaterial& m = d.materials[i];
            rtd::Float4 a{0, 0, 0, 0}; uint32_t tex = rtd::TEX_INLINE;
            switch (m.kind) {
            case RT_MAT_LAMBERTIAN: case RT_MAT_DIFFUSE_LIGHT: case RT_MAT_ISOTROPIC: {
                if (m.texture < 0 || (uint64_t)m.texture >= d.n_textures) { fail("material texture id out of range"); return; }
                const RtTexture& t = d.textures[m.texture];
                if (t.kind == RT_TEX_SOLID) a = rtd::Float4{(float)t.color.x, (float)t.color.y, (float)t.color.z, 0.f};
                else tex = (uint32_t)m.texture;
                break;
            }
            case RT_MAT_METAL: a = rtd::Float4{(float)m.albedo.x, (float)m.albedo.y, (float)m.albedo.z, (float)m.fuzz}; break;
            case RT_MAT_DIELECTRIC: a = rtd::Float4{1.f, 1.f, 1.f, (float)m.ir}; break;
            default: fail("unknown material kind"); return;
            }
            out.mat_a.push_back(a);
            out.mat_b.push_back(rtd::make_mat_b((uint32_t)m.kind, tex));
        }
    }

    void compile_lights() {
        out.has_lights = d.lights >= 0;
        if (!out.has_lights) return;
        const RtHittable* L = H(d.lights); if (!L) return;
        std::vector<int> ids;
        if (L->kind == RT_HIT_LIST) {
            for (int c = 0; c < L->n_children; ++c) {
                if ((uint64_t)(L->first_child + c) >= d.n_children) { fail("children out of range"); return; }
                ids.push_back(d.children[L->first_child + c]);
            }
        } else ids.push_back(d.lights);
        if (ids.empty()) { fail("empty lights list (the reference divides by its length, hittable_list.rs:74,82)"); return; }
        for (int id : ids) {
            const RtHittable* h = H(id); if (!h) return;
            rtd::Light l{};
            if (h->kind == RT_HIT_XZ_RECT) { l.kind = rtd::LK_XZRECT; for (int i = 0; i < 5; ++i) l.p[i] = (float)h->p[i]; }
            else if (h->kind == RT_HIT_SPHERE) { l.kind = rtd::LK_SPHERE; for (int i = 0; i < 4; ++i) l.p[i] = (float)h->p[i]; }
            else l.kind = rtd::LK_DEFAULT;   // trait defaults: pdf_value = 0, random = (1,0,0)  (hittable.rs:54-59)
            out.lights.push_back(l);
        }
    }
};

}  // namespace

int compile_scene(const RtSceneDesc& desc, const CompileOptions& opt, CompiledScene& out) {
    out = CompiledScene();
    if (desc.abi_version != RT_ABI_VERSION) { out.error = "RtSceneDesc.abi_version mismatch"; return RT_ERR_INVALID; }
    if (!desc.hittables || desc.n_hittables == 0) { out.error = "scene has no hittables"; return RT_ERR_INVALID; }
    if ((desc.n_children && !desc.children) || (desc.n_materials && !desc.materials) || (desc.n_textures && !desc.textures) ||
        (desc.n_perlins && !desc.perlins) || (desc.n_images && !desc.images)) { out.error = "null array with non-zero count"; return RT_ERR_INVALID; }
    Compiler c(desc, out);
    // Culling list members pays where the walks of a wave have already drifted apart (a BVH of some size in the scene): measured on the
    // book-2 final scene k_extend 124 -> 108 ms. In a scene that is only a list (the Cornell boxes) every lane of a wave stops at the
    // same leaves in the same order, a stop costs the wave the same with 64 lanes as with 20, and culling only breaks that step
    // (Cornell 23 -> 33 ms, Cornell smoke 52 -> 60 ms): there the members stay as the reference has them.
    uint64_t bvh_members = 0;
    for (uint64_t i = 0; i < desc.n_hittables; ++i) if (desc.hittables[i].kind == RT_HIT_BVH && desc.hittables[i].n_children > 0) bvh_members += (uint64_t)desc.hittables[i].n_children;
    c.cull_lists = opt.cull_lists < 0 ? bvh_members >= 32 : opt.cull_lists != 0;
    c.box_pair_members = opt.member_boxes < 0 ? desc.n_hittables < 8192 : opt.member_boxes != 0;      // scenes of that size are LDS-resident (rt_api.cpp: 144 KB of records and spheres)
    c.park_cost = std::max(0.0, opt.park_cost);
    c.big_spheres_first = opt.big_spheres_first;
    out.xforms.push_back(rtd::Xform{0.f, 1.f, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}});
    out.wraps.push_back(rtd::Wrap{});
    c.compile_materials();
    if (!c.ok()) return out.error.find("must be") != std::string::npos ? RT_ERR_UNSUPPORTED : RT_ERR_INVALID;
    Chain root;
    c.emit(desc.world, root);
    if (!c.ok()) return out.error.find("must be") != std::string::npos ? RT_ERR_UNSUPPORTED : RT_ERR_INVALID;
    if (out.xforms.size() > 255) { out.error = "more than 255 distinct instance transforms must be flattened by the caller"; return RT_ERR_UNSUPPORTED; }
    c.merge_runs();
    c.fold_box_leaf();
    if (opt.leaf_collapse > 1) c.collapse_small_subtrees(opt.leaf_collapse);
    c.compile_lights();
    if (!c.ok()) return RT_ERR_INVALID;
    out.background_mode = desc.background_mode;
    out.background[0] = (float)desc.background.x; out.background[1] = (float)desc.background.y; out.background[2] = (float)desc.background.z;
    return RT_OK;
}

}  // namespace rtc
