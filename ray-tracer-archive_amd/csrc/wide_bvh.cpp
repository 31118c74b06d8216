// wide_bvh.cpp — host builder of the 8-wide BVH (wide_bvh.hpp). Replaces, for scenes in HBM, the reference's binary BVHNode::hit
// recursion (bvh.rs:134-143) by a walk that fetches ONE 128-byte line per visit and tests its eight children side by side.
#include "wide_bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rtw {
namespace {

struct Box { float mn[3], mx[3]; };
inline double half_area(const Box& b) {
    const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}
struct Run { uint32_t type, first, count; bool ok; };

struct Builder {
    const std::vector<rtd::Node>& nodes;
    float margin;
    WideTree& out;
    std::vector<Run> run;
    size_t n;
    Builder(const std::vector<rtd::Node>& nd, float m, WideTree& o) : nodes(nd), margin(m), out(o), n(nd.size()) {}

    size_t sub_end(size_t i) const { return std::min<size_t>(std::max<size_t>(nodes[i].skip, i + 1), n); }
    static uint32_t ltype(uint32_t leaf) { return leaf >> 28; }
    static uint32_t lcount(uint32_t leaf) { return (leaf >> 24) & 15u; }
    static uint32_t lfirst(uint32_t leaf) { return leaf & rtd::LEAF_MAX_FIRST; }
    // a leaf record without a box of its own (the two unlike members of a span-2 node, bvh.rs:99-107) counts with its parent's box
    std::vector<Box> eff;
    void effective_boxes() {
        eff.resize(n);
        std::vector<std::pair<size_t, Box>> open;        // enclosing boxed subtrees: (end, box)
        for (size_t i = 0; i < n; ++i) {
            while (!open.empty() && open.back().first <= i) open.pop_back();
            Box b;
            if (std::isfinite(nodes[i].mn[0])) { for (int a = 0; a < 3; ++a) { b.mn[a] = nodes[i].mn[a]; b.mx[a] = nodes[i].mx[a]; } }
            else b = open.back().second;                 // eligible() guarantees there is one
            eff[i] = b;
            if (std::isfinite(nodes[i].mn[0]) && sub_end(i) > i + 1) open.push_back({sub_end(i), b});
        }
    }
    Box box_of(size_t i) const { return eff[i]; }

    // primitives of the subtree at i as one contiguous run of one kind, if they are one
    void find_runs() {
        run.assign(n, Run{0, 0, 0, false});
        for (size_t i = n; i-- > 0;) {
            const rtd::Node& nd = nodes[i];
            Run r{0, 0, 0, true}; bool first = true;
            if (nd.leaf != 0u) { r = Run{ltype(nd.leaf), lfirst(nd.leaf), lcount(nd.leaf), true}; first = false; }
            for (size_t c = i + 1; c < sub_end(i) && r.ok; c = sub_end(c)) {
                const Run& k = run[c];
                if (!k.ok) { r.ok = false; break; }
                if (first) { r = k; first = false; }
                else if (k.type == r.type && k.first == r.first + r.count) r.count += k.count;
                else r.ok = false;
            }
            run[i] = (first || !r.ok) ? Run{0, 0, 0, false} : r;
        }
    }
    bool is_leaf_entry(size_t i) const { return run[i].ok && run[i].count <= 8u; }

    struct Entry { Box box; bool leaf; uint32_t word; size_t bin; };   // leaf: word = payload; else bin = binary node to open into a wide node

    uint32_t alloc() { const uint32_t k = out.n_nodes++; out.words.resize((size_t)out.n_nodes * 32u, 0u); return k; }

    // children of a wide node rooted at binary node i: open the largest inner child until eight are held
    void gather(size_t i, std::vector<Entry>& es) {
        std::vector<size_t> kids;
        auto kids_of = [&](size_t p, std::vector<size_t>& into) {
            // a record with a payload AND a subtree cannot come out of the scene compiler (fold_box_leaf folds only childless runs)
            for (size_t c = p + 1; c < sub_end(p); c = sub_end(c)) into.push_back(c);
        };
        kids_of(i, kids);
        // first the children that cannot be a leaf entry (more than 8 primitives, or of two kinds), largest surface first; then, while slots
        // are still free, leaf entries of more than kSplitAbove members: the box tests of a node's eight slots cost one visit however many
        // are used, and two boxes around 3 + 4 primitives cull better than one around 7 — but a leaf visit keeps as many lanes busy as it has
        // members, so small entries stay whole
        constexpr uint32_t kSplitAbove = 4;
        for (int pass = 0; pass < 2;) {
            if (kids.size() >= 8) break;
            int best = -1; double area = -1.0;
            for (size_t k = 0; k < kids.size(); ++k) {
                const size_t c = kids[k];
                if (nodes[c].leaf != 0u) continue;
                const bool leafable = is_leaf_entry(c);
                if (pass == 0 ? leafable : !(leafable && run[c].count > kSplitAbove)) continue;
                std::vector<size_t> g; kids_of(c, g);
                if (g.empty() || kids.size() - 1 + g.size() > 8) continue;
                const double a = half_area(box_of(c));
                if (a > area) { area = a; best = (int)k; }
            }
            if (best < 0) { ++pass; continue; }
            std::vector<size_t> g; kids_of(kids[(size_t)best], g);
            kids.erase(kids.begin() + best);
            kids.insert(kids.end(), g.begin(), g.end());
        }
        for (size_t c : kids) {
            Entry e; e.box = box_of(c); e.bin = c; e.leaf = false; e.word = 0;
            if (is_leaf_entry(c)) { e.leaf = true; e.word = 0x80000000u | rtd::make_leaf(run[c].type, run[c].first, run[c].count); }
            es.push_back(e);
        }
    }

    // a binary leaf (or a one-kind run) of more than 8 primitives: a wide node of its own whose entries are the run's pieces
    uint32_t make_run_node(const Box& b, const Run& r, int depth) {
        const uint32_t w = alloc();
        std::vector<Entry> es;
        for (uint32_t at = 0; at < r.count; at += 8u) {
            Entry e; e.box = b; e.leaf = true; e.bin = 0; e.word = 0x80000000u | rtd::make_leaf(r.type, r.first + at, std::min(8u, r.count - at));
            es.push_back(e);
        }
        encode(w, es);
        out.depth = std::max(out.depth, (uint32_t)depth);
        return w;
    }

    uint32_t make_wide(size_t i, int depth) {
        const uint32_t w = alloc();
        std::vector<Entry> es;
        if (nodes[i].leaf != 0u && sub_end(i) == i + 1) {                 // the root itself is a leaf (a scene of one run)
            Entry e; e.box = box_of(i); e.leaf = true; e.bin = i; e.word = 0;
            es.push_back(e);
        } else gather(i, es);
        for (Entry& e : es) {
            if (e.leaf && e.word != 0u) continue;
            const size_t c = e.bin;
            if (nodes[c].leaf != 0u || (run[c].ok && run[c].count > 8u && sub_end(c) == c + 1)) {
                // a binary leaf of up to 15 primitives: one entry when it fits the eight lanes, else a node of its own
                const Run r{ltype(nodes[c].leaf), lfirst(nodes[c].leaf), lcount(nodes[c].leaf), true};
                if (r.count <= 8u) { e.leaf = true; e.word = 0x80000000u | nodes[c].leaf; }
                else { e.leaf = false; e.word = make_run_node(e.box, r, depth + 1); }
            } else e.word = make_wide(c, depth + 1);
        }
        encode(w, es);
        out.depth = std::max(out.depth, (uint32_t)depth);
        return w;
    }

    static float dq(uint32_t q, float s, float o) { return std::fmaf((float)q, s, o); }   // the device's decode, to the bit

    void encode(uint32_t w, const std::vector<Entry>& es) {
        uint32_t* W = out.words.data() + (size_t)w * 32u;
        std::memset(W, 0, 128);
        if (es.empty()) return;
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
        for (const Entry& e : es) for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], std::nextafterf(e.box.mn[a] - margin, -std::numeric_limits<float>::infinity()));
            mx[a] = std::max(mx[a], std::nextafterf(e.box.mx[a] + margin, std::numeric_limits<float>::infinity()));
        }
        float org[3], scl[3]; uint32_t ebyte[3];
        for (int a = 0; a < 3; ++a) {
            // grid: step 2^e, origin two steps below the lowest plane (one for the outward step of the lowest child, one for the float
            // rounding of the origin itself); coarser until every child's planes decode (in the device's float arithmetic) to a box that
            // contains the child's own, margin included, inside 0..255 — which the first candidate does unless coordinates dwarf the node
            int e = (int)std::ceil(std::log2(std::max((double)mx[a] - (double)mn[a], 1e-30) / 250.0));
            e = std::max(-100, std::min(100, e));
            for (;; ++e) {
                const float s = std::ldexp(1.0f, e);
                const double od = (double)mn[a] - 2.0 * (double)s;
                float o = (float)od; if ((double)o > od) o = std::nextafterf(o, -std::numeric_limits<float>::infinity());
                bool ok = true;
                for (const Entry& en : es) {
                    const double lo = (double)en.box.mn[a] - margin, hi = (double)en.box.mx[a] + margin;
                    const double ql = std::floor((lo - (double)o) / (double)s) - 1.0, qh = std::ceil((hi - (double)o) / (double)s) + 1.0;
                    if (ql < 0.0 || qh > 255.0) { ok = false; break; }
                    if (!((double)dq((uint32_t)ql, s, o) <= lo && (double)dq((uint32_t)qh, s, o) >= hi)) { ok = false; break; }
                }
                if (ok || e >= 120) { org[a] = o; scl[a] = s; ebyte[a] = (uint32_t)(e + 127); break; }
            }
        }
        for (size_t k = 0; k < es.size() && k < 8; ++k) {
            const Entry& en = es[k];
            uint32_t q[6];
            for (int a = 0; a < 3; ++a) {
                const double lo = (double)en.box.mn[a] - margin, hi = (double)en.box.mx[a] + margin;
                q[a] = (uint32_t)std::max(0.0, std::min(255.0, std::floor((lo - (double)org[a]) / (double)scl[a]) - 1.0));
                q[3 + a] = (uint32_t)std::max(0.0, std::min(255.0, std::ceil((hi - (double)org[a]) / (double)scl[a]) + 1.0));
            }
            uint32_t* c = W + 4 * k;
            c[0] = en.leaf ? en.word : (en.word + 1u);             // inner: wide node index + 1 (0 = empty slot); leaf: bit 31 | payload
            c[1] = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
            c[2] = q[4] | (q[5] << 8);
            if (en.leaf) { out.n_leaf_entries++; out.prims_in_leaves += lcount(en.word); } else out.n_inner_entries++;
        }
        // the node's grid rides in the spare word of chunks 0..3
        std::memcpy(&W[4 * 0 + 3], &org[0], 4); std::memcpy(&W[4 * 1 + 3], &org[1], 4); std::memcpy(&W[4 * 2 + 3], &org[2], 4);
        W[4 * 3 + 3] = ebyte[0] | (ebyte[1] << 8) | (ebyte[2] << 16);
    }
};

}  // namespace

bool eligible(const std::vector<rtd::Node>& nodes) {
    if (nodes.empty()) return false;
    for (size_t i = 0; i < nodes.size(); ++i) {
        const rtd::Node& nd = nodes[i];
        bool boxed = true;
        for (int a = 0; a < 3; ++a) if (!std::isfinite(nd.mn[a]) || !std::isfinite(nd.mx[a])) boxed = false;
        if (!boxed && (i == 0 || nd.leaf == 0u)) return false;     // only a LEAF may lack a box (it counts with its parent's), never the root
        if (nd.leaf != 0u) {
            const uint32_t t = nd.leaf >> 28;
            if (!(t == rtd::LT_SPHERE || t == rtd::LT_RECT || t == rtd::LT_TRI || t == rtd::LT_BOX)) return false;
        }
    }
    // one root: the first record's subtree is the whole array
    return std::min<size_t>(std::max<size_t>(nodes[0].skip, 1), nodes.size()) == nodes.size();
}

bool build(const std::vector<rtd::Node>& nodes, float margin, WideTree& out, std::string& err) {
    out = WideTree();
    if (!eligible(nodes)) { err = "not a static BVH"; return false; }
    Builder b(nodes, margin, out);
    b.effective_boxes();
    b.find_runs();
    if (b.run[0].ok && b.run[0].count > 8u && nodes[0].leaf != 0u) b.make_run_node(b.box_of(0), b.run[0], 0);
    else b.make_wide(0, 0);
    if ((uint64_t)out.n_nodes * 128ull >= 0x7FFFFFF0ull) { err = "wide node array beyond 2 GB"; return false; }
    return true;
}

}  // namespace rtw
