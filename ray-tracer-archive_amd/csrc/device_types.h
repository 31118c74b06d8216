// device_types.h — layout of a compiled scene in HBM (and, when it fits, in LDS).
// Shared by the host-side scene compiler (scene_compile.cpp) and the HIP kernels (kernels.hip).
#pragma once
#include <stdint.h>

namespace rtd {

// ---- threaded BVH -------------------------------------------------------------------------------
// One 32-byte record per node, laid out in depth-first pre-order = the order BVHNode::hit
// (bvh.rs:134-143) and HittableList::hit (hittable_list.rs:39-51) visit things: left before right,
// list members in insertion order. Traversal needs no stack:
//     box test passes (or the node has no box)  -> run the node's leaf payload, go to i+1
//     box test fails                            -> go to `skip` (first node after this subtree)
// A node is two float4: (mn.xyz, skip) and (mx.xyz, leaf).
struct Node {
    float mn[3];
    uint32_t skip;
    float mx[3];
    uint32_t leaf;   // 0: inner node. else (type << 28) | (count << 24) | first
};
static_assert(sizeof(Node) == 32, "node record is 32 bytes");

enum LeafType : uint32_t {
    LT_NONE = 0,
    LT_SPHERE = 1,   // first/count index spheres[]
    LT_MOVING = 2,   // moving[]
    LT_RECT = 3,     // rects[]
    LT_TRI = 4,      // tris[]
    LT_MEDIUM = 5,   // media[]
    LT_XFORM = 6,    // Translate/RotateY::hit: first & 0xFFFF = xform id the ray moves to (0 = the world ray); XFORM_EXIT set on the record that
                     // closes a wrapper (restores the enclosing space), clear on the one that opens it
    LT_BOX = 7       // boxes[]: a Box (boxes.rs:11-75) as ONE 32-byte record; its six sides stay in rects[] for shading, the hit id is the side's rect id
};
constexpr uint32_t XFORM_EXIT = 1u << 23;
constexpr uint32_t LEAF_MAX_COUNT = 15;
constexpr uint32_t MAX_PROLOGUE = 4;   // leaf payloads tested at the start of every walk instead of being met by it
constexpr uint32_t LEAF_MAX_FIRST = (1u << 24) - 1;
inline uint32_t make_leaf(uint32_t type, uint32_t first, uint32_t count) { return (type << 28) | (count << 24) | first; }

// A node without a box (list members, wrappers) stores mn.x = -inf: the slab test passes and the
// node is not counted as an Aabb::hit.
//
// The DEVICE copy of the array (rt_api.cpp: device_nodes, which documents the address space) holds the same records regrouped for
// the kernel, boxes as centre c and half extent h: a = (c.x, c.y, h.x, h.y), b = (c.z, h.z, skip, hit). A ray meets an axis' slab at
// tc -+ th (tc = c/d - o/d, th = h/|d|): no min/max to order the planes, and one visit is two packed FMAs, a packed multiply and two
// packed adds. h is rounded up and padded so that [c-h, c+h] contains the (padded) host box; a record without a box has c = 0,
// h = inf. `skip` and `hit` are the byte addresses of the record the walk visits next when the box is missed / passed: a lane's whole
// state is its address. `hit` of a record with a leaf payload is that leaf's PARK TWIN (a record whose box cannot be passed and whose
// links lead to itself; its first two words are the address to resume at and the leaf payload); DONE and IDLE are such records too.
struct NodeDev { float cx, cy, hx, hy, cz, hz; uint32_t skip_bytes, leaf /* = hit link */; };
static_assert(sizeof(NodeDev) == 32, "device node record is 32 bytes");
// Compressed record for scenes that do not fit LDS (kernels.hip M_C16; rt_api.cpp device_nodes16): the box corners as u16 on a grid
// over the scene's bounds (lo rounded down and hi rounded up by one more step than needed, so the decoded box contains the host box
// plus the decode's rounding), one 16-byte load per visit. link: bit 31 set = a leaf, the low bits its payload (type | count |
// first; type < 8), its successor is the next record whether the box is passed or not; bit 31 clear = an inner record, link = byte
// offset of the record to visit when the box is MISSED (passed: the next record). lo.x > hi.x marks a record without a box.
struct Node16 { uint16_t lo[3], hi[3]; uint32_t link; };
static_assert(sizeof(Node16) == 16, "compressed node record is 16 bytes");
// 8-wide node of the walk from HBM (kernels.hip k_extend_wide; wide_bvh.cpp builds it): 128 bytes = one cache line = 8 chunks of 16 bytes,
// chunk j for child j, fetched by the 8 lanes that share a ray with ONE load instruction:
//   word 0  child: 0 = empty slot; bit 31 clear = index of the child's wide node + 1; bit 31 set = leaf payload (type | count <= 8 | first)
//   word 1  qlo.x | qlo.y << 8 | qlo.z << 16 | qhi.x << 24     the child's box on the node's grid, 8 bits a plane, rounded outwards
//   word 2  qhi.y | qhi.z << 8
//   word 3  chunk 0..2: the grid's origin x, y, z (f32); chunk 3: the biased exponents of its power-of-two steps, ex | ey << 8 | ez << 16
// plane = fma((float)q, 2^(e - 127), origin): the host replays this arithmetic when it picks the grid, so containment holds to the bit.
constexpr uint32_t WIDE_NODE_BYTES = 128;
constexpr uint32_t WIDE_MAX_DEPTH = 16;     // levels of the wide tree the walk's stack is sized for (one whole-node entry per level at worst); a deeper tree keeps the binary walk

// payload words of the two shared self-loop records (leaf type 0 = no primitive work):
constexpr uint32_t LEAF_IDLE = 1u << 24;   // the lane holds no ray
constexpr uint32_t LEAF_DONE = 2u << 24;   // the lane's ray has visited every node

// ---- primitives: one geometry array per type (16-byte records) + one u32 `meta` per primitive ----
// meta = material id (22 bits) | wrap id << 22 (10 bits)
// A "wrap" is the chain of Translate / RotateY / FlipFace objects above the primitive (hittable.rs:62-205).
// Each of them post-processes the HitRecord on the way out: Translate and RotateY call set_face_normal
// again (hittable.rs:82-83, 173) — RotateY with the ray of its CHILD space against the normal it has
// just rotated back to its PARENT space, which can flip the normal — and FlipFace negates front_face
// (hittable.rs:199). k_shade replays that sequence op by op; wrap 0 = no wrappers.
constexpr uint32_t META_MAT_MASK = (1u << 22) - 1;
constexpr uint32_t MAX_WRAPS = 1u << 10;
inline uint32_t make_meta(uint32_t mat, uint32_t wrap) { return (mat & META_MAT_MASK) | (wrap << 22); }
enum WrapOp : uint32_t { WO_TRANSLATE = 0, WO_ROTATE_Y = 1, WO_FLIP_FACE = 2 };
constexpr uint32_t MAX_WRAP_OPS = 6;
struct Wrap {
    uint32_t xform;      // composite transform of the chain (0 = identity)
    uint32_t n_ops;      // ops[0] is the outermost wrapper
    uint32_t _pad[2];
    struct { uint32_t kind; float sin_t, cos_t; uint32_t _p; } op[MAX_WRAP_OPS];
};

struct Float4 { float x, y, z, w; };
// sphere   : 1 x Float4  (center.xyz, radius)                                   sphere.rs:11-15
// moving   : 3 x Float4  (center0.xyz, radius) (center1.xyz, time0) (time1,0,0,0)  moving_sphere.rs:8-15
// rect     : 2 x Float4  (a0, a1, b0, b1) (k, kaxis as float 0/1/2, 0, 0)        aarect.rs:10-17 (kaxis 2=Xy,1=Xz,0=Yz)
// triangle : 3 x Float4  (v0,0) (v1,0) (v2,0)
// box      : 2 x Float4  (x0, x1, y0, y1) (z0, z1, first side's rect index as u32 bits, 0): the six sides in boxes.rs order are rects
//            first .. first+5 (z1 z0 y1 y0 x1 x0); k_extend tests them from this record in straight-line code (kernels.hip box_sides_hit)

// hit record prim id: (leaf type << 28) | index into that type's array; 0 = miss
constexpr uint32_t HIT_NONE = 0;

// ---- transforms ----------------------------------------------------------------------------------
// A chain of Translate / RotateY wrappers composed into: local = RotY(-theta)(world - offset).
struct Xform { float sin_t, cos_t, off[3]; float _pad[3]; };   // 32 B; id 0 = identity

// ---- media (constant_medium.rs) --------------------------------------------------------------------
struct Medium {
    uint32_t boundary_type;   // LT_SPHERE or LT_RECT (a Box: six consecutive rects)
    uint32_t boundary_first;
    uint32_t boundary_count;
    uint32_t boundary_xform;  // xform applied to the ray before the boundary test (0 = none)
    float neg_inv_density;
    uint32_t meta;            // phase-function material etc.
    uint32_t medium_id;       // key of the free-path draw (= hittable id in the scene graph)
    uint32_t _pad;
};

// ---- materials / textures ----------------------------------------------------------------------------
enum MatKind : uint32_t { MK_LAMBERTIAN = 0, MK_METAL = 1, MK_DIELECTRIC = 2, MK_DIFFUSE_LIGHT = 3, MK_ISOTROPIC = 4 };
constexpr uint32_t TEX_INLINE = 0xFFFFFu;   // solid colour folded into the material record
// mat_a[i] = (colour.rgb, param) with colour = solid albedo / emit colour, param = fuzz or ir
// mat_b[i] = kind | (texture id << 4)         texture id TEX_INLINE: use mat_a colour
inline uint32_t make_mat_b(uint32_t kind, uint32_t tex) { return kind | (tex << 4); }

enum TexKind : uint32_t { TK_SOLID = 0, TK_CHECKER = 1, TK_NOISE = 2, TK_IMAGE = 3 };
constexpr uint32_t MAX_CHECKER_NESTING = 8;   // CheckerTexture inside CheckerTexture (texture.rs:60-69 recurses); deeper graphs and cycles are rejected at compile
struct Texture { uint32_t kind; int32_t a, b; float scale; float color[3]; uint32_t _pad; };   // 32 B
struct PerlinTable { Float4 ranvec[256]; uint32_t perm_x[256], perm_y[256], perm_z[256]; };
struct Image { uint64_t offset; uint32_t width, height; };   // offset into the image byte pool

// ---- lights (main.rs:669-686; only XzRect and Sphere implement pdf_value/random) ---------------------
enum LightKind : uint32_t { LK_DEFAULT = 0, LK_XZRECT = 1, LK_SPHERE = 2 };
struct Light { uint32_t kind; float p[5]; uint32_t _pad[2]; };   // rect: a0,a1,b0,b1,k   sphere: c.xyz, r

}  // namespace rtd
