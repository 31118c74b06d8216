// scene_compile.hpp — RtSceneDesc (the reference's object graph) -> device layout (device_types.h).
#pragma once
#include "../../include/rt_hip.h"
#include "device_types.h"

#include <string>
#include <vector>

namespace rtc {

struct CompiledScene {
    std::vector<rtd::Node> nodes;
    std::vector<rtd::Float4> spheres;  std::vector<uint32_t> sphere_meta;
    std::vector<rtd::Float4> moving;   std::vector<uint32_t> moving_meta;    // 3 per primitive
    std::vector<rtd::Float4> rects;    std::vector<uint32_t> rect_meta;      // 2 per primitive
    std::vector<rtd::Float4> tris;     std::vector<uint32_t> tri_meta;       // 3 per primitive
    std::vector<rtd::Float4> boxes;                                          // 2 per Box (device_types.h): bounds + index of its first side in rects
    std::vector<rtd::Medium> media;
    std::vector<rtd::Xform> xforms;                                          // [0] = identity
    std::vector<rtd::Wrap> wraps;                                            // [0] = no wrappers
    std::vector<rtd::Float4> mat_a;    std::vector<uint32_t> mat_b;
    std::vector<rtd::Texture> textures;
    std::vector<rtd::PerlinTable> perlins;
    std::vector<rtd::Image> images;    std::vector<uint8_t> image_bytes;
    std::vector<rtd::Light> lights;
    uint32_t first_leaf = 0;                                                 // leaf word of the spheres a walk tests before it enters the tree (emit_bvh: spheres as large as the scene), 0: none
    std::vector<uint32_t> prologue;                                          // leaf payloads of the root list's every-ray members (scene_compile.cpp: prologue_member)
    int background_mode = 0; float background[3] = {0, 0, 0};
    bool has_lights = false;
    // statistics
    uint64_t n_box_nodes = 0;
    std::string error;
};

// Layout choices of the compiler (include/rt_hip.h RtUploadOptions); none changes a hit.
struct CompileOptions {
    int cull_lists = -1;        // HittableList members behind culling boxes: -1 = when the scene holds a BVH of >= 32 members, 0 = never (the reference's walk), 1 = always
    int member_boxes = -1;      // a sphere of a span-2 BVH node / SAH leaf gets a box of its own: -1 = in LDS-sized scenes, 0 = never (bvh.rs:99-107), 1 = always
    double park_cost = 6.0;     // a stop of the walk at a leaf, in primitive tests (grouping of culled list members)
    uint32_t leaf_collapse = 0; // a box node whose subtree is <= n primitives of one kind becomes a leaf (0/1: off)
    bool big_spheres_first = true;  // a sphere of the root BVH whose box is most of the scene (a ground sphere) is tested when a walk begins, not met by it
};

// Returns RT_OK or a negative RtStatus; `out.error` explains.
int compile_scene(const RtSceneDesc& desc, const CompileOptions& opt, CompiledScene& out);
inline int compile_scene(const RtSceneDesc& desc, CompiledScene& out) { return compile_scene(desc, CompileOptions(), out); }

}  // namespace rtc
