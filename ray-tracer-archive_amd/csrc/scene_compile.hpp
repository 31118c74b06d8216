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
    std::vector<rtd::Medium> media;
    std::vector<rtd::Xform> xforms;                                          // [0] = identity
    std::vector<rtd::Wrap> wraps;                                            // [0] = no wrappers
    std::vector<rtd::Float4> mat_a;    std::vector<uint32_t> mat_b;
    std::vector<rtd::Texture> textures;
    std::vector<rtd::PerlinTable> perlins;
    std::vector<rtd::Image> images;    std::vector<uint8_t> image_bytes;
    std::vector<rtd::Light> lights;
    std::vector<uint32_t> prologue;                                          // leaf payloads of the root list's every-ray members (scene_compile.cpp: prologue_member)
    int background_mode = 0; float background[3] = {0, 0, 0};
    bool has_lights = false;
    // statistics
    uint64_t n_box_nodes = 0;
    std::string error;
};

// Returns RT_OK or a negative RtStatus; `out.error` explains.
int compile_scene(const RtSceneDesc& desc, CompiledScene& out);

}  // namespace rtc
