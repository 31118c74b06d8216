// wide_bvh.hpp — the 8-wide BVH of scenes walked from HBM (kernels.hip k_extend_wide): built on the host from the threaded binary tree.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "device_types.h"

namespace rtw {

// One 128-byte node = 8 chunks of 16 bytes, chunk j describing child j (device_types.h Wide8 has the bit layout).
struct WideTree {
    std::vector<uint32_t> words;          // 32 per node
    uint32_t n_nodes = 0, n_leaf_entries = 0, n_inner_entries = 0, depth = 0;
    uint64_t prims_in_leaves = 0;
};

// True when every record of `nodes` carries a finite box and every leaf holds spheres, rects, triangles or boxes: a static BVH, the shape
// k_extend_wide walks. (Lists, wrappers, moving spheres and media keep the binary walk.)
bool eligible(const std::vector<rtd::Node>& nodes);

// Collapses the binary tree (any builder's: the reference-shaped one or SAH) into 8-wide nodes: the child with the largest surface is opened
// until eight are held; a subtree of at most 8 primitives of one kind in one contiguous run becomes ONE leaf entry (the lanes of a group
// test its members side by side). Boxes are quantised to 8 bits per plane on a per-node power-of-two grid, outwards, with `margin` (the
// slab test's own rounding, in space units) added first; the float arithmetic of the device's decode is replayed to check containment.
bool build(const std::vector<rtd::Node>& nodes, float margin, WideTree& out, std::string& err);

}  // namespace rtw
