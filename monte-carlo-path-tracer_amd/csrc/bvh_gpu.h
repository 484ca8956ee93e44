// GPU construction of the binary BVH (bvh_gpu.hip).  Input: one fp32 box per triangle (lo xyz, hi xyz; already rounded outward from
// the fp64 vertices).  Output: the same arrays the host SAH builder hands to the rest of scene_build.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "scene_build.h"

struct GpuBvh {
    std::vector<f4h> nodes;        // 4 per inner node, host builder's layout (scene_build.cpp write_node); node 0 = root
    std::vector<uint32_t> order;   // leaf order: order[i] = input triangle at position i
    uint32_t depth = 0, max_leaf = 0;
    double ms = 0.0;               // wall time incl. upload / download
};

// Needs n > MCPT_LEAF_MAX and a current HIP device.  Returns false with `err` set on any HIP error.
bool gpu_build_bvh2(const float* tri_boxes, uint32_t n, GpuBvh& out, std::string& err);
// Same contract, SAH-costed: PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) over the Morton order -- every merge is
// the one that minimises the merged box's surface area within a +-16 window.  The default of MCPT_FLAG_GPU_BVH_BUILD.
bool gpu_build_ploc(const float* tri_boxes, uint32_t n, GpuBvh& out, std::string& err);

// The 8-wide collapse of build_bvh8 (scene_build.cpp) on the device: same dynamic programme, same octant slots, same numbering -- the records
// and the leaf order are the host's bit for bit.  In/out: `nodes2` (renumbered binary tree; its leaf codes follow the new leaf order on return)
// and `order` (leaf order); out: nodes8 (5 x 16 B per node) and the depth of the 8-wide tree.
bool gpu_collapse_bvh8(std::vector<f4h>& nodes2, std::vector<int>& order, std::vector<f4h>& nodes8, uint32_t& depth8, std::string& err);
