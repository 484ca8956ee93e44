// Host-side scene preparation: the MI355X replacement for Render::tranform_triangle (Render.cpp:12-44) and
// BVH::BVH / BVH::build (BVH.cpp:6-54).  Output = the flat arrays of device_scene.h, ready for one hipMemcpy each.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include "../../include/mcpt.h"
#include "device_scene.h"

struct f4h { float x, y, z, w; };   // host mirror of float4 (16 B)

struct HostScene {
    std::vector<f4h> nodes;          // 4 per inner node (binary tree: megakernel + probes)
    std::vector<f4h> nodes8;         // 5 per node of the 8-wide compressed tree (wavefront trace kernel)
    std::vector<f4h> tri_isect;      // 3 per triangle (leaf order)
    std::vector<f4h> tri_shade;      // MCPT_TRI_SHADE_F4 = 8 per triangle: shading record (4) + fp64 plane (2) + spare (2), device_scene.h
    std::vector<double> tri_pos64;   // 9 per triangle
    std::vector<int32_t> tri_face;   // leaf order -> input face index
    std::vector<DevMaterial> mats;
    std::vector<DevLight> lights;
    std::vector<double> light_pos64; // 9 per light
    std::vector<f4h> texels;
    DevCamera cam;
    double centre[3] = {0.0, 0.0, 0.0};   // the point every coordinate above is relative to (device_scene.h: DevScene::centre)
    uint32_t bvh_depth = 0, max_leaf = 0, bvh8_depth = 0;
    std::vector<int> subtree_begin;  // binary nodes: first index of every depth-first-numbered subtree below the breadth-first top levels (ascending)
    bool reference_tie_order = false; // in: MCPT_FLAG_REFERENCE_TIE_ORDER -- the tie rank of a triangle (low 28 bits of tri_isect[3 i].w) is its position in the
                                     //     reference's BVH::triangles after BVH::build instead of its position in this library's leaf order
    bool allow_deep_binary = false;  // in: the caller never traverses `nodes` (wavefront pipeline only) -> a device tree deeper than MCPT_STACK_DEPTH is fine
    bool binary_ok = true;           // out: `nodes` fits the binary-tree kernels' stack
    double bvh_build_ms = 0.0;
};

// Optional replacement for the host SAH builder (bvh_gpu.hip): gets one fp32 box per face (lo xyz, hi xyz, rounded outward) and
// fills the binary tree in the host builder's node layout, the leaf order, the depth in inner levels and the largest leaf.
using BvhBuildFn = std::function<bool(const float* boxes, uint32_t n, std::vector<f4h>& nodes, std::vector<int>& order, uint32_t& depth,
                                      uint32_t& max_leaf, std::string& err)>;

// Optional replacement for the host's 8-wide collapse + quantisation (bvh_gpu.hip: gpu_collapse_bvh8): binary nodes (renumbered, root = 0) in, nodes8 + depth
// out; rewrites the binary tree's leaf codes and the leaf order like build_bvh8 does.
using Collapse8Fn = std::function<bool(std::vector<f4h>& nodes2, std::vector<int>& order, std::vector<f4h>& nodes8, uint32_t& depth8, std::string& err)>;

// Validates the description (indices in range, sizes non-zero), flattens faces, collects lights, builds the BVH.
// Returns MCPT_OK or an error code with `err` filled.
mcpt_status build_host_scene(const mcpt_scene_desc* d, HostScene& out, std::string& err, const BvhBuildFn& custom_bvh = nullptr,
                             const Collapse8Fn& custom_collapse8 = nullptr);

// Host-side soundness check of the quantised 8-wide tree (empty string = sound); run by mcpt_check_scene.
std::string validate_bvh8(const HostScene& hs);
inline std::string validate_wide_bvh(const HostScene& hs) { return validate_bvh8(hs); }
