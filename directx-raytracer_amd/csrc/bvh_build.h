// Host BVH builder of the product: stands in for BuildRaytracingAccelerationStructure
// (R/DXRTRenderer.cpp:548-806; driver-built BLAS per mesh + TLAS with identity transforms => one flat
// world-space hierarchy here).  Deterministic binned SAH, spec in DESIGN.md "BVH build".
#pragma once

#include "../../include/crt_hip.h"

#include <cstdint>
#include <vector>

struct ihipStream_t;

namespace crt {

constexpr int kLeafMax = 4;       // triangles per leaf
constexpr int kLbvhLeafMax = 2;   // GPU LBVH: a Karras node over at most this many triangles becomes a leaf (oracle: LBVH_LEAF_MAX)
constexpr int kMaxDepth = 32;     // leaves at depth <= kMaxDepth => traversal stack <= kMaxDepth entries
constexpr int kBins = 16;
constexpr float kTravCost = 1.0f; // SAH cost of an inner-node visit, in triangle tests

struct Bvh {
    std::vector<crt_bvh_node> nodes;   // binary tree, 64 B each, DFS pre-order, node 0 = root (builder output, host only)
    std::vector<crt_bvh_node4> nodes4; // wide tree collapsed from it, 128 B each, DFS pre-order: what is uploaded and traversed
    std::vector<crt_bvh_node4q> nodes4q; // its 64-byte quantised form (quantizeBvh4): what is uploaded and traversed
    uint32_t depth4 = 0;               // levels of the wide tree
    std::vector<crt_bvh_tri> tris;     // 48 B each, leaf order
    std::vector<crt_bvh_shade> shade;  // 48 B each, leaf order
    std::vector<crt_bvh_uv> uvs;       // 24 B each, leaf order; empty when no mesh has uvs
    uint32_t maxDepth = 0;
    uint32_t nTris = 0;                // triangles; = tris.size() unless the leaf-ordered records exist on the device only:
    // GPU builder: tris / shade / uvs already sit in HBM (hipMalloc'ed, + 64 bytes of slack); whoever takes the Bvh owns them
    void* devTris = nullptr;
    void* devShade = nullptr;
    void* devUvs = nullptr;
    // ... and so do the trees: binary nodes, wide nodes, quantised wide nodes (nodes / nodes4 / nodes4q stay empty on the host;
    // nNodes / nNodes4 hold the counts).  For a tree built on the host nNodes = nodes.size(), nNodes4 = nodes4.size().
    void* devNodes = nullptr;
    void* devNodes4 = nullptr;
    void* devNodes4q = nullptr;
    uint32_t nNodes = 0, nNodes4 = 0;
    // packed wide tree (bvh_pack.h; width 4 or 8, 0 = none: the legacy 64-byte nodes above are what is traversed): node and
    // triangle records in ONE buffer; shade / uvs are then indexed by gid (input order), not by leaf position
    uint32_t width = 0;
    std::vector<uint32_t> packed;      // dwords
    uint32_t packedGranules = 0;       // length of the buffer in granules (48 B for width 4, 16 B for width 8)
    uint32_t nWide = 0, depthWide = 0; // wide nodes, levels
    void* devPacked = nullptr;
};

// meshes in InstanceID order; triangle gid = running ordinal over meshes. Throws std::runtime_error on bad input.
// width: 0 = legacy 4-wide tree with 64-byte nodes and separate triangle array; 4 / 8 = packed wide tree (bvh_pack.h)
void buildBvh(const crt_mesh_view* meshes, uint32_t n_meshes, Bvh& out, int width = 0);
// shared first step of both builders (see bvh_build.cpp)
void flattenMeshes(const crt_mesh_view* meshes, uint32_t n_meshes, std::vector<crt_bvh_tri>& inTri, std::vector<crt_bvh_shade>& inShade,
                   std::vector<float>& boxCent);
// LBVH on the GPU (bvh_gpu.hip): same output layout, lower quality, much faster; throws std::runtime_error on HIP errors
void buildBvhGpu(const crt_mesh_view* meshes, uint32_t n_meshes, Bvh& out, struct ihipStream_t* stream, double* device_ms);
// per-triangle uvs in input (gid) order, empty when no mesh has any; and their permutation to leaf order
void flattenUvs(const crt_mesh_view* meshes, uint32_t n_meshes, std::vector<crt_bvh_uv>& inUv);
void reorderUvs(const std::vector<crt_bvh_uv>& inUv, Bvh& bvh);
// binary -> wide collapse (DESIGN.md "BVH4"); called by both builders
void collapseBvh4(Bvh& bvh);
// binary -> W-wide collapse + packing into the single buffer; needs bvh.nodes and the leaf-ordered bvh.tris
void packBvh(Bvh& bvh, int width);
// (the collapse and quantisation rules themselves: bvh_wide.h, shared with the GPU builder)

} // namespace crt
