// Launch interface between the C-ABI layer (crt_api.cpp, host C++) and the HIP kernels (render_kernels.hip, path_kernels.hip; device code shared through traversal.hip.h / shading.hip.h).
#pragma once

#include <cstdint>

struct ihipStream_t;

namespace crt {

constexpr int kMaxBatch = 4;       // frames per launch (crt_render_tiles_batch_device)
constexpr int kTile = 16;          // macro tile edge: one 256-thread workgroup = 4 wavefronts of 8x8 pixels
constexpr int kStackEntries = 32;  // upper bound of the per-lane LDS traversal stack = kMaxDepth of the builder

struct RenderParams {
    // scene (HBM)
    const void* nodes;   // crt_bvh_node4q[n_nodes], 64 B (the quantised wide tree); layout 4 / 8: the packed buffer (bvh_pack.h)
    const void* tris;    // crt_bvh_tri[n_tris], 48 B; layout 4 / 8: the same packed buffer
    uint32_t layout;     // 0: legacy 64-byte 4-wide nodes + triangle array; 4 / 8: packed wide tree of that width
    const void* shade;   // crt_bvh_shade[n_tris], 48 B
    const void* lights;  // crt_light[n_lights]
    const void* mats;    // crt_material[n_mats] (28 B)
    const void* uvs;     // crt_bvh_uv[n_tris] (24 B, leaf order) or null
    const void* textures; // TextureRec[n_textures] (below), bitmaps' texels in `texels`
    const unsigned char* texels;
    uint32_t n_textures;
    uint32_t n_nodes, n_tris, n_lights, n_mats;
    // per-frame constants: CameraCB (R/DXRTRenderer.h:54-59) + DebugCB (:68-72)
    float pos[3];
    float rot[9];
    float miss[3];
    uint32_t mode;
    uint32_t spp, max_bounces, seed; // mode 200 (path tracing)
    float phong_ks;                  // mode 100: specular coefficient (0 = plain Lambert), options "phong_ks" / "phong_exponent"
    uint32_t phong_exp;
    uint32_t width, height;
    // tiling
    uint32_t tiles_x, tiles_y;   // ceil(width/16), ceil(height/16)
    uint32_t rank, n_ranks;      // this launch renders macro tiles k with k % n_ranks == rank
    uint32_t n_local_tiles;      // grid size: number of such tiles
    uint32_t staging;            // 0: write row-major frame buffers; 1: rgba8 goes to the tile-major staging buffer
    uint32_t tune_inner_min;     // wave scheduling knob of the closest-hit traversal, see closestIteration() (int: negative = adaptive)
    uint32_t tune_inner_min_any; // the same for the any-hit (shadow ray) traversal
    uint32_t stack_entries;      // per-lane stack entries kept in LDS; deeper ones go to the spill arena
    uint32_t debug_skip_units;   // diagnostics: with unit_order, the first N work units are not rendered
    uint32_t boost_units;        // with unit_order: the first boost_units (most expensive) work units run at raised priority
    uint32_t split_units;        // with unit_order: the first split_units work units are rendered by FOUR wavefronts, one per 4x4
                                 // quarter of the 8x8 packet, every ray by four lanes (four segments of its way through the scene)
    uint32_t split_rays_log2;    // rays (pixels) per wavefront of a split packet: 4 -> 16 (four wavefronts per packet), 3 -> 8, 2 -> 4 (sixteen)
    uint32_t split_segs_log2;    // pieces a split ray is cut into: 2 -> 4, 3 -> 8, 4 -> 16
    float scene_lo[3], scene_hi[3]; // the scene's box (root of the tree): where a split ray's segments are cut
    uint32_t xcd_group;          // consecutive tiles of the list handed to one XCD before moving to the next (1..16, power of 2)
    // outputs (device pointers, nullable except rgba8)
    uint32_t* rgba8;
    uint32_t* hit_inst;
    uint32_t* hit_prim;
    float* hit_t;
    float* rgb_f32;
    unsigned long long* counters; // [0] nodes fetched, [1] triangles fetched, [2] shadow rays, [3] closest-hit rays; counting variant
    int* spill;                   // stack spill arena: renderUnitCount() x 64 lanes x spill_stride ints, rarely touched
    uint32_t spill_stride;        // >= 3 * wide depth + 1 - stack_entries: the deepest stack any ray can build
    const uint32_t* unit_order;   // nullable: work units sorted by descending cost of the previous frame (launch order)
    uint32_t* unit_cost;          // nullable: per work unit, traversal-loop iterations of its wavefront (this frame)
    // batch: n_batch frames (1..kMaxBatch) in ONE launch, grid = n_batch x units_per_frame; frame 0 uses pos / rot / rgba8
    // above, frame f > 0 its own camera and output below (hit ids / float colour are frame 0's only)
    uint32_t n_batch;
    uint32_t units_per_frame;
    float batch_pos[3][3];
    float batch_rot[3][9];
    uint32_t* batch_rgba8[3];
    // mode 200: per-workgroup scratch of the wavefront-private path pipeline (path_kernels.hip pathKernel)
    unsigned char* path_scratch;  // pathGridSize() regions of path_region_bytes: one per RESIDENT workgroup of the persistent kernel
    uint32_t* path_counter;       // work-item counter of the launch (zeroed on the stream before it)
    uint32_t path_work_items;     // pathWorkgroupCount(): pixel tiles x frames of the batch
    uint32_t path_ranges;         // 1 or 8: contiguous ranges of the work items, one counter (64 bytes apart) and one home XCD each
    size_t path_region_bytes;     // pathRegionBytes(path_samples)
    uint32_t path_tile;           // 16: one workgroup per 16x16 macro tile; 8: one per 8x8 packet
    uint32_t path_samples;        // samples of the tile carried through the pipeline together: B = tile^2 x this paths (<= 1024)
    // mode 200, wavefront pipeline (path_pipeline 1): the stages as separate launches over global queues (path_kernels.hip)
    uint32_t path_wavefront;      // 1: launchPath runs the wavefront pipeline
    void* wf_shade_q;             // float4 arrays.  3 planes x wf_paths: {o, rng} {d, id | bounce << 25} {t, u, v, tri}
    void* wf_trace_q;         // 2 planes x wf_paths
    void* wf_done;            // wf_paths: a path's radiance so far
    void* wf_thr;             // wf_paths: its throughput
    void* wf_accum;           // nullable: 64 per work item of a pass, sums of earlier passes (spp > path_samples)
    uint32_t* wf_counts;          // kWfHeadBytes: work counters of the camera launch, then {length, cursor} per queue
    uint32_t wf_paths;            // capacity of a pass in paths = plane stride (pathWavefrontPassItems() x 64 x path_samples)
    uint32_t wf_stride, wf_chunk; // the queues' capacity in entries (= plane stride) and the entries a wavefront reserves per atomic (pathWavefrontLayout)
    uint32_t wf_item0, wf_items;  // the work items of this pass
    uint32_t wf_queue;            // queue a shade / trace launch consumes (it fills wf_queue + 1)
    uint32_t wf_s0;               // first sample of this pass
    unsigned long long* timeline; // counting variant only, nullable: per workgroup {start, end} of s_memrealtime (100 MHz) + XCC id
};

// Enqueue the fused rayGen -> traverse -> shade -> store kernel. counting selects the instrumented variant.
int launchRender(const RenderParams& p, bool counting, ihipStream_t* stream);
// number of work units (= workgroups) launchRender uses for p: size of unit_order / unit_cost
uint32_t renderUnitCount(const RenderParams& p);
// mode 200 (path_kernels.hip): the wavefront-private path pipeline; launchRender forwards to launchPath
int launchPath(const RenderParams& p, bool counting, ihipStream_t* stream);
// mode 200 scratch sizing
size_t pathRegionBytes(uint32_t tile, uint32_t samples_per_pass);
uint32_t pathWorkgroupCount(const RenderParams& p);
// wavefront pipeline: bytes in front of the queues (counters), work items one pass may carry for a budget of paths, and the
// arena a pass of `items` work items needs
constexpr size_t kWfHeadBytes = 4096;
uint32_t pathWavefrontPassItems(const RenderParams& p, uint32_t max_paths);
void pathWavefrontLayout(const RenderParams& p, uint32_t items, uint32_t& chunk, uint32_t& stride);
size_t pathWavefrontBytes(const RenderParams& p, uint32_t items);
uint32_t pathGridSize(const RenderParams& p); // workgroups the persistent path kernel starts: min(work items, what the chip holds at once)
// unit_cost -> unit_order (descending)
int launchSortUnits(const uint32_t* cost, uint32_t* order, uint32_t n, bool xcdAffine, ihipStream_t* stream);
// tile-major gathered buffer -> row-major frame
int launchUntile(const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height, uint32_t n_ranks,
                 uint32_t rank_stride, uint32_t first_slot, ihipStream_t* stream);

// device-side texture record (crt_texture with the pixel pointer replaced by an offset into the texel pool)
struct TextureRec {
    uint32_t type;
    float a[3], b[3];
    float scalar;
    uint32_t texel_offset, width, height, channels;
};

} // namespace crt
