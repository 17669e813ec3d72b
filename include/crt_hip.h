/*
 * crt_hip.h -- C ABI of libcrt_hip.so, the MI355X (gfx950) drop-in for the reference's per-pixel
 * render loop.  Plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *
 * R/ = /root/reference/DirectX-RayTracer/DirectX-RayTracer/.  The reference has no FFI: its seam is the
 * C++ class DXRTRenderer (R/DXRTRenderer.h:74-94) plus the CRT* scene layer it owns.  Every entry point
 * below names the reference member it stands in for; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions: every function returning int returns CRT_OK (0) or a CRT_E* code and records a message
 * retrievable with crt_last_error().  The reference reports failure with assert()/ignored HRESULTs
 * (R/DXRTRenderer.cpp:75,113,130,...); the error code replaces that.  A context is used from one host
 * thread at a time (the reference is single threaded, R/DXRTApp.cpp:109-120).  The caller owns every host
 * buffer it passes; the context owns all device memory.  There is NO CPU fallback: without a usable HIP
 * device crt_create() fails with CRT_ENODEVICE.
 */
#ifndef CRT_HIP_H
#define CRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRT_ABI_VERSION 1

enum {
    CRT_OK = 0,
    CRT_EINVAL = 1,    /* bad argument */
    CRT_ENODEVICE = 2, /* no HIP device / HIP runtime error at init */
    CRT_EHIP = 3,      /* HIP runtime error */
    CRT_ENOMEM = 4,
    CRT_ESTATE = 5,    /* call order (e.g. render before upload) */
    CRT_EIO = 6,       /* scene file */
    CRT_EPARSE = 7
};

#define CRT_MISS 0xFFFFFFFFu
/* shading modes: 0..6 are the reference's closest-hit modes (R/HLSL/ray_tracing_shaders.hlsl:78-169,
 * UI names R/DXRTMainWindow.cpp:100-106; any value in 6..99 behaves as 6, like the shader's final else).
 * >= 100 are extensions that do not exist in the reference. */
#define CRT_MODE_RANDOM_TRIANGLE 0u
#define CRT_MODE_OBJECT_CELLS 1u
#define CRT_MODE_OBJECT_TRIANGLE 2u
#define CRT_MODE_BARYCENTRIC 3u
#define CRT_MODE_HEIGHT 4u
#define CRT_MODE_DISTANCE 5u
#define CRT_MODE_CHECKER 6u
#define CRT_MODE_LAMBERT 100u /* Lambert (+ optional Phong highlight, options "phong_ks" / "phong_exponent") + one shadow ray per light
                                 (BASELINE.json north_star: "Lambert/Phong shading") */
#define CRT_MODE_PATH 200u    /* path tracing: options "spp" (default 4), "max_bounces" (3), "seed" (1234); BASELINE.json configs[4] */

/* ---- geometry handed over at upload: exactly what createVertexBuffers / createIndexBuffers memcpy ----
 * (R/DXRTRenderer.cpp:391-392,411 and :314-315,334): float xyz stride 12, uint32 indices, mesh ordinal
 * = InstanceID (R/DXRTRenderer.cpp:696). */
typedef struct crt_mesh_view {
    const float* xyz;      /* n_vertices * 3                                  (CRTMesh::getVertices) */
    const uint32_t* idx;   /* n_triangles * 3                                 (CRTMesh::getIndices) */
    const float* normals;  /* n_vertices * 3 or NULL                          (CRTMesh::getVertexNormals) */
    const float* uvs;      /* n_vertices * 3 (u, v, unused) or NULL           (CRTMesh::getUV) */
    uint32_t n_vertices;
    uint32_t n_triangles;
    int32_t material_index; /*                                                (CRTMesh::getMaterialIndex) */
} crt_mesh_view;

typedef struct crt_light { float pos[3]; float intensity; } crt_light;                 /* R/CRTLight.h:4-16 */
typedef struct crt_material { float albedo[3]; uint32_t type; uint32_t smooth; float ior; int32_t texture; } crt_material; /* R/CRTMaterial.h:4-36;
    type = CRTMaterialType; texture = index into crt_set_textures' array when CRTMaterial::isTexture(), else -1 */
/* R/CRTTexture*.h: type 0 albedo (color_a), 1 edges (color_a = edge colour, color_b = inner colour, scalar = edge width; evaluated
 * on the hit's barycentrics), 2 checker (color_a / color_b, scalar = square size), 3 bitmap (pixels: height x width x channels
 * bytes, channels >= 3, nearest texel, v flipped); 2 and 3 are evaluated on the mesh uvs interpolated at the hit */
enum { CRT_TEX_ALBEDO = 0, CRT_TEX_EDGES = 1, CRT_TEX_CHECKER = 2, CRT_TEX_BITMAP = 3 };
typedef struct crt_texture { uint32_t type; float color_a[3]; float color_b[3]; float scalar; const uint8_t* pixels; uint32_t width, height, channels; } crt_texture;

/* 64-byte BVH node and 48-byte leaf-ordered triangle/shading records as they sit in HBM */
typedef struct crt_bvh_node {
    float lx0, lx1, ly0, ly1, rx0, rx1, ry0, ry1, lz0, lz1, rz0, rz1;
    int32_t left, right; /* >= 0 inner node index; < 0 leaf: ~ref = (first_tri << 3) | count */
    int32_t pad0, pad1;
} crt_bvh_node;
/* 128-byte wide node (full-precision child boxes; the builders' output, host side): up to 4 children, planes stored per axis across the children; collapsed from
 * the binary tree above. ref >= 0: wide node index; negative: leaf (as above).  An unused slot is CRT_BVH_EMPTY = the leaf
 * of no triangles (~0); here its box is inverted (+inf, -inf), in the quantised node it is a single point.  Traversal has no
 * separate test for it: a ray misses the point like any other box it does not pass through, and a ray that did pass exactly
 * through it would visit a leaf without triangles. */
#define CRT_BVH_EMPTY ((int32_t)-1)
typedef struct crt_bvh_node4 {
    float minx[4], maxx[4], miny[4], maxy[4], minz[4], maxz[4];
    int32_t ref[4];
    int32_t pad[4];
} crt_bvh_node4;
/* 64-byte quantised form of the wide node -- what sits in HBM and what the kernels traverse.  lo = minimum corner of the
 * node's own box (union of its children), s = per-axis quantum; child k's box on axis a is
 * [fma(qlo_a.byte[k], s_a, lo_a), fma(qhi_a.byte[k], s_a, lo_a)], rounded outwards (it always contains the full-precision
 * box of crt_bvh_node4), so traversal results are unchanged and only the fetch counts differ by a percent or two.
 * Derived from crt_bvh_node4 by a fixed rule (DESIGN.md "Quantised nodes"; csrc/bvh_build.cpp quantizeBvh4 and the
 * oracle's restatement agree byte for byte).  Unused slots: ref = CRT_BVH_EMPTY, qlo = qhi = 0: the point at the node's minimum
 * corner (a valid box, so the min/max form of the slab test and the octant-specialised one agree on it as on any other). */
typedef struct crt_bvh_node4q {
    float lo[3];
    float s[3];
    uint32_t qlo_x, qhi_x, qlo_y, qhi_y, qlo_z, qhi_z; /* byte k (bits 8k..8k+7) = child k */
    int32_t ref[4];
} crt_bvh_node4q;
typedef struct crt_bvh_tri { float v0[3]; uint32_t inst; float e1[3]; uint32_t prim; float e2[3]; uint32_t gid; } crt_bvh_tri;
typedef struct crt_bvh_shade { float n0[3], n1[3], n2[3]; uint32_t material; uint32_t pad[2]; } crt_bvh_shade;
typedef struct crt_bvh_uv { float uv0[2], uv1[2], uv2[2]; } crt_bvh_uv; /* 24 B, leaf order, only when some mesh has uvs */

typedef struct crt_frame_stats {
    double kernel_ms;        /* HIP-event time of the render kernel(s) on the context's stream */
    double total_ms;         /* wall time of the call (includes D2H copies when host outputs are requested) */
    uint64_t rays_primary;   /* closest-hit rays: pixels rendered by this call (x spp + bounce rays in mode 200, exact when counting) */
    uint64_t rays_shadow;    /* counted only when counting is enabled, else 0 */
    uint64_t nodes_visited;  /* idem: 64-byte quantised wide-node records fetched, summed over all rays */
    uint64_t tris_tested;    /* idem: 48-byte triangle records fetched */
} crt_frame_stats;

typedef struct crt_ctx crt_ctx;

/* ---------------------------------------------------------------------------------------------------
 * Renderer: stands in for DXRTRenderer (R/DXRTRenderer.h:74-94)
 * ------------------------------------------------------------------------------------------------- */

/* DXRTRenderer::prepareForRendering minus window/swap chain (R/DXRTRenderer.cpp:44-62): bind to one HIP
 * device (one process per GPU), create the stream and timing events. */
int crt_create(crt_ctx** out, int device_id);
void crt_destroy(crt_ctx* ctx);
const char* crt_last_error(const crt_ctx* ctx); /* ctx may be NULL: last error of a failed crt_create */
uint32_t crt_abi_version(void);

/* createVertexBuffers + createIndexBuffers + createAccelerationStructures
 * (R/DXRTRenderer.cpp:379-453, 302-376, 548-806): copies geometry, builds the BVH on the host (binned SAH)
 * and uploads nodes / leaf-ordered triangles / shading records to HBM. The scene is immutable afterwards
 * (the reference never refits either); a second call replaces it. */
int crt_upload_scene(crt_ctx* ctx, const crt_mesh_view* meshes, uint32_t n_meshes,
                     const crt_light* lights, uint32_t n_lights,
                     const crt_material* materials, uint32_t n_materials);

/* textures the materials refer to by index (CRTScene::getTextures / getTextureByName, R/CRTScene.h:32-34); copied, pixels
 * included; may be called before or after crt_upload_scene. The reference parses them but its renderer never samples
 * them; here they drive the albedo of modes 100 and 200 */
int crt_set_textures(crt_ctx* ctx, const crt_texture* textures, uint32_t n_textures);

/* updateCameraCB (R/DXRTRenderer.cpp:248-270): position + 3x3 row-major rotation, dirWorld = R * dirCam */
int crt_set_camera(crt_ctx* ctx, const float pos[3], const float rot3x3_rowmajor[9]);
/* changeShadingMode (R/DXRTRenderer.cpp:1359-1363) */
int crt_set_shading_mode(crt_ctx* ctx, uint32_t mode);
/* miss colour; default (0,1,1) = the reference's miss shader (hlsl:72-76) */
int crt_set_miss_color(crt_ctx* ctx, const float rgb[3]);
/* when enabled the render kernels also count node/triangle fetches and shadow rays (slower variant) */
int crt_set_counting(crt_ctx* ctx, int enabled);

/* renderFrame (R/DXRTRenderer.cpp:1370-1408), synchronous like the reference (fence wait :521-527).
 * Host outputs, any may be NULL: rgba8 w*h*4 (R8G8B8A8_UNORM, row-major, top-left origin), hit_inst / hit_prim
 * w*h uint32 (CRT_MISS on miss), hit_t w*h float, rgb_f32 w*h*3 float (pre-quantisation colour). */
int crt_render_frame(crt_ctx* ctx, uint32_t width, uint32_t height,
                     uint8_t* rgba8, uint32_t* hit_inst, uint32_t* hit_prim, float* hit_t, float* rgb_f32,
                     crt_frame_stats* stats);

/* Same frame, outputs left in HBM: device pointers (any may be NULL except d_rgba8). Asynchronous on the
 * context's stream unless stats != NULL (then it synchronises to read the timers). */
int crt_render_frame_device(crt_ctx* ctx, uint32_t width, uint32_t height,
                            void* d_rgba8, void* d_hit_inst, void* d_hit_prim, void* d_hit_t, void* d_rgb_f32,
                            crt_frame_stats* stats);

/* ---- tile-partitioned rendering for N GPUs (no reference counterpart) ----------
 * The frame is cut into 16x16-pixel macro tiles, numbered row-major; macro tile k belongs to rank k % n_ranks.
 * A rank renders its tiles into a rank-contiguous, tile-major staging buffer: slot j (= k / n_ranks) holds
 * 256 RGBA8 pixels (row-major inside the tile; pixels outside the frame are left untouched).  All ranks use the
 * same slot count crt_tile_slots() so the per-rank buffers can be gathered with one RCCL all-gather/gather. */
uint32_t crt_tile_count(uint32_t width, uint32_t height);
uint32_t crt_tile_slots(uint32_t width, uint32_t height, uint32_t n_ranks); /* ceil(tile_count / n_ranks) */
int crt_render_tiles_device(crt_ctx* ctx, uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks,
                            void* d_staging_rgba8 /* slots*256*4 bytes */, crt_frame_stats* stats);
/* Throughput mode (no reference counterpart: DXRTRenderer::renderFrame issues one DispatchRays per call, R/DXRTRenderer.cpp:1348-1350):
 * n_frames (1..4) frames of the current scene, mode and size in ONE launch.  A launch lasts as long as its slowest 8x8 packet
 * (a grazing ray walks hundreds of dependent steps); a batch shares that critical path, which is what bounds an N-GPU tile
 * share (1/N of the work, same critical path).  cameras = n_frames x 12 floats {pos[3], rot3x3_rowmajor[9]} or NULL (the
 * current camera for every frame); d_rgba8[f] / d_staging[f] = frame f's output, laid out as in crt_render_frame_device /
 * crt_render_tiles_device.  Frames are independent: each equals what a single-frame call with its camera renders. */
int crt_render_frames_batch_device(crt_ctx* ctx, uint32_t width, uint32_t height, uint32_t n_frames, const float* cameras,
                                   void* const* d_rgba8, crt_frame_stats* stats);
int crt_render_tiles_batch_device(crt_ctx* ctx, uint32_t width, uint32_t height, uint32_t rank, uint32_t n_ranks, uint32_t n_frames,
                                  const float* cameras, void* const* d_staging, crt_frame_stats* stats);

/* De-interleave a gathered buffer (n_ranks * slots * 1024 bytes, rank-major) into a row-major frame. */
int crt_untile_device(crt_ctx* ctx, uint32_t width, uint32_t height, uint32_t n_ranks,
                      const void* d_gathered, void* d_rgba8_rowmajor);
/* the same for frame `frame` of an all-gathered BATCH: when each rank sends its n_frames staging buffers as one contiguous
 * message (n_frames x crt_tile_slots() tiles), d_gathered holds n_ranks x n_frames x crt_tile_slots() tiles */
int crt_untile_batch_device(crt_ctx* ctx, uint32_t width, uint32_t height, uint32_t n_ranks, uint32_t n_frames, uint32_t frame,
                            const void* d_gathered, void* d_rgba8);

/* ---- native RCCL frame assembly: the N-GPU frame without Python (no reference counterpart: the reference is one GPU, one
 * DispatchRays, R/DXRTRenderer.cpp:1346-1350, 1405).  One process per GPU, each with its own context and the same scene.
 * Rank 0 calls crt_comm_unique_id and hands the 128 bytes to the other ranks (file, pipe, MPI: the caller's business); every
 * rank calls crt_comm_init (collective: returns when all n_ranks have called it).  crt_render_frame_distributed then renders
 * this rank's macro tiles (crt_render_tiles_device), moves them with ONE ncclAllGather over xGMI -- crt_tile_slots() * 1024 bytes
 * per rank: 1 044 480 B at 1920x1080 on 8 GPUs, 4 177 920 B on 2 -- and de-interleaves the gathered
 * buffer; every rank ends up with the whole frame.  Asynchronous on the context's stream unless stats or host_rgba8 is given.
 * d_rgba8: device frame (w*h*4 bytes) or NULL (a context-owned buffer is used); host_rgba8: optional host copy.
 * RCCL is loaded at run time (librccl.so.1): a process that never calls these needs no RCCL. */
#define CRT_COMM_ID_BYTES 128
int crt_comm_unique_id(void* id_out /* CRT_COMM_ID_BYTES */);
int crt_comm_init(crt_ctx* ctx, uint32_t rank, uint32_t n_ranks, const void* unique_id);
/* The same frame assembly with shared host memory as the transport, for ranks that share ONE GPU (RCCL refuses that: rehearsing the
 * multi-rank path on a one-GPU machine) or where RCCL cannot be loaded: `name` ("/something") names a POSIX shared-memory object
 * that rank 0 creates and removes; collective like crt_comm_init.  Per frame every rank copies its tiles into the object, waits for
 * the others, and copies all tiles out.  Correct, slow, never a measurement. */
int crt_comm_init_host(crt_ctx* ctx, uint32_t rank, uint32_t n_ranks, const char* name);
int crt_comm_destroy(crt_ctx* ctx);
int crt_comm_info(const crt_ctx* ctx, uint32_t* rank, uint32_t* n_ranks); /* n_ranks = 0: no communicator */
int crt_render_frame_distributed(crt_ctx* ctx, uint32_t width, uint32_t height, void* d_rgba8, uint8_t* host_rgba8, crt_frame_stats* stats);

/* options. Phong term of mode 100: the reference's CRTMaterial has no specular coefficient or exponent
 * (R/CRTMaterial.h:30-35 holds type, albedo, smooth_shading, ior, texture), so both are renderer-wide options invented here:
 * "phong_ks" = specular coefficient in thousandths (0 = off, the default; 300 = 0.3), "phong_exponent" = integer exponent
 * 1..65536 (default 32).  Per unoccluded light: rgb += ks * intensity / (4 pi r^2) * max(0, R . V)^n, R = light direction
 * mirrored about the shading normal, V = direction to the eye; white highlight, x^n by square and multiply.
 * "gpu_build" 0/1: acceleration structure built on the GPU (LBVH) at the next crt_upload_scene.
 * Rendering parameters of mode 200: "spp", "max_bounces", "seed". Tuning knobs (speed only, results never
 * change): "inner_min" / "inner_min_any" wave scheduling of the closest-hit / any-hit traversal loops (1..65: node steps while that
 * many lanes stand on inner nodes; -1..-8: while that many eighths of the wavefront's live lanes do; default -6), "xcd_group", "adaptive_order" (launch the most expensive 8x8
 * packets of the previous frame first: 0 never, 1 always, 2 = default: only for a frame issued on the same stream as the frame
 * before, where frames run one after another), "remeasure_every" (a moving camera re-measures packet costs every n-th use of a scratch slot; default 1), "boost_units",
 * "split_units" (with a launch order: the n most expensive 8x8 packets are rendered by several wavefronts each, whose lanes
 * share the pieces of the block's rays -- same frame, shorter critical path; -1 = automatic, the default: none for a whole frame
 * on one GPU, 64 / 128 packets for the tile share of 2 / >= 4 ranks, whose lone launch goes from 221 / 199 / 178 us to 188 / 132 /
 * 118 us at 2 / 4 / 8 ranks; 0 = off; fetch counters of split packets grow), "split_rays" (16, 8 or 4 rays per wavefront of a
 * split packet, default 4), "split_segments" (4, 8 or 16 pieces per split ray, default 16),
 * "xcd_affine_order" 0/1 (with a launch order: the frame is cut into eight regions of equal cost, one per XCD and its L2, each launched
 * most expensive packet first; default 0 -- primary rays alone gain 5 %, a shaded frame loses 1 %),
 * "path_tile" (mode 200 work split: pixel-tile edge per workgroup, 8 (default, also 0) or 16), "path_pipeline" (mode 200: 0 (default) = one persistent kernel
 * whose wavefronts carry a pixel tile's paths through all stages with private queues; 1 = the stages as separate launches (camera rays,
 * shade + shadow rays, bounce rays, resolve) over global queues: no register spills, 6-7 wavefronts per SIMD instead of 5, identical
 * frames, 8 % slower on the 5M-triangle 4K frame because the compute-bound camera stage no longer overlaps the fetch-bound stream
 * stages; its queues hold "path_pass_paths" paths per pass (65536..2^25, default 2^24 = 1.9 GB), a frame with more is rendered in
 * several passes), "path_ranges" (mode 200: 8 (default) = the
 * work items are cut into eight contiguous ranges and a wavefront works through the range of the XCD it runs on before taking from the others', 1 = one shared work counter), "stack_entries" (0 = default 16; deeper entries spill to a
 * global arena). The diagnostic options "timeline", "debug_skip_units" and "debug_force_measure" (which do change what a frame
 * does) exist only in the diagnostic build of the library (tools/diag_build.sh); the product returns CRT_EINVAL for them. */
int crt_set_option(crt_ctx* ctx, const char* name, int value);

/* diagnostic build only (the product returns 0 words): with option "timeline" = 1 and counting enabled, a render records per workgroup {start, end} on the
 * 100 MHz s_memrealtime clock and (XCC id << 32 | tile_y << 16 | tile_x); this copies them out (3 words per workgroup) */
int crt_debug_read_timeline(crt_ctx* ctx, unsigned long long* out, size_t max_words, size_t* n_words);
/* raw device counters of the last counting render: [0] nodes [1] triangles [2] shadow rays [3] closest-hit rays; [4..31]
 * are filled only by the CRT_PROF diagnostic build of the kernels (tools/prof_build.sh, meanings in tools/prof_run.py) */
int crt_debug_read_counters(crt_ctx* ctx, unsigned long long out[32]);

/* stream plumbing: run on an external hipStream_t (e.g. torch's current stream; NULL = HIP's default stream);
 * crt_reset_stream goes back to the context's private non-blocking stream */
int crt_set_stream(crt_ctx* ctx, void* hip_stream);
int crt_reset_stream(crt_ctx* ctx);
int crt_synchronize(crt_ctx* ctx);

/* BVH introspection (tests, tooling): sizes, then copies of the host-side arrays uploaded to HBM */
int crt_bvh_info(const crt_ctx* ctx, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* max_depth);
int crt_bvh_export(const crt_ctx* ctx, crt_bvh_node* nodes, crt_bvh_tri* tris, crt_bvh_shade* shade);
/* per-triangle texture coordinates in leaf order; *has_uvs = 0 (and nothing copied) when no mesh carried uvs */
int crt_bvh_export_uv(const crt_ctx* ctx, crt_bvh_uv* uvs, int* has_uvs);
/* wall time of the last crt_upload_scene (flatten + build + collapse + upload) and, with option "gpu_build" = 1 (LBVH
 * built by HIP kernels instead of the host SAH builder: 8.7 against 360 ms at 1M triangles, frames over its tree 1.0 .. 1.14 x
 * the SAH tree's), the device time of the build kernels */
int crt_build_stats(const crt_ctx* ctx, double* upload_ms, double* device_build_ms);
/* the wide tree as it sits in HBM: count/depth (any pointer may be NULL), then a copy of the nodes */
int crt_bvh_info4(const crt_ctx* ctx, uint32_t* n_nodes4, uint32_t* depth4);
int crt_bvh_export4(const crt_ctx* ctx, crt_bvh_node4* nodes4);
/* the same nodes in the 64-byte quantised form the kernels fetch (count = n_nodes4) */
int crt_bvh_export4q(const crt_ctx* ctx, crt_bvh_node4q* nodes4q);
/* host-only BVH build, no device needed (used by crt_upload_scene; exposed for tests and tooling) */
int crt_bvh_build_host(const crt_mesh_view* meshes, uint32_t n_meshes,
                       crt_bvh_node** nodes, uint32_t* n_nodes,
                       crt_bvh_tri** tris, crt_bvh_shade** shade, uint32_t* n_tris, uint32_t* max_depth);
/* same build, additionally returning the collapsed wide tree (nodes4 freed with crt_free) */
int crt_bvh_build_host4(const crt_mesh_view* meshes, uint32_t n_meshes, crt_bvh_node4** nodes4, uint32_t* n_nodes4, uint32_t* depth4);
/* the quantisation rule alone: n wide nodes -> n quantised nodes (caller-provided output array) */
int crt_bvh_quantize4(const crt_bvh_node4* nodes4, uint32_t n, crt_bvh_node4q* out);
void crt_free(void* p);

/* page-locked host memory for crt_render_frame's output buffers: the device-to-host copy of a frame then runs at PCIe speed
 * instead of through the driver's staging of pageable memory (8.3 MB at 1080p: 0.36 -> 0.17 ms).  Optional: any host
 * pointer works.  Needs a HIP device; NULL on failure.  Release with crt_host_free. */
void* crt_host_alloc(size_t bytes);
void crt_host_free(void* p);

/* ---------------------------------------------------------------------------------------------------
 * Scene layer: stands in for CRTScene / CRTSceneParser / CRTCamera (kept API surface, host only, no GPU)
 * ------------------------------------------------------------------------------------------------- */
typedef struct crt_scene crt_scene;

/* CRTScene::CRTScene(file) -> CRTSceneParser::parseScene (R/CRTScene.cpp:7-15, R/CRTSceneParser.cpp:407-427).
 * Accepts .crtscene (JSON) and, as extensions, .obj and the binary cache .crtbin. Absent optional keys take defaults instead of the
 * reference's undefined behaviour. */
int crt_scene_load(const char* path, crt_scene** out, char* err, size_t err_len);
/* binary scene cache (.crtbin): writes everything a .crtscene yields, vertex normals included,
 * as raw arrays; crt_scene_load reads it back by extension at memcpy speed */
int crt_scene_save(const crt_scene* s, const char* path, char* err, size_t err_len);
/* empty scene to be filled programmatically */
int crt_scene_new(crt_scene** out);
void crt_scene_free(crt_scene* s);
/* CRTMesh::addVertex/addIndex/setMaterialIndex + calculateVertexNormals (R/CRTMesh.cpp:6-24,66-94) */
int crt_scene_add_mesh(crt_scene* s, const float* xyz, uint32_t n_vertices, const uint32_t* idx, uint32_t n_triangles,
                       int32_t material_index);
int crt_scene_add_light(crt_scene* s, const float pos[3], float intensity);
int crt_scene_add_material(crt_scene* s, const crt_material* m);

uint32_t crt_scene_mesh_count(const crt_scene* s);                               /* getObjects().size() */
int crt_scene_mesh(const crt_scene* s, uint32_t i, crt_mesh_view* out);          /* views into scene-owned memory */
uint32_t crt_scene_light_count(const crt_scene* s);                              /* getLights() */
int crt_scene_light(const crt_scene* s, uint32_t i, crt_light* out);
uint32_t crt_scene_material_count(const crt_scene* s);                           /* getMaterials() */
int crt_scene_material(const crt_scene* s, uint32_t i, crt_material* out);
uint32_t crt_scene_texture_count(const crt_scene* s);                            /* getTextures().size() */
/* CRTTexture::getColor(u, v) of texture i (R/CRTTexture*.cpp), evaluated on the host */
int crt_scene_texture_color(const crt_scene* s, uint32_t i, float u, float v, float out_rgb[3]);
/* programmatic textures: type = "albedo" | "edges" | "checker" | "bitmap" (file_path: binary PPM); a material refers to one by name */
int crt_scene_add_texture(crt_scene* s, const char* name, const char* type, const float color_a[3], const float color_b[3], float scalar,
                          const char* file_path);
int crt_scene_set_material_texture(crt_scene* s, uint32_t material, const char* texture_name);
int crt_scene_set_mesh_uvs(crt_scene* s, uint32_t mesh, const float* uvs /* n_vertices * 3 */);
int crt_scene_settings(const crt_scene* s, uint32_t* width, uint32_t* height, float background_rgb[3]); /* getSettings() */

/* CRTCamera (R/CRTCamera.h:5-32, .cpp:9-130) on the scene's camera */
int crt_scene_camera_get(const crt_scene* s, float pos[3], float rot[9]);
int crt_scene_camera_set(crt_scene* s, const float pos[3], const float rot[9]);
int crt_scene_camera_rotate(crt_scene* s, float delta_yaw_deg, float delta_pitch_deg);
int crt_scene_camera_zoom(crt_scene* s, float amount);
int crt_scene_camera_move_forward(crt_scene* s, float distance);
int crt_scene_camera_move_right(crt_scene* s, float distance);
int crt_scene_camera_pan(crt_scene* s, float degrees);
int crt_scene_camera_tilt(crt_scene* s, float degrees);
int crt_scene_camera_roll(crt_scene* s, float degrees);
int crt_scene_camera_pan_around_target(crt_scene* s, float degrees, const float target[3]);

/* convenience: crt_upload_scene + crt_set_camera from a loaded scene (what DXRTRenderer::createScene +
 * create*Buffers + createAccelerationStructures do with the CRTScene it owns, R/DXRTRenderer.cpp:243-246) */
int crt_upload_scene_from(crt_ctx* ctx, const crt_scene* s);
int crt_set_camera_from(crt_ctx* ctx, const crt_scene* s);

#ifdef __cplusplus
}
#endif
#endif
