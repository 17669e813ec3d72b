/*
 * crt_oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (directx-raytracer_amd/, include/crt_hip.h) never links, imports or calls it.
 *
 * What it restates (R/ = /root/reference/DirectX-RayTracer/DirectX-RayTracer/):
 *   - rayGen      R/HLSL/ray_tracing_shaders.hlsl:21-70   (pinned by reference source)
 *   - miss        R/HLSL/ray_tracing_shaders.hlsl:72-76   (pinned)
 *   - closestHit  R/HLSL/ray_tracing_shaders.hlsl:78-169  (pinned, 7 shading modes)
 *   - TraceRay semantics selected by the host code: opaque, no culling, TMin < t < TMax,
 *     InstanceID = mesh ordinal, PrimitiveIndex = triangle ordinal
 *     (R/DXRTRenderer.cpp:588-616,690-704; hlsl:51-66)
 *   - RGBA8 UNORM store (R/DXRTRenderer.cpp:921-946)
 *
 * PARITY UNPINNED for everything the reference delegates to the closed DXR driver / RT hardware and
 * therefore has no source, test or golden image for: BVH construction, BVH traversal order, the
 * ray/triangle intersection arithmetic (here: Moeller-Trumbore as BASELINE.json's north_star asks),
 * equal-t tie breaking, the GPU's sin() and the Lambert/shadow-ray extension (mode 100).  For those
 * this file is the *specification* the HIP kernels are held to bit-for-bit (hit ids) / 1e-4 (floats);
 * see DESIGN.md "Arithmetic contract".
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MISS 0xFFFFFFFFu
#define ORACLE_MODE_LAMBERT 100u
#define ORACLE_MODE_PATH 200u

/* 64-byte inner node, both children's boxes stored in the parent. child ref >= 0: inner node index;
 * child ref < 0: leaf, ~ref = (first_triangle << 3) | count (count 0..4). */
typedef struct oracle_node {
    float lx0, lx1, ly0, ly1; /* left  child: min.x max.x min.y max.y */
    float rx0, rx1, ry0, ry1; /* right child: min.x max.x min.y max.y */
    float lz0, lz1, rz0, rz1; /* left min.z max.z, right min.z max.z  */
    int32_t left, right;
    int32_t pad0, pad1;
} oracle_node;

/* 128-byte wide node: up to four children, planes stored per axis across the children (SoA). Built by collapsing the
 * binary tree (DESIGN.md "BVH4"). ref >= 0: wide node index; negative: leaf as above; unused slot: ORACLE_EMPTY = the leaf of no
 * triangles; inverted box here, the point at the node's minimum corner in the quantised node (missed by the slab test like
 * any box the ray does not pass through; never a test of its own). */
#define ORACLE_EMPTY ((int32_t)-1)
typedef struct oracle_node4 {
    float minx[4], maxx[4], miny[4], maxy[4], minz[4], maxz[4];
    int32_t ref[4];
    int32_t pad[4];
} oracle_node4;

/* 64-byte quantised wide node: what the kernels fetch and what the wide walk below tests (DESIGN.md "Quantised nodes").
 * Child k's box on axis a = [fma(qlo_a.byte[k], s[a], lo[a]), fma(qhi_a.byte[k], s[a], lo[a])], a superset of the
 * full-precision box in oracle_node4. */
typedef struct oracle_node4q {
    float lo[3];
    float s[3];
    uint32_t qlo_x, qhi_x, qlo_y, qhi_y, qlo_z, qhi_z;
    int32_t ref[4];
} oracle_node4q;

/* 48-byte leaf-ordered triangle: v0 and the two edges, ids in the w lanes. */
typedef struct oracle_tri {
    float v0[3]; uint32_t inst; /* mesh ordinal      (DXR InstanceID)     */
    float e1[3]; uint32_t prim; /* triangle ordinal  (DXR PrimitiveIndex) */
    float e2[3]; uint32_t gid;  /* global triangle ordinal, tie-break key */
} oracle_tri;

/* 48-byte shading record, same order as oracle_tri: three vertex normals + material. */
typedef struct oracle_shade {
    float n0[3]; float n1[3]; float n2[3];
    uint32_t material; uint32_t pad[2];
} oracle_shade;

/* 24-byte per-triangle texture coordinates, same order as oracle_tri (only when some mesh has uvs) */
typedef struct oracle_uv { float uv0[2], uv1[2], uv2[2]; } oracle_uv;

typedef struct oracle_mesh {
    const float* xyz;       /* n_vertices * 3 */
    const uint32_t* idx;    /* n_triangles * 3 */
    const float* normals;   /* n_vertices * 3, or NULL (then flat shading) */
    const float* uvs;       /* n_vertices * 3 (u, v, unused: the .crtscene "uvs" layout), or NULL */
    uint32_t n_vertices;
    uint32_t n_triangles;
    int32_t material_index;
} oracle_mesh;

typedef struct oracle_light { float pos[3]; float intensity; } oracle_light;
typedef struct oracle_material { float albedo[3]; uint32_t type; uint32_t smooth; float ior; int32_t texture; /* -1 none */ } oracle_material;
/* R/CRTTexture*.cpp: type 0 albedo (color_a), 1 edges (color_a edge, color_b inner, scalar = edge width; sampled with the
 * hit's barycentrics), 2 checker (color_a/b, scalar = square size), 3 bitmap (pixels, nearest texel, v flipped); checker and
 * bitmap are sampled with the interpolated mesh uvs */
typedef struct oracle_texture { uint32_t type; float color_a[3]; float color_b[3]; float scalar; const uint8_t* pixels; uint32_t width, height, channels; } oracle_texture;

typedef struct oracle_stats {
    uint64_t rays_primary, rays_shadow; /* closest-hit rays (camera + bounce) / any-hit shadow rays */
    uint64_t nodes_visited, tris_tested; /* over all rays traced */
    uint64_t pixels;
} oracle_stats;

typedef struct oracle_scene oracle_scene;

/* Build the scene (flattens meshes, builds the BVH with the deterministic binned-SAH spec). */
oracle_scene* oracle_scene_create(const oracle_mesh* meshes, uint32_t n_meshes,
                                  const oracle_light* lights, uint32_t n_lights,
                                  const oracle_material* mats, uint32_t n_mats);
void oracle_scene_destroy(oracle_scene* s);
/* build_mode 0: binned SAH (default, as oracle_scene_create); 1: LBVH (30-bit Morton codes + Karras 2012 hierarchy,
 * ranges of <= 4 triangles collapsed to leaves) -- the spec of the product's GPU builder (csrc/bvh_gpu.hip) */
oracle_scene* oracle_scene_create_ex(const oracle_mesh* meshes, uint32_t n_meshes,
                                     const oracle_light* lights, uint32_t n_lights,
                                     const oracle_material* mats, uint32_t n_mats, int build_mode);

/* Replace the BVH by an externally built one (nodes + leaf-ordered triangles + shading records). */
int oracle_scene_set_bvh(oracle_scene* s, const oracle_node* nodes, uint32_t n_nodes,
                         const oracle_tri* tris, const oracle_shade* shade, uint32_t n_tris);

/* textures referenced by oracle_material.texture (copied, pixels too) */
int oracle_scene_set_textures(oracle_scene* s, const oracle_texture* tex, uint32_t n);
const oracle_uv* oracle_scene_uvs(const oracle_scene* s); /* NULL when no mesh has uvs */
/* R/CRTTexture*.cpp getColor restated (known answers: tests/golden/texture_known_answers.json) */
void oracle_texture_color(const oracle_texture* t, float u, float v, float out_rgb[3]);
uint32_t oracle_scene_node_count(const oracle_scene* s);
uint32_t oracle_scene_node4_count(const oracle_scene* s);
const oracle_node4* oracle_scene_nodes4(const oracle_scene* s);
const oracle_node4q* oracle_scene_nodes4q(const oracle_scene* s); /* node4_count entries */
uint32_t oracle_scene_depth4(const oracle_scene* s);
/* traversal width used by oracle_render: 4 (default, the wide tree the kernels walk) or 2 (the binary tree it is collapsed from) */
void oracle_scene_set_width(oracle_scene* s, int width);
uint32_t oracle_scene_tri_count(const oracle_scene* s);
const oracle_node* oracle_scene_nodes(const oracle_scene* s);
const oracle_tri* oracle_scene_tris(const oracle_scene* s);
const oracle_shade* oracle_scene_shade(const oracle_scene* s);
uint32_t oracle_scene_max_depth(const oracle_scene* s);

/*
 * Render rows y = y_begin, y_begin + y_step, ... < y_end of a w x h frame.  Output arrays are full
 * frame sized (w*h), rows not rendered are left untouched.  Any output pointer may be NULL.
 *   rot: 3x3 row-major camera matrix, dirWorld = rot * dirCam  (R/DXRTRenderer.cpp:259-264, hlsl:47)
 *   brute_force != 0: closest hit / occlusion by testing every triangle (no BVH) -- pins the BVH path.
 *   n_threads <= 0: OpenMP default.
 */
/* optional per-pixel instrumentation for the next oracle_render call of the calling thread's process: two w*h uint32
 * arrays receiving (node fetches << 8 | triangle fetches... ) see crt_oracle.c; pass NULL to disable */
void oracle_set_cost_outputs(uint32_t* primary_nodes, uint32_t* primary_tris, uint32_t* shadow_nodes, uint32_t* shadow_tris);

int oracle_render(const oracle_scene* s, const float pos[3], const float rot[9], uint32_t mode,
                  const float miss_rgb[3], uint32_t w, uint32_t h,
                  uint32_t y_begin, uint32_t y_end, uint32_t y_step,
                  uint8_t* rgba8, uint32_t* hit_inst, uint32_t* hit_prim, float* hit_t, float* rgb_f32,
                  oracle_stats* stats, int brute_force, int n_threads);

/* mode 200 parameters (process-wide): samples per pixel, bounces after the camera ray, RNG seed */
void oracle_set_path_params(uint32_t spp, uint32_t max_bounces, uint32_t seed);
/* Phong specular term of mode 100 (extension, see crt_hip.h "phong_ks"): ks in thousandths (0 = off, the default), integer exponent */
void oracle_set_phong(uint32_t ks_permille, uint32_t exponent);
void oracle_set_stack_output(uint32_t* max_sp);

/* Small pure functions exposed for known-answer tests. */
void oracle_ray_dir(const float rot[9], uint32_t px, uint32_t py, uint32_t w, uint32_t h, float out_dir[3]);
float oracle_sinf(float x);
void oracle_shade_mode(uint32_t mode, uint32_t inst, uint32_t prim, float t, float u, float v,
                       const float o[3], const float d[3], float out_rgb[3]);
uint8_t oracle_unorm8(float c);
/* one triangle, Moeller-Trumbore as specified; returns 1 on hit with t in (tmin, tmax) */
int oracle_occluded(const oracle_scene* s, const float o[3], const float d[3], float tmin, float tmax, int brute);
int oracle_intersect_tri(const float o[3], const float d[3], const float v0[3], const float v1[3],
                         const float v2[3], float tmin, float tmax, float* t, float* u, float* v);
int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
