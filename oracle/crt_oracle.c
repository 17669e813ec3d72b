/*
 * crt_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE (see crt_oracle.h).
 *
 * Plain C11 restatement of the reference's per-pixel render loop
 *   R/HLSL/ray_tracing_shaders.hlsl:21-169  (R/ = /root/reference/DirectX-RayTracer/DirectX-RayTracer/)
 * plus a build-defined specification of what the reference leaves to the DXR driver
 * (BVH build + traversal + ray/triangle test) and of the Lambert/shadow extension.
 *
 * Pinning status:
 *   - scene layer the oracle is fed from: pinned by tests/golden/dragon_scene_layer.json, produced by
 *     the reference's own CRT* sources compiled in place (oracle/Makefile, oracle/make_golden.py);
 *   - rayGen / miss / closestHit colour functions: restated line by line from the HLSL (citations at
 *     each function); the reference holds no golden image, so closed-form known answers are used;
 *   - traversal / intersection / tie-break / sin() / Lambert: PARITY UNPINNED against the reference
 *     (closed driver code, no fixtures); pinned internally by brute force (no BVH) == BVH.
 *
 * Arithmetic contract (mirrored exactly by the HIP kernels): IEEE-754 binary32, round to
 * nearest even, no flush-to-zero, compiled with -ffp-contract=off; a multiply-add is fused only where
 * fmaf()/fma() is written; division and sqrt correctly rounded; saturate uses fminf/fmaxf (IEEE minNum/maxNum: NaN -> 0); traversal/builder min/max are selects.
 */
#include "crt_oracle.h"

#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * constants
 * ---------------------------------------------------------------------------------------------- */
#define RAY_TMIN 0.001f   /* hlsl:51 */
#define RAY_TMAX 10000.0f /* hlsl:52 */
#define DIR_EPS 1e-20f    /* |d| below this is replaced by +-DIR_EPS for the slab reciprocal only */
#define LEAF_MAX 4
#define LBVH_LEAF_MAX 2 /* LBVH (build mode 1): a Karras node over at most this many triangles becomes a leaf */
#define MAX_DEPTH 32      /* leaves at depth <= MAX_DEPTH  => traversal stack <= MAX_DEPTH entries */
#define N_BINS 16
#define C_TRAV 1.0f       /* SAH: cost of visiting an inner node, in triangle tests */
#define CULL_PAD 1.00000381469726562f /* 1 + 2^-18: boxes are culled against best_t * CULL_PAD (see trace_closest) */
#define SHADOW_BIAS 1e-3f
#define FOUR_PI 12.566370614359172f

/* ------------------------------------------------------------------------------------------------
 * small math (explicit FMA placement is part of the contract)
 * ---------------------------------------------------------------------------------------------- */
static _Thread_local int t_max_sp; /* deepest stack of the rays traced by this thread since last reset (analysis only) */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 v3_normalize(v3 a)
{
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_make(a.x * inv, a.y * inv, a.z * inv);
}
/* min/max of the traversal and the builder: plain selects (x86 minss/maxss, GPU v_min/v_max agree with
 * them for every non-NaN input; the slab test and the builder never see a NaN for finite scenes) */
static inline float minf_(float a, float b) { return a < b ? a : b; }
static inline float maxf_(float a, float b) { return a > b ? a : b; }
static inline float fracf_(float x) { return x - floorf(x); }
static inline float saturatef_(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float lerpf_(float a, float b, float t) { return a + t * (b - a); }

/* sin() with a fixed operation sequence shared with the kernel: double range reduction by 2*pi
 * (two-term Cody-Waite with fma), odd Taylor polynomial to r^23 on [-pi, pi] in double (Horner, fma),
 * rounded once to float.  |error| < 1 ulp(float) for every finite float x up to 2^40. */
float oracle_sinf(float x)
{
    const double INV_2PI = 0x1.45f306dc9c883p-3;
    const double TWO_PI_HI = 0x1.921fb54442d18p+2;
    const double TWO_PI_LO = 0x1.1a62633145c07p-52;
    double xd = (double)x;
    double k = rint(xd * INV_2PI);
    double r = fma(-k, TWO_PI_HI, xd);
    r = fma(-k, TWO_PI_LO, r);
    double r2 = r * r;
    double p = -0x1.761b41316381ap-75;         /* -1/23! */
    p = fma(p, r2, 0x1.71b8ef6dcf572p-66);     /* +1/21! */
    p = fma(p, r2, -0x1.2f49b46814157p-57);    /* -1/19! */
    p = fma(p, r2, 0x1.952c77030ad4ap-49);     /* +1/17! */
    p = fma(p, r2, -0x1.ae7f3e733b81fp-41);    /* -1/15! */
    p = fma(p, r2, 0x1.6124613a86d09p-33);     /* +1/13! */
    p = fma(p, r2, -0x1.ae64567f544e4p-26);    /* -1/11! */
    p = fma(p, r2, 0x1.71de3a556c734p-19);     /* +1/9!  */
    p = fma(p, r2, -0x1.a01a01a01a01ap-13);    /* -1/7!  */
    p = fma(p, r2, 0x1.1111111111111p-7);      /* +1/5!  */
    p = fma(p, r2, -0x1.5555555555555p-3);     /* -1/3!  */
    p = p * r2;
    return (float)fma(p, r, r);
}

/* float4 -> R8G8B8A8_UNORM channel (R/DXRTRenderer.cpp:921-946): saturate, scale, round half up */
uint8_t oracle_unorm8(float c)
{
    float s = saturatef_(c); /* NaN -> 0 */
    return (uint8_t)(s * 255.0f + 0.5f);
}

/* ------------------------------------------------------------------------------------------------
 * scene
 * ---------------------------------------------------------------------------------------------- */
struct oracle_scene {
    oracle_node* nodes;   /* binary tree (the builder's output) */
    uint32_t n_nodes;
    oracle_node4* nodes4; /* wide tree collapsed from it */
    oracle_node4q* nodes4q; /* its quantised form (what the render loop walks) */
    uint32_t n_nodes4;
    uint32_t depth4;
    int width;
    oracle_tri* tris;     /* leaf order */
    oracle_shade* shade;  /* leaf order */
    uint32_t n_tris;
    uint32_t max_depth;
    oracle_light* lights;
    uint32_t n_lights;
    oracle_material* mats;
    uint32_t n_mats;
    oracle_uv* uvs;       /* leaf order, NULL when no mesh has uvs */
    oracle_texture* tex;  /* owns copies of the pixel buffers */
    uint32_t n_tex;
};

typedef struct { float mn[3], mx[3]; } aabb;

static inline void aabb_empty(aabb* b)
{
    for (int a = 0; a < 3; a++) { b->mn[a] = INFINITY; b->mx[a] = -INFINITY; }
}
static inline void aabb_grow(aabb* b, const aabb* o)
{
    for (int a = 0; a < 3; a++) { b->mn[a] = minf_(b->mn[a], o->mn[a]); b->mx[a] = maxf_(b->mx[a], o->mx[a]); }
}
static inline float aabb_half_area(const aabb* b)
{
    float dx = b->mx[0] - b->mn[0], dy = b->mx[1] - b->mn[1], dz = b->mx[2] - b->mn[2];
    return (dx * dy + dy * dz) + dz * dx;
}

typedef struct {
    const aabb* pbox;    /* per input triangle */
    const float* pcent;  /* per input triangle, 3 floats */
    uint32_t* order;     /* permutation being partitioned */
    uint32_t* tmp;
    oracle_node* nodes;
    uint32_t n_nodes;
    uint32_t max_depth;
} builder;

/* bin of a scaled centroid offset: [0, N_BINS-1]; NaN (a triangle with a NaN vertex) goes to bin 0 instead of UB */
static inline int bin_of(float f) { return f >= 0.0f ? (f < (float)N_BINS ? (int)f : N_BINS - 1) : 0; }

static inline int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | count); }

/* Deterministic binned-SAH build of order[first, first+count). Returns the child reference and the
 * exact bounds of the range. Spec (DESIGN.md "BVH build"): 16 bins per axis over the centroid bounds,
 * candidates scanned axis 0..2, plane 1..15, strict '<' keeps the first minimum; stable partition;
 * ranges of <= LEAF_MAX triangles are split only if that lowers the SAH cost; a split that would make
 * the depth bound unreachable, or no valid split, falls back to the median of the current order. */
static int32_t build_range(builder* B, uint32_t first, uint32_t count, uint32_t depth, aabb* out_box)
{
    aabb box, cbox;
    aabb_empty(&box);
    aabb_empty(&cbox);
    for (uint32_t i = first; i < first + count; i++) {
        uint32_t p = B->order[i];
        aabb_grow(&box, &B->pbox[p]);
        for (int a = 0; a < 3; a++) {
            float c = B->pcent[3 * p + a];
            cbox.mn[a] = minf_(cbox.mn[a], c);
            cbox.mx[a] = maxf_(cbox.mx[a], c);
        }
    }
    *out_box = box;
    if (depth > B->max_depth) B->max_depth = depth;
    if (count <= 1 || depth >= MAX_DEPTH) return leaf_ref(first, count);

    /* --- best binned split ------------------------------------------------------------------- */
    float best_cost = INFINITY;
    int best_axis = -1, best_plane = 0;
    float best_scale = 0.0f;
    for (int a = 0; a < 3; a++) {
        float ext = cbox.mx[a] - cbox.mn[a];
        if (!(ext > 0.0f)) continue;
        float scale = (float)N_BINS / ext;
        aabb bbox[N_BINS];
        uint32_t bcnt[N_BINS];
        for (int b = 0; b < N_BINS; b++) { aabb_empty(&bbox[b]); bcnt[b] = 0; }
        for (uint32_t i = first; i < first + count; i++) {
            uint32_t p = B->order[i];
            int b = bin_of((B->pcent[3 * p + a] - cbox.mn[a]) * scale);
            bcnt[b]++;
            aabb_grow(&bbox[b], &B->pbox[p]);
        }
        /* suffix boxes */
        aabb rbox[N_BINS];
        uint32_t rcnt[N_BINS];
        aabb acc;
        aabb_empty(&acc);
        uint32_t n = 0;
        for (int b = N_BINS - 1; b >= 1; b--) {
            aabb_grow(&acc, &bbox[b]);
            n += bcnt[b];
            rbox[b] = acc;
            rcnt[b] = n;
        }
        aabb_empty(&acc);
        n = 0;
        for (int s = 1; s < N_BINS; s++) { /* plane between bin s-1 and bin s */
            aabb_grow(&acc, &bbox[s - 1]);
            n += bcnt[s - 1];
            if (n == 0 || rcnt[s] == 0) continue;
            float cost = aabb_half_area(&acc) * (float)n + aabb_half_area(&rbox[s]) * (float)rcnt[s];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = s; best_scale = scale; }
        }
    }

    float area = aabb_half_area(&box);
    if (count <= LEAF_MAX) {
        /* leaf unless splitting is cheaper: C_TRAV*A + best < count*A */
        if (best_axis < 0 || !(C_TRAV * area + best_cost < (float)count * area)) return leaf_ref(first, count);
    }

    uint32_t n_left = 0;
    if (best_axis >= 0) {
        /* stable partition by bin < plane */
        uint32_t nl = 0, nr = 0;
        for (uint32_t i = first; i < first + count; i++) {
            uint32_t p = B->order[i];
            int b = bin_of((B->pcent[3 * p + best_axis] - cbox.mn[best_axis]) * best_scale);
            if (b < best_plane) B->order[first + nl++] = p; /* nl <= i - first: never overtakes the read */
            else B->tmp[nr++] = p;
        }
        memcpy(&B->order[first + nl], B->tmp, nr * sizeof(uint32_t));
        n_left = nl;
        /* depth bound: each side must still fit below MAX_DEPTH with median splits */
        uint64_t cap = (uint64_t)LEAF_MAX << (MAX_DEPTH - depth - 1);
        uint32_t big = nl > nr ? nl : nr;
        if ((uint64_t)big > cap) n_left = 0; /* order stays partitioned (still deterministic); median below */
    }
    if (n_left == 0) n_left = count / 2;

    uint32_t me = B->n_nodes++;
    aabb lb, rb;
    int32_t l = build_range(B, first, n_left, depth + 1, &lb);
    int32_t r = build_range(B, first + n_left, count - n_left, depth + 1, &rb);
    oracle_node* N = &B->nodes[me];
    N->lx0 = lb.mn[0]; N->lx1 = lb.mx[0]; N->ly0 = lb.mn[1]; N->ly1 = lb.mx[1]; N->lz0 = lb.mn[2]; N->lz1 = lb.mx[2];
    N->rx0 = rb.mn[0]; N->rx1 = rb.mx[0]; N->ry0 = rb.mn[1]; N->ry1 = rb.mx[1]; N->rz0 = rb.mn[2]; N->rz1 = rb.mx[2];
    N->left = l; N->right = r; N->pad0 = 0; N->pad1 = 0;
    return (int32_t)me;
}

/* ------------------------------------------------------------------------------------------------
 * BVH4: collapse of the binary tree.  A wide node starts with the two children of a binary node; while it has fewer than
 * four slots, the inner slot with the largest half-area (first one on ties) is replaced in place by its two children
 * (left at its position, right right after it).  Wide nodes are numbered in DFS pre-order, children in slot order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int32_t ref; aabb box; } slot4;

static void child_slots(const oracle_node* N, slot4* l, slot4* r)
{
    l->ref = N->left; r->ref = N->right;
    l->box.mn[0] = N->lx0; l->box.mx[0] = N->lx1; l->box.mn[1] = N->ly0; l->box.mx[1] = N->ly1; l->box.mn[2] = N->lz0; l->box.mx[2] = N->lz1;
    r->box.mn[0] = N->rx0; r->box.mx[0] = N->rx1; r->box.mn[1] = N->ry0; r->box.mx[1] = N->ry1; r->box.mn[2] = N->rz0; r->box.mx[2] = N->rz1;
}

typedef struct { const oracle_node* bin; oracle_node4* wide; uint32_t n_wide; uint32_t depth; } collapser;

static int32_t collapse_node(collapser* C, int32_t b, uint32_t depth)
{
    slot4 sl[4];
    int n = 2;
    child_slots(&C->bin[b], &sl[0], &sl[1]);
    while (n < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int i = 0; i < n; i++)
            if (sl[i].ref >= 0) {
                float a = aabb_half_area(&sl[i].box);
                if (a > best_area) { best_area = a; best = i; }
            }
        if (best < 0) break;
        slot4 l, r;
        child_slots(&C->bin[sl[best].ref], &l, &r);
        for (int i = n; i > best + 1; i--) sl[i] = sl[i - 1];
        sl[best] = l;
        sl[best + 1] = r;
        n++;
    }
    const uint32_t me = C->n_wide++;
    if (depth + 1 > C->depth) C->depth = depth + 1;
    oracle_node4 W;
    memset(&W, 0, sizeof(W));
    for (int i = 0; i < 4; i++) {
        if (i < n) {
            W.minx[i] = sl[i].box.mn[0]; W.maxx[i] = sl[i].box.mx[0];
            W.miny[i] = sl[i].box.mn[1]; W.maxy[i] = sl[i].box.mx[1];
            W.minz[i] = sl[i].box.mn[2]; W.maxz[i] = sl[i].box.mx[2];
            W.ref[i] = sl[i].ref; /* binary index for now */
        } else { /* unused slot: inverted box (+inf, -inf) that no ray can enter */
            W.minx[i] = W.miny[i] = W.minz[i] = INFINITY;
            W.maxx[i] = W.maxy[i] = W.maxz[i] = -INFINITY;
            W.ref[i] = ORACLE_EMPTY;
        }
    }
    C->wide[me] = W;
    for (int i = 0; i < n; i++)
        if (sl[i].ref >= 0) C->wide[me].ref[i] = collapse_node(C, sl[i].ref, depth + 1);
    return (int32_t)me;
}

/* Quantised nodes: the rule of csrc/bvh_build.cpp quantizeNode4, restated.  Per axis: lo / hi over the children whose box
 * is finite and ordered there; quantum s = (hi - lo) / 255 nudged up so that fma(255, s, lo) >= hi; child planes = largest
 * q with fma(q, s, lo) <= min and smallest q with fma(q, s, lo) >= max (checked with the decode expression itself). */
static inline float decode_plane(uint32_t q, float s, float lo) { return fmaf((float)q, s, lo); }

static void quantize_node4(const oracle_node4* W, oracle_node4q* Q)
{
    const float* mins[3] = { W->minx, W->miny, W->minz };
    const float* maxs[3] = { W->maxx, W->maxy, W->maxz };
    uint32_t qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 };
    for (int a = 0; a < 3; a++) {
        float lo = INFINITY, hi = -INFINITY;
        int valid[4];
        for (int k = 0; k < 4; k++) {
            const float mn = mins[a][k], mx = maxs[a][k];
            valid[k] = W->ref[k] != ORACLE_EMPTY && isfinite(mn) && isfinite(mx) && mn <= mx;
            if (valid[k]) {
                lo = mn < lo ? mn : lo;
                hi = mx > hi ? mx : hi;
            }
        }
        if (!(lo <= hi)) lo = hi = 0.0f;
        float ext = hi - lo;
        if (!(ext < 3.0e38f)) ext = 3.0e38f;
        float sc = (ext * (1.0f / 255.0f)) * 1.000001f;
        if (!(sc >= FLT_MIN)) sc = FLT_MIN;
        Q->lo[a] = lo;
        Q->s[a] = sc;
        for (int k = 0; k < 4; k++) {
            uint32_t l = 0, h = 255;
            if (W->ref[k] == ORACLE_EMPTY) { /* a point at the node's minimum corner: see crt_oracle.h */
                l = 0;
                h = 0;
            } else if (valid[k]) {
                const float fl = (mins[a][k] - lo) / sc, fh = (maxs[a][k] - lo) / sc;
                l = fl >= 255.0f ? 255u : (fl > 0.0f ? (uint32_t)fl : 0u);
                while (l > 0 && decode_plane(l, sc, lo) > mins[a][k]) l--;
                h = fh >= 255.0f ? 255u : (fh > 0.0f ? (uint32_t)fh : 0u);
                while (h < 255 && decode_plane(h, sc, lo) < maxs[a][k]) h++;
            }
            qlo[a] |= l << (8 * k);
            qhi[a] |= h << (8 * k);
        }
    }
    Q->qlo_x = qlo[0]; Q->qhi_x = qhi[0];
    Q->qlo_y = qlo[1]; Q->qhi_y = qhi[1];
    Q->qlo_z = qlo[2]; Q->qhi_z = qhi[2];
    for (int k = 0; k < 4; k++) Q->ref[k] = W->ref[k];
}

static void build_wide(oracle_scene* s)
{
    free(s->nodes4);
    free(s->nodes4q);
    s->nodes4q = NULL;
    s->nodes4 = (oracle_node4*)calloc(s->n_nodes ? s->n_nodes : 1, sizeof(oracle_node4));
    s->n_nodes4 = 0;
    s->depth4 = 0;
    if (s->n_nodes == 0) return;
    collapser C = { s->nodes, s->nodes4, 0, 0 };
    collapse_node(&C, 0, 0);
    s->n_nodes4 = C.n_wide;
    s->depth4 = C.depth;
    s->nodes4q = (oracle_node4q*)calloc(s->n_nodes4 ? s->n_nodes4 : 1, sizeof(oracle_node4q));
    for (uint32_t i = 0; i < s->n_nodes4; i++) quantize_node4(&s->nodes4[i], &s->nodes4q[i]);
}

/* ------------------------------------------------------------------------------------------------
 * LBVH (build mode 1): spec of the GPU builder.  Keys = 30-bit Morton code of the quantised box centroid (10 bits per
 * axis over the centroid bounds) << 32 | input ordinal (unique, so the order is total); sorted ascending; hierarchy of
 * Karras 2012 ("Maximizing parallelism in the construction of BVHs, octrees, and k-d trees") on the common-prefix
 * length of the 64-bit keys; an internal node whose range holds <= LBVH_LEAF_MAX (2) triangles becomes a leaf (4, the SAH builder's
 * leaf size, measured +94 % triangle tests against the SAH tree on the 1M-triangle frame; 2: -13 %); Morton cells are cubic; boxes are exact
 * unions; kept internal nodes are numbered by ascending Karras index (the root is index 0).
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t expand_bits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t quant10(float f) { return f >= 0.0f ? (f < 1024.0f ? (uint32_t)f : 1023u) : 0u; }
static int cmp_u64(const void* a, const void* b)
{
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static inline int delta64(const uint64_t* keys, int64_t n, int64_t i, int64_t j)
{
    if (j < 0 || j >= n) return -1;
    return __builtin_clzll(keys[i] ^ keys[j]);
}

typedef struct { int32_t left, right; /* >= 0 internal index, < 0: ~leaf position */ uint32_t lo, hi; } karras_node;

static void lbvh_box(const karras_node* K, const aabb* sbox, int32_t ref, aabb* out, uint32_t depth, uint32_t* max_depth)
{
    if (ref < 0) { *out = sbox[~ref]; if (depth > *max_depth) *max_depth = depth; return; }
    aabb l, r;
    lbvh_box(K, sbox, K[ref].left, &l, depth + 1, max_depth);
    lbvh_box(K, sbox, K[ref].right, &r, depth + 1, max_depth);
    *out = l;
    aabb_grow(out, &r);
}

/* fills order[] (sorted triangle permutation) and the binary nodes; returns node count */
static uint32_t build_lbvh(const aabb* pbox, const float* pcent, uint32_t n, uint32_t* order, oracle_node* nodes, uint32_t* max_depth)
{
    aabb cb;
    aabb_empty(&cb);
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { cb.mn[a] = minf_(cb.mn[a], pcent[3 * i + a]); cb.mx[a] = maxf_(cb.mx[a], pcent[3 * i + a]); }
    float scale[3];
    /* one cell size for all three axes (that of the longest extent): cubic cells; per-axis scaling cuts a flat scene into thin slabs */
    float extm = 0.0f;
    for (int a = 0; a < 3; a++) { float ext = cb.mx[a] - cb.mn[a]; if (ext > extm) extm = ext; }
    for (int a = 0; a < 3; a++) scale[a] = extm > 0.0f ? 1024.0f / extm : 0.0f;
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int a = 0; a < 3; a++) q[a] = quant10((pcent[3 * i + a] - cb.mn[a]) * scale[a]);
        uint32_t code = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
        keys[i] = ((uint64_t)code << 32) | i;
    }
    qsort(keys, n, sizeof(uint64_t), cmp_u64);
    for (uint32_t i = 0; i < n; i++) order[i] = (uint32_t)(keys[i] & 0xFFFFFFFFu);
    aabb* sbox = (aabb*)malloc(sizeof(aabb) * n);
    for (uint32_t i = 0; i < n; i++) sbox[i] = pbox[order[i]];
    *max_depth = 0;
    if (n <= LEAF_MAX) { /* one leaf: wrapped by the caller like the SAH case */
        free(keys); free(sbox);
        return 0;
    }
    const int64_t N = n;
    karras_node* K = (karras_node*)malloc(sizeof(karras_node) * (n - 1));
    for (int64_t i = 0; i < N - 1; i++) {
        int d = (delta64(keys, N, i, i + 1) - delta64(keys, N, i, i - 1)) < 0 ? -1 : 1;
        int dmin = delta64(keys, N, i, i - d);
        int64_t lmax = 2;
        while (delta64(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
        int64_t l = 0;
        for (int64_t t = lmax / 2; t >= 1; t /= 2)
            if (delta64(keys, N, i, i + (l + t) * d) > dmin) l += t;
        int64_t j = i + l * d;
        int dnode = delta64(keys, N, i, j);
        int64_t sp = 0;
        for (int64_t t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta64(keys, N, i, i + (sp + t) * d) > dnode) sp += t;
            if (t == 1) break;
        }
        int64_t gamma = i + sp * d + (d < 0 ? -1 : 0);
        int64_t lo = i < j ? i : j, hi = i < j ? j : i;
        K[i].lo = (uint32_t)lo; K[i].hi = (uint32_t)hi;
        K[i].left = (lo == gamma) ? ~(int32_t)gamma : (int32_t)gamma;
        K[i].right = (hi == gamma + 1) ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
    }
    /* kept nodes: ranges of more than LBVH_LEAF_MAX triangles, numbered by ascending index */
    uint32_t* rank = (uint32_t*)malloc(sizeof(uint32_t) * (n - 1));
    uint32_t kept = 0;
    for (uint32_t i = 0; i + 1 < n; i++) { rank[i] = kept; if (K[i].hi - K[i].lo + 1 > LBVH_LEAF_MAX) kept++; }
    for (uint32_t i = 0; i + 1 < n; i++) {
        if (K[i].hi - K[i].lo + 1 <= LBVH_LEAF_MAX) continue;
        oracle_node* Nn = &nodes[rank[i]];
        aabb b[2];
        int32_t ref[2];
        const int32_t ch[2] = { K[i].left, K[i].right };
        for (int c = 0; c < 2; c++) {
            uint32_t dummy = 0;
            lbvh_box(K, sbox, ch[c], &b[c], 0, &dummy);
            if (ch[c] < 0) ref[c] = leaf_ref((uint32_t)~ch[c], 1);
            else {
                uint32_t cnt = K[ch[c]].hi - K[ch[c]].lo + 1;
                ref[c] = cnt <= LBVH_LEAF_MAX ? leaf_ref(K[ch[c]].lo, cnt) : (int32_t)rank[ch[c]];
            }
        }
        Nn->lx0 = b[0].mn[0]; Nn->lx1 = b[0].mx[0]; Nn->ly0 = b[0].mn[1]; Nn->ly1 = b[0].mx[1]; Nn->lz0 = b[0].mn[2]; Nn->lz1 = b[0].mx[2];
        Nn->rx0 = b[1].mn[0]; Nn->rx1 = b[1].mx[0]; Nn->ry0 = b[1].mn[1]; Nn->ry1 = b[1].mx[1]; Nn->rz0 = b[1].mn[2]; Nn->rz1 = b[1].mx[2];
        Nn->left = ref[0]; Nn->right = ref[1]; Nn->pad0 = Nn->pad1 = 0;
    }
    /* depth of the emitted tree = deepest kept node + 1 (leaf level) */
    {
        uint32_t* dep = (uint32_t*)calloc(kept ? kept : 1, sizeof(uint32_t));
        uint32_t deepest = 1;
        for (uint32_t k = 0; k < kept; k++) { /* parents have smaller or larger Karras index: do a stack walk from the root */ (void)k; }
        int32_t* st = (int32_t*)malloc(sizeof(int32_t) * (kept ? 2 * kept : 2));
        uint32_t* sd = (uint32_t*)malloc(sizeof(uint32_t) * (kept ? 2 * kept : 2));
        int top = 0;
        st[top] = 0; sd[top++] = 0;
        while (top) {
            int32_t nidx = st[--top];
            uint32_t d0 = sd[top];
            if (d0 + 1 > deepest) deepest = d0 + 1;
            if (nodes[nidx].left >= 0) { st[top] = nodes[nidx].left; sd[top++] = d0 + 1; }
            if (nodes[nidx].right >= 0) { st[top] = nodes[nidx].right; sd[top++] = d0 + 1; }
        }
        *max_depth = deepest;
        free(dep); free(st); free(sd);
    }
    free(keys); free(sbox); free(K); free(rank);
    return kept;
}

oracle_scene* oracle_scene_create(const oracle_mesh* meshes, uint32_t n_meshes,
                                  const oracle_light* lights, uint32_t n_lights,
                                  const oracle_material* mats, uint32_t n_mats)
{
    return oracle_scene_create_ex(meshes, n_meshes, lights, n_lights, mats, n_mats, 0);
}

oracle_scene* oracle_scene_create_ex(const oracle_mesh* meshes, uint32_t n_meshes,
                                     const oracle_light* lights, uint32_t n_lights,
                                     const oracle_material* mats, uint32_t n_mats, int build_mode)
{
    oracle_scene* s = (oracle_scene*)calloc(1, sizeof(*s));
    if (!s) return NULL;
    uint64_t total = 0;
    for (uint32_t m = 0; m < n_meshes; m++) total += meshes[m].n_triangles;
    if (total >= (1u << 28)) { free(s); return NULL; }
    uint32_t n = (uint32_t)total;
    s->n_tris = n;
    s->n_lights = n_lights;
    s->n_mats = n_mats;
    s->lights = (oracle_light*)malloc(sizeof(oracle_light) * (n_lights ? n_lights : 1));
    s->mats = (oracle_material*)malloc(sizeof(oracle_material) * (n_mats ? n_mats : 1));
    if (n_lights) memcpy(s->lights, lights, sizeof(oracle_light) * n_lights);
    if (n_mats) memcpy(s->mats, mats, sizeof(oracle_material) * n_mats);

    /* flatten in input order: gid = running triangle ordinal over meshes */
    oracle_tri* in_tri = (oracle_tri*)malloc(sizeof(oracle_tri) * (n ? n : 1));
    oracle_shade* in_sh = (oracle_shade*)malloc(sizeof(oracle_shade) * (n ? n : 1));
    int any_uv = 0;
    for (uint32_t m = 0; m < n_meshes; m++) any_uv |= meshes[m].uvs != NULL && meshes[m].n_triangles > 0;
    oracle_uv* in_uv = any_uv ? (oracle_uv*)calloc(n ? n : 1, sizeof(oracle_uv)) : NULL;
    aabb* pbox = (aabb*)malloc(sizeof(aabb) * (n ? n : 1));
    float* pcent = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
    uint32_t g = 0;
    for (uint32_t m = 0; m < n_meshes; m++) {
        const oracle_mesh* M = &meshes[m];
        for (uint32_t t = 0; t < M->n_triangles; t++, g++) {
            const uint32_t i0 = M->idx[3 * t], i1 = M->idx[3 * t + 1], i2 = M->idx[3 * t + 2];
            const float* a = &M->xyz[3 * i0];
            const float* b = &M->xyz[3 * i1];
            const float* c = &M->xyz[3 * i2];
            oracle_tri* T = &in_tri[g];
            for (int k = 0; k < 3; k++) {
                T->v0[k] = a[k];
                T->e1[k] = b[k] - a[k];
                T->e2[k] = c[k] - a[k];
                pbox[g].mn[k] = minf_(minf_(a[k], b[k]), c[k]);
                pbox[g].mx[k] = maxf_(maxf_(a[k], b[k]), c[k]);
                pcent[3 * g + k] = (pbox[g].mn[k] + pbox[g].mx[k]) * 0.5f;
            }
            T->inst = m; T->prim = t; T->gid = g;
            oracle_shade* S = &in_sh[g];
            memset(S, 0, sizeof(*S));
            S->material = (uint32_t)M->material_index;
            if (M->normals) {
                memcpy(S->n0, &M->normals[3 * i0], 12);
                memcpy(S->n1, &M->normals[3 * i1], 12);
                memcpy(S->n2, &M->normals[3 * i2], 12);
            }
            if (in_uv && M->uvs) {
                memcpy(in_uv[g].uv0, &M->uvs[3 * i0], 8);
                memcpy(in_uv[g].uv1, &M->uvs[3 * i1], 8);
                memcpy(in_uv[g].uv2, &M->uvs[3 * i2], 8);
            }
        }
    }

    builder B;
    B.pbox = pbox; B.pcent = pcent;
    B.order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    B.tmp = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    B.nodes = (oracle_node*)calloc(n ? n : 1, sizeof(oracle_node)); /* <= n-1 inner nodes, +1 for tiny scenes */
    B.n_nodes = 0; B.max_depth = 0;
    for (uint32_t i = 0; i < n; i++) B.order[i] = i;

    if (n > 0 && build_mode == 1) {
        B.n_nodes = build_lbvh(pbox, pcent, n, B.order, B.nodes, &B.max_depth);
        if (B.n_nodes == 0) { /* <= LEAF_MAX triangles: one leaf wrapped in a node, like the SAH case */
            aabb rootbox;
            aabb_empty(&rootbox);
            for (uint32_t i = 0; i < n; i++) aabb_grow(&rootbox, &pbox[i]);
            oracle_node* N = &B.nodes[0];
            B.n_nodes = 1;
            N->lx0 = N->rx0 = rootbox.mn[0]; N->lx1 = N->rx1 = rootbox.mx[0];
            N->ly0 = N->ry0 = rootbox.mn[1]; N->ly1 = N->ry1 = rootbox.mx[1];
            N->lz0 = N->rz0 = rootbox.mn[2]; N->lz1 = N->rz1 = rootbox.mx[2];
            N->left = leaf_ref(0, n); N->right = leaf_ref(0, 0); N->pad0 = N->pad1 = 0;
            B.max_depth = 1;
        }
    } else if (n > 0) {
        aabb rootbox;
        int32_t root = build_range(&B, 0, n, 0, &rootbox);
        if (root < 0) {
            /* whole scene is one leaf: wrap it so that node 0 exists; right child = empty leaf, same box */
            oracle_node* N = &B.nodes[0];
            B.n_nodes = 1;
            N->lx0 = N->rx0 = rootbox.mn[0]; N->lx1 = N->rx1 = rootbox.mx[0];
            N->ly0 = N->ry0 = rootbox.mn[1]; N->ly1 = N->ry1 = rootbox.mx[1];
            N->lz0 = N->rz0 = rootbox.mn[2]; N->lz1 = N->rz1 = rootbox.mx[2];
            N->left = root; N->right = leaf_ref(0, 0); N->pad0 = N->pad1 = 0;
        }
    }
    s->nodes = B.nodes;
    s->n_nodes = B.n_nodes;
    s->max_depth = B.max_depth;
    s->width = 4;
    build_wide(s);
    s->tris = (oracle_tri*)malloc(sizeof(oracle_tri) * (n ? n : 1));
    s->shade = (oracle_shade*)malloc(sizeof(oracle_shade) * (n ? n : 1));
    for (uint32_t i = 0; i < n; i++) { s->tris[i] = in_tri[B.order[i]]; s->shade[i] = in_sh[B.order[i]]; }
    if (in_uv) {
        s->uvs = (oracle_uv*)malloc(sizeof(oracle_uv) * (n ? n : 1));
        for (uint32_t i = 0; i < n; i++) s->uvs[i] = in_uv[B.order[i]];
        free(in_uv);
    }
    free(B.order); free(B.tmp); free(in_tri); free(in_sh); free(pbox); free(pcent);
    return s;
}

void oracle_scene_destroy(oracle_scene* s)
{
    if (!s) return;
    free(s->nodes); free(s->nodes4); free(s->nodes4q); free(s->tris); free(s->shade); free(s->lights); free(s->mats); free(s->uvs);
    for (uint32_t i = 0; i < s->n_tex; i++) free((void*)s->tex[i].pixels);
    free(s->tex);
    free(s);
}

int oracle_scene_set_bvh(oracle_scene* s, const oracle_node* nodes, uint32_t n_nodes,
                         const oracle_tri* tris, const oracle_shade* shade, uint32_t n_tris)
{
    if (!s) return 1;
    free(s->nodes); free(s->tris); free(s->shade); free(s->uvs);
    s->uvs = NULL;
    s->nodes = (oracle_node*)malloc(sizeof(oracle_node) * (n_nodes ? n_nodes : 1));
    s->tris = (oracle_tri*)malloc(sizeof(oracle_tri) * (n_tris ? n_tris : 1));
    s->shade = (oracle_shade*)calloc(n_tris ? n_tris : 1, sizeof(oracle_shade));
    memcpy(s->nodes, nodes, sizeof(oracle_node) * n_nodes);
    memcpy(s->tris, tris, sizeof(oracle_tri) * n_tris);
    if (shade) memcpy(s->shade, shade, sizeof(oracle_shade) * n_tris);
    s->n_nodes = n_nodes; s->n_tris = n_tris;
    build_wide(s);
    return 0;
}

int oracle_scene_set_textures(oracle_scene* s, const oracle_texture* tex, uint32_t n)
{
    if (!s || (!tex && n)) return 1;
    for (uint32_t i = 0; i < s->n_tex; i++) free((void*)s->tex[i].pixels);
    free(s->tex);
    s->tex = (oracle_texture*)calloc(n ? n : 1, sizeof(oracle_texture));
    s->n_tex = n;
    for (uint32_t i = 0; i < n; i++) {
        s->tex[i] = tex[i];
        s->tex[i].pixels = NULL;
        if (tex[i].type == 3u && tex[i].pixels && tex[i].width && tex[i].height && tex[i].channels) {
            size_t bytes = (size_t)tex[i].width * tex[i].height * tex[i].channels;
            uint8_t* copy = (uint8_t*)malloc(bytes);
            memcpy(copy, tex[i].pixels, bytes);
            s->tex[i].pixels = copy;
        }
    }
    return 0;
}
const oracle_uv* oracle_scene_uvs(const oracle_scene* s) { return s->uvs; }

/* R/CRTTexture*.cpp getColor, operation for operation.  Bitmaps need >= 3 channels here (the reference reads the green
 * byte of 1-channel images from the next pixel: not reproduced). */
static v3 texture_color(const oracle_texture* t, float u, float v)
{
    const v3 A = v3_make(t->color_a[0], t->color_a[1], t->color_a[2]);
    const v3 Bc = v3_make(t->color_b[0], t->color_b[1], t->color_b[2]);
    if (t->type == 1u) /* R/CRTTextureEdges.cpp:9-15 */
        return (u < t->scalar || v < t->scalar || (1.0f - u - v) < t->scalar) ? A : Bc;
    if (t->type == 2u) { /* R/CRTTextureChecker.cpp:9-20 */
        int width = (int)(1.0f / t->scalar);
        int u2 = (int)floorf(u * (float)width);
        int v2 = (int)floorf(v * (float)width);
        return ((u2 + v2) % 2 == 0) ? A : Bc;
    }
    if (t->type == 3u) { /* R/CRTTextureBitmap.cpp:12-36 */
        if (!t->pixels || t->channels < 3u) return v3_make(0.0f, 0.0f, 0.0f);
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        int row = (int)((1.0f - v) * (float)((int)t->height - 1));
        int col = (int)(u * (float)((int)t->width - 1));
        size_t index = ((size_t)row * t->width + (size_t)col) * t->channels;
        return v3_make((float)t->pixels[index] / 255.0f, (float)t->pixels[index + 1] / 255.0f, (float)t->pixels[index + 2] / 255.0f);
    }
    return A; /* albedo texture, R/CRTTextureAlbedo.cpp */
}
void oracle_texture_color(const oracle_texture* t, float u, float v, float out_rgb[3])
{
    v3 c = texture_color(t, u, v);
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}

uint32_t oracle_scene_node_count(const oracle_scene* s) { return s->n_nodes; }
uint32_t oracle_scene_node4_count(const oracle_scene* s) { return s->n_nodes4; }
const oracle_node4* oracle_scene_nodes4(const oracle_scene* s) { return s->nodes4; }
const oracle_node4q* oracle_scene_nodes4q(const oracle_scene* s) { return s->nodes4q; }
uint32_t oracle_scene_depth4(const oracle_scene* s) { return s->depth4; }
void oracle_scene_set_width(oracle_scene* s, int width) { s->width = width == 2 ? 2 : 4; }
uint32_t oracle_scene_tri_count(const oracle_scene* s) { return s->n_tris; }
const oracle_node* oracle_scene_nodes(const oracle_scene* s) { return s->nodes; }
const oracle_tri* oracle_scene_tris(const oracle_scene* s) { return s->tris; }
const oracle_shade* oracle_scene_shade(const oracle_scene* s) { return s->shade; }
uint32_t oracle_scene_max_depth(const oracle_scene* s) { return s->max_depth; }

/* ------------------------------------------------------------------------------------------------
 * rays
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    v3 o, d;
    v3 idir, noid; /* 1/d (clamped) and -(o * idir) for the fma slab test */
} ray;

static inline float safe_rcp_dir(float d)
{
    float ds = (fabsf(d) < DIR_EPS) ? copysignf(DIR_EPS, d) : d;
    return 1.0f / ds;
}

static inline void ray_setup(ray* r, v3 o, v3 d)
{
    r->o = o; r->d = d;
    r->idir = v3_make(safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z));
    r->noid = v3_make(-(o.x * r->idir.x), -(o.y * r->idir.y), -(o.z * r->idir.z));
}

typedef struct { float t, u, v; uint32_t tri; /* leaf-order index */ uint32_t gid; int hit; } hit_rec;

typedef struct { uint64_t nodes, tris; } trav_count;

/* Moeller-Trumbore, two sided (no culling: R/DXRTRenderer.cpp:590,697-699 set no cull flags).
 * u = weight of v1, v = weight of v2 (DXR barycentrics.x / .y, hlsl:127-131).  Rejections are written
 * as positive-form compares so that NaN/inf (det == 0) reject without a separate test. */
static inline int tri_test(const ray* r, const oracle_tri* T, float tmin, float* t, float* u, float* v)
{
    v3 e1 = v3_make(T->e1[0], T->e1[1], T->e1[2]);
    v3 e2 = v3_make(T->e2[0], T->e2[1], T->e2[2]);
    v3 p = v3_cross(r->d, e2);
    float det = v3_dot(e1, p);
    float inv = 1.0f / det;
    v3 s = v3_sub(r->o, v3_make(T->v0[0], T->v0[1], T->v0[2]));
    float uu = v3_dot(s, p) * inv;
    v3 q = v3_cross(s, e1);
    float vv = v3_dot(r->d, q) * inv;
    float tt = v3_dot(e2, q) * inv;
    *t = tt; *u = uu; *v = vv;
    return (uu >= 0.0f) & (vv >= 0.0f) & (uu + vv <= 1.0f) & (tt > tmin);
}

static inline int box_test(float x0, float x1, float y0, float y1, float z0, float z1, const ray* r,
                           float tmin, float tmax, float* tnear)
{
    float ax = fmaf(x0, r->idir.x, r->noid.x), bx = fmaf(x1, r->idir.x, r->noid.x);
    float ay = fmaf(y0, r->idir.y, r->noid.y), by = fmaf(y1, r->idir.y, r->noid.y);
    float az = fmaf(z0, r->idir.z, r->noid.z), bz = fmaf(z1, r->idir.z, r->noid.z);
    float tn = maxf_(maxf_(minf_(ax, bx), minf_(ay, by)), maxf_(minf_(az, bz), tmin));
    float tf = minf_(minf_(maxf_(ax, bx), maxf_(ay, by)), minf_(maxf_(az, bz), tmax));
    *tnear = tn;
    return tn <= tf;
}

/* closest hit in (tmin, tmax); equal t resolved towards the lower global triangle ordinal.
 * Boxes are culled against tcull = best_t * CULL_PAD, not best_t: the slab distances and the Moeller-Trumbore t
 * round differently, and without the pad a box holding an equal-t (or one-ulp-closer) triangle on a shared edge
 * can be culled, making the winner depend on traversal order. tcull changes only when a hit is accepted. */
static void trace_closest2(const oracle_scene* s, const ray* r, float tmin, float tmax, hit_rec* h, trav_count* c)
{
    h->t = tmax; h->u = 0.0f; h->v = 0.0f; h->tri = 0; h->gid = 0; h->hit = 0;
    if (s->n_nodes == 0) return;
    int32_t stack[MAX_DEPTH + 1];
    int sp = 0;
    int32_t cur = 0;
    float tcull = tmax * CULL_PAD;
    for (;;) {
        if (cur >= 0) {
            const oracle_node* N = &s->nodes[cur];
            c->nodes++;
            float tnl, tnr;
            int hl = box_test(N->lx0, N->lx1, N->ly0, N->ly1, N->lz0, N->lz1, r, tmin, tcull, &tnl);
            int hr = box_test(N->rx0, N->rx1, N->ry0, N->ry1, N->rz0, N->rz1, r, tmin, tcull, &tnr);
            if (hl & hr) {
                int right_first = tnr < tnl;
                stack[sp++] = right_first ? N->left : N->right;
                if (sp > t_max_sp) t_max_sp = sp;
                cur = right_first ? N->right : N->left;
                continue;
            }
            if (hl) { cur = N->left; continue; }
            if (hr) { cur = N->right; continue; }
        } else {
            uint32_t code = (uint32_t)~cur;
            uint32_t first = code >> 3, cnt = code & 7u;
            for (uint32_t i = first; i < first + cnt; i++) {
                const oracle_tri* T = &s->tris[i];
                float t, u, v;
                c->tris++;
                if (tri_test(r, T, tmin, &t, &u, &v)) {
                    if ((t < h->t) | ((t == h->t) & (T->gid < h->gid))) {
                        h->t = t; h->u = u; h->v = v; h->tri = i; h->gid = T->gid; h->hit = 1;
                        tcull = t * CULL_PAD;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}

/* any hit in (tmin, tmax): order independent */
static int trace_any2(const oracle_scene* s, const ray* r, float tmin, float tmax, trav_count* c)
{
    if (s->n_nodes == 0) return 0;
    int32_t stack[MAX_DEPTH + 1];
    int sp = 0;
    int32_t cur = 0;
    const float tcull = tmax * CULL_PAD;
    for (;;) {
        if (cur >= 0) {
            const oracle_node* N = &s->nodes[cur];
            c->nodes++;
            float tnl, tnr;
            int hl = box_test(N->lx0, N->lx1, N->ly0, N->ly1, N->lz0, N->lz1, r, tmin, tcull, &tnl);
            int hr = box_test(N->rx0, N->rx1, N->ry0, N->ry1, N->rz0, N->rz1, r, tmin, tcull, &tnr);
            if (hl & hr) {
                int right_first = tnr < tnl;
                stack[sp++] = right_first ? N->left : N->right;
                if (sp > t_max_sp) t_max_sp = sp;
                cur = right_first ? N->right : N->left;
                continue;
            }
            if (hl) { cur = N->left; continue; }
            if (hr) { cur = N->right; continue; }
        } else {
            uint32_t code = (uint32_t)~cur;
            uint32_t first = code >> 3, cnt = code & 7u;
            for (uint32_t i = first; i < first + cnt; i++) {
                float t, u, v;
                c->tris++;
                if (tri_test(r, &s->tris[i], tmin, &t, &u, &v) & (t < tmax)) return 1;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return 0;
}

/* ---- wide (BVH4) traversal: what the HIP kernels do.  Per step: four slab tests; hit children are ordered by the key
 * (bits(t_near) & 0x7FFFFFFC) | slot (t_near >= 0, so its bit pattern orders like the float; the two low bits carry the
 * slot and make keys unique); nearest becomes current, the others are pushed farthest first. */
#define STACK4 (3 * 32 + 4)

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* the four slab tests of a quantised wide node in one SSE pass: lane k = child k.  Planes are decoded inside the test:
 * t(q) = fma(q, s * idir, fma(lo, idir, -o * idir)), the generic min/max form (the kernel's octant-specialised form picks
 * the same members: t is monotonic in q).  _mm_min_ps(a, b) is exactly minf_(a, b) = a < b ? a : b (and max likewise),
 * _mm_fmadd_ps is fmaf per lane.  Returns the hit mask. */
static inline __m128 q_bytes(uint32_t w) { return _mm_cvtepi32_ps(_mm_cvtepu8_epi32(_mm_cvtsi32_si128((int)w))); }

static inline int slab4(const oracle_node4q* N, const ray* r, float tmin, float tcull, float tn_out[4])
{
    const __m128 ax = _mm_set1_ps(N->s[0] * r->idir.x), ay = _mm_set1_ps(N->s[1] * r->idir.y), az = _mm_set1_ps(N->s[2] * r->idir.z);
    const __m128 bx = _mm_set1_ps(fmaf(N->lo[0], r->idir.x, r->noid.x)), by = _mm_set1_ps(fmaf(N->lo[1], r->idir.y, r->noid.y)),
                 bz = _mm_set1_ps(fmaf(N->lo[2], r->idir.z, r->noid.z));
    const __m128 x0 = _mm_fmadd_ps(q_bytes(N->qlo_x), ax, bx), x1 = _mm_fmadd_ps(q_bytes(N->qhi_x), ax, bx);
    const __m128 y0 = _mm_fmadd_ps(q_bytes(N->qlo_y), ay, by), y1 = _mm_fmadd_ps(q_bytes(N->qhi_y), ay, by);
    const __m128 z0 = _mm_fmadd_ps(q_bytes(N->qlo_z), az, bz), z1 = _mm_fmadd_ps(q_bytes(N->qhi_z), az, bz);
    const __m128 tn = _mm_max_ps(_mm_max_ps(_mm_min_ps(x0, x1), _mm_min_ps(y0, y1)), _mm_max_ps(_mm_min_ps(z0, z1), _mm_set1_ps(tmin)));
    const __m128 tf = _mm_min_ps(_mm_min_ps(_mm_max_ps(x0, x1), _mm_max_ps(y0, y1)), _mm_min_ps(_mm_max_ps(z0, z1), _mm_set1_ps(tcull)));
    _mm_storeu_ps(tn_out, tn);
    return _mm_movemask_ps(_mm_cmple_ps(tn, tf)); /* unused slots are point boxes: no test of their own */
}

static inline int wide_step(const oracle_node4q* N, const ray* r, float tmin, float tcull, uint32_t key[4])
{
    float tn[4];
    const int mask = slab4(N, r, tmin, tcull, tn);
    if (mask == 0) return 0;
    if ((mask & (mask - 1)) == 0) { /* one child hit: nothing to order (same outcome as the network below) */
        key[0] = (uint32_t)__builtin_ctz((unsigned)mask);
        return 1;
    }
    int n_hit = 0;
    for (int k = 0; k < 4; k++) {
        const int hit = (mask >> k) & 1;
        key[k] = hit ? ((f2u(tn[k]) & 0x7FFFFFFCu) | (uint32_t)k) : 0xFFFFFFFFu;
        n_hit += hit;
    }
    /* sorting network (0,1)(2,3)(0,2)(1,3)(1,2), ascending */
#define CSWAP(a, b) { uint32_t lo = key[a] < key[b] ? key[a] : key[b]; uint32_t hi = key[a] < key[b] ? key[b] : key[a]; key[a] = lo; key[b] = hi; }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
    return n_hit;
}

static void trace_closest4(const oracle_scene* s, const ray* r, float tmin, float tmax, hit_rec* h, trav_count* c)
{
    h->t = tmax; h->u = 0.0f; h->v = 0.0f; h->tri = 0; h->gid = 0; h->hit = 0;
    if (s->n_nodes4 == 0) return;
    int32_t stack[STACK4];
    int sp = 0;
    int32_t cur = 0;
    float tcull = tmax * CULL_PAD;
    for (;;) {
        if (cur >= 0) {
            const oracle_node4q* N = &s->nodes4q[cur];
            uint32_t key[4];
            c->nodes++;
            int n_hit = wide_step(N, r, tmin, tcull, key);
            if (n_hit > 0) {
                for (int k = n_hit - 1; k >= 1; k--) stack[sp++] = N->ref[key[k] & 3u];
                if (sp > t_max_sp) t_max_sp = sp;
                cur = N->ref[key[0] & 3u];
                continue;
            }
        } else {
            uint32_t code = (uint32_t)~cur;
            uint32_t first = code >> 3, cnt = code & 7u;
            for (uint32_t i = first; i < first + cnt; i++) {
                const oracle_tri* T = &s->tris[i];
                float t, u, v;
                c->tris++;
                if (tri_test(r, T, tmin, &t, &u, &v)) {
                    if ((t < h->t) | ((t == h->t) & (T->gid < h->gid))) {
                        h->t = t; h->u = u; h->v = v; h->tri = i; h->gid = T->gid; h->hit = 1;
                        tcull = t * CULL_PAD;
                    }
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
}

/* any hit: order independent, children taken in slot order (first hit slot next, the others pushed last slot first) */
static int trace_any4(const oracle_scene* s, const ray* r, float tmin, float tmax, trav_count* c)
{
    if (s->n_nodes4 == 0) return 0;
    int32_t stack[STACK4];
    int sp = 0;
    int32_t cur = 0;
    const float tcull = tmax * CULL_PAD;
    for (;;) {
        if (cur >= 0) {
            const oracle_node4q* N = &s->nodes4q[cur];
            c->nodes++;
            int first_hit = -1;
            int hits[4];
            float tn4[4];
            const int mask = slab4(N, r, tmin, tcull, tn4);
            for (int k = 0; k < 4; k++) {
                hits[k] = (mask >> k) & 1;
                if (hits[k] && first_hit < 0) first_hit = k;
            }
            if (first_hit >= 0) {
                for (int k = 3; k > first_hit; k--)
                    if (hits[k]) stack[sp++] = N->ref[k];
                if (sp > t_max_sp) t_max_sp = sp;
                cur = N->ref[first_hit];
                continue;
            }
        } else {
            uint32_t code = (uint32_t)~cur;
            uint32_t first = code >> 3, cnt = code & 7u;
            for (uint32_t i = first; i < first + cnt; i++) {
                float t, u, v;
                c->tris++;
                if (tri_test(r, &s->tris[i], tmin, &t, &u, &v) & (t < tmax)) return 1;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    return 0;
}

static void trace_closest(const oracle_scene* s, const ray* r, float tmin, float tmax, hit_rec* h, trav_count* c)
{
    if (s->width == 2) trace_closest2(s, r, tmin, tmax, h, c);
    else trace_closest4(s, r, tmin, tmax, h, c);
}

static int trace_any(const oracle_scene* s, const ray* r, float tmin, float tmax, trav_count* c)
{
    return s->width == 2 ? trace_any2(s, r, tmin, tmax, c) : trace_any4(s, r, tmin, tmax, c);
}

static void brute_closest(const oracle_scene* s, const ray* r, float tmin, float tmax, hit_rec* h, trav_count* c)
{
    h->t = tmax; h->u = 0.0f; h->v = 0.0f; h->tri = 0; h->gid = 0; h->hit = 0;
    for (uint32_t i = 0; i < s->n_tris; i++) {
        const oracle_tri* T = &s->tris[i];
        float t, u, v;
        c->tris++;
        if (tri_test(r, T, tmin, &t, &u, &v)) {
            if ((t < h->t) | ((t == h->t) & (T->gid < h->gid))) {
                h->t = t; h->u = u; h->v = v; h->tri = i; h->gid = T->gid; h->hit = 1;
            }
        }
    }
}

static int brute_any(const oracle_scene* s, const ray* r, float tmin, float tmax, trav_count* c)
{
    for (uint32_t i = 0; i < s->n_tris; i++) {
        float t, u, v;
        c->tris++;
        if (tri_test(r, &s->tris[i], tmin, &t, &u, &v) & (t < tmax)) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * rayGen (hlsl:21-55).  The reference hard-codes width=1920,height=1080 (hlsl:24-25); here they are
 * parameters, used in the same float expressions.
 * ---------------------------------------------------------------------------------------------- */
static inline v3 ray_dir_j(const float rot[9], uint32_t px, uint32_t py, float jx, float jy, float width, float height)
{
    float x = (float)px, y = (float)py;
    x += jx; y += jy;                /* hlsl:35-36 with 0.5, 0.5; the path tracer jitters inside the pixel */
    x /= width; y /= height;         /* hlsl:38-39 */
    x = (2.0f * x) - 1.0f;           /* hlsl:41 */
    y = 1.0f - (2.0f * y);           /* hlsl:42 */
    x *= width / height;             /* hlsl:44 */
    v3 dc = v3_normalize(v3_make(x, y, -1.0f));                       /* hlsl:46 */
    /* mul(cameraRotation, v): column vector, out_i = sum_j M[i][j] v_j   (hlsl:47, cpp:259-264) */
    v3 dw = v3_make(v3_dot(v3_make(rot[0], rot[1], rot[2]), dc),
                    v3_dot(v3_make(rot[3], rot[4], rot[5]), dc),
                    v3_dot(v3_make(rot[6], rot[7], rot[8]), dc));
    return v3_normalize(dw);
}

static inline v3 ray_dir(const float rot[9], uint32_t px, uint32_t py, float width, float height)
{
    return ray_dir_j(rot, px, py, 0.5f, 0.5f, width, height);
}

void oracle_ray_dir(const float rot[9], uint32_t px, uint32_t py, uint32_t w, uint32_t h, float out_dir[3])
{
    v3 d = ray_dir(rot, px, py, (float)w, (float)h);
    out_dir[0] = d.x; out_dir[1] = d.y; out_dir[2] = d.z;
}

/* ------------------------------------------------------------------------------------------------
 * closestHit colour functions (hlsl:78-169)
 * ---------------------------------------------------------------------------------------------- */
static inline float hash_sin(float x, float k) { return fracf_(oracle_sinf(x) * k); }

static inline v3 object_base_colour(uint32_t inst) /* hlsl:97-101, 117-121 */
{
    float f = (float)inst;
    return v3_make(hash_sin(f * 12.9898f, 43758.5453f), hash_sin(f * 78.233f, 12345.6789f),
                   hash_sin(f * 39.425f, 34567.8901f));
}

static v3 shade_debug(uint32_t mode, uint32_t inst, uint32_t prim, float t, float u, float v, v3 o, v3 d)
{
    v3 wp = v3_make(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t); /* WorldRayOrigin + WorldRayDirection * RayTCurrent */
    if (mode == 0) { /* hlsl:84-91 */
        float f = (float)prim;
        return v3_make(hash_sin(f * 12.9898f, 43758.5453f), hash_sin(f * 78.233f, 43758.5453f),
                       hash_sin(f * 45.164f, 43758.5453f));
    }
    if (mode == 1) { /* hlsl:93-112 */
        v3 base = object_base_colour(inst);
        int32_t cx = (int32_t)floorf(wp.x / 2.0f), cy = (int32_t)floorf(wp.y / 2.0f), cz = (int32_t)floorf(wp.z / 2.0f);
        uint32_t hash = ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u) ^ ((uint32_t)cz * 83492791u);
        float variation = hash_sin((float)hash * 12.9898f, 43758.5453f);
        return v3_make(lerpf_(base.x * 0.7f, base.x * 1.3f, variation), lerpf_(base.y * 0.7f, base.y * 1.3f, variation),
                       lerpf_(base.z * 0.7f, base.z * 1.3f, variation));
    }
    if (mode == 2) { /* hlsl:113-124 */
        v3 base = object_base_colour(inst);
        float shade = hash_sin((float)prim * 12.9898f, 43758.5453f);
        float k = lerpf_(0.6f, 1.0f, shade);
        return v3_make(base.x * k, base.y * k, base.z * k);
    }
    if (mode == 3) /* hlsl:125-134 */
        return v3_make(1.0f - u - v, u, v);
    if (mode == 4) { /* hlsl:135-147 */
        float h = saturatef_((wp.y + 10.0f) / 20.0f);
        return v3_make(lerpf_(0.1f, 0.9f, h), lerpf_(0.2f, 0.9f, h), lerpf_(0.6f, 0.9f, h));
    }
    if (mode == 5) { /* hlsl:148-154 */
        float c = saturatef_(t * 0.05f);
        return v3_make(c, c, c);
    }
    { /* hlsl:155-166 */
        int32_t checker = ((int32_t)floorf(wp.x) ^ (int32_t)floorf(wp.z)) & 1;
        float c = checker ? 0.9f : 0.2f;
        return v3_make(c, c, c);
    }
}

void oracle_shade_mode(uint32_t mode, uint32_t inst, uint32_t prim, float t, float u, float v,
                       const float o[3], const float d[3], float out_rgb[3])
{
    v3 c = shade_debug(mode, inst, prim, t, u, v, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}

/* any-hit query on an arbitrary ray interval (tests: the segments of split packets, csrc/split_packet.hip.h) */
int oracle_occluded(const oracle_scene* s, const float o[3], const float d[3], float tmin, float tmax, int brute)
{
    ray r;
    ray_setup(&r, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
    trav_count c = { 0, 0 };
    return brute ? brute_any(s, &r, tmin, tmax, &c) : trace_any(s, &r, tmin, tmax, &c);
}

int oracle_intersect_tri(const float o[3], const float d[3], const float v0[3], const float v1[3],
                         const float v2[3], float tmin, float tmax, float* t, float* u, float* v)
{
    ray r;
    oracle_tri T;
    ray_setup(&r, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]));
    for (int k = 0; k < 3; k++) { T.v0[k] = v0[k]; T.e1[k] = v1[k] - v0[k]; T.e2[k] = v2[k] - v0[k]; }
    T.inst = T.prim = T.gid = 0;
    return tri_test(&r, &T, tmin, t, u, v) & (*t < tmax);
}

/* Surface at a closest hit, shared by mode 100 and the path tracer: hit point, shading normal (face normal, or the
 * barycentric blend of the CRTMesh vertex normals for smooth_shading materials), flipped to face the ray, material. */
typedef struct { v3 P, N; v3 albedo; uint32_t mtype; int entering; float ior; } surface;

static void surface_at(const oracle_scene* s, const ray* r, const hit_rec* h, surface* sf)
{
    const oracle_tri* T = &s->tris[h->tri];
    const oracle_shade* S = &s->shade[h->tri];
    sf->P = v3_make(r->o.x + r->d.x * h->t, r->o.y + r->d.y * h->t, r->o.z + r->d.z * h->t);
    sf->albedo = v3_make(1.0f, 1.0f, 1.0f);
    sf->mtype = 1; /* DIFFUSE */
    sf->ior = 1.0f;
    int smooth = 0;
    if (S->material < s->n_mats) {
        const oracle_material* M = &s->mats[S->material];
        sf->albedo = v3_make(M->albedo[0], M->albedo[1], M->albedo[2]);
        smooth = M->smooth != 0;
        sf->mtype = M->type;
        sf->ior = M->ior;
        if (M->texture >= 0 && (uint32_t)M->texture < s->n_tex) {
            /* CRTMaterial::isTexture: the albedo comes from the texture. Edges textures are functions of the hit's
             * barycentrics; the others of the mesh uvs interpolated at the hit (0,0 when the mesh has none). */
            const oracle_texture* tx = &s->tex[M->texture];
            float tu = h->u, tv = h->v;
            if (tx->type != 1u) {
                tu = 0.0f; tv = 0.0f;
                if (s->uvs) {
                    const oracle_uv* U = &s->uvs[h->tri];
                    const float w = 1.0f - h->u - h->v;
                    tu = fmaf(U->uv2[0], h->v, fmaf(U->uv1[0], h->u, U->uv0[0] * w));
                    tv = fmaf(U->uv2[1], h->v, fmaf(U->uv1[1], h->u, U->uv0[1] * w));
                }
            }
            sf->albedo = texture_color(tx, tu, tv);
        }
    }
    v3 N = v3_cross(v3_make(T->e1[0], T->e1[1], T->e1[2]), v3_make(T->e2[0], T->e2[1], T->e2[2]));
    if (smooth) {
        float w = 1.0f - h->u - h->v;
        v3 Ns = v3_make(fmaf(S->n2[0], h->v, fmaf(S->n1[0], h->u, S->n0[0] * w)),
                        fmaf(S->n2[1], h->v, fmaf(S->n1[1], h->u, S->n0[1] * w)),
                        fmaf(S->n2[2], h->v, fmaf(S->n1[2], h->u, S->n0[2] * w)));
        if (v3_dot(Ns, Ns) > 0.0f) N = Ns; /* zero / missing normals (NaN compares false) -> face normal */
    }
    N = v3_normalize(N);
    sf->entering = 1;
    if (v3_dot(N, r->d) > 0.0f) { N = v3_make(-N.x, -N.y, -N.z); sf->entering = 0; } /* two sided */
    sf->N = N;
}

/* direct light at Po with normal N: one shadow ray per light whose cosine is positive (BASELINE.json north_star's
 * "Lambert ... + shadow rays"; lights and materials: R/CRTLight.h:4-16, R/CRTMaterial.h:4-36, parsed but never
 * evaluated by the reference -- SURVEY.md section 8 row a13: NOT IN THE REFERENCE, the build's specification) */
/* Phong specular term of mode 100 (NOT in the reference: CRTMaterial has no specular coefficient or exponent,
 * R/CRTMaterial.h:30-35, so both are renderer options: ks in thousandths, an integer exponent).  x^n by square and multiply
 * in a fixed order, so that both sides round alike. */
static float g_phong_ks = 0.0f;
static uint32_t g_phong_exp = 32;
void oracle_set_phong(uint32_t ks_permille, uint32_t exponent)
{
    g_phong_ks = (float)ks_permille / 1000.0f;
    g_phong_exp = exponent ? exponent : 1;
}
static inline float pow_uint(float x, uint32_t n)
{
    float result = 1.0f, base = x;
    while (n) {
        if (n & 1u) result *= base;
        base *= base;
        n >>= 1;
    }
    return result;
}

/* view = direction from the surface to the eye, or NULL: no specular term (path tracing) */
static v3 direct_light(const oracle_scene* s, v3 Po, v3 N, v3 albedo, const v3* view, int brute, trav_count* c, uint64_t* n_shadow)
{
    v3 rgb = v3_make(0.0f, 0.0f, 0.0f);
    for (uint32_t li = 0; li < s->n_lights; li++) {
        const oracle_light* L = &s->lights[li];
        v3 Lv = v3_sub(v3_make(L->pos[0], L->pos[1], L->pos[2]), Po);
        float r2 = v3_dot(Lv, Lv);
        float dist = sqrtf(r2);
        float invr = 1.0f / dist;
        v3 Ld = v3_make(Lv.x * invr, Lv.y * invr, Lv.z * invr);
        float cosv = fmaxf(0.0f, v3_dot(N, Ld));
        if (cosv > 0.0f) {
            ray sr;
            ray_setup(&sr, Po, Ld);
            (*n_shadow)++;
            int occluded = brute ? brute_any(s, &sr, 0.0f, dist, c) : trace_any(s, &sr, 0.0f, dist, c);
            if (!occluded) {
                float k = (L->intensity / (FOUR_PI * r2)) * cosv;
                rgb.x = fmaf(albedo.x, k, rgb.x);
                rgb.y = fmaf(albedo.y, k, rgb.y);
                rgb.z = fmaf(albedo.z, k, rgb.z);
                if (view && g_phong_ks > 0.0f) { /* white highlight: the light mirrored about N against the eye direction */
                    const float nl2 = 2.0f * v3_dot(N, Ld);
                    const v3 R = v3_make(fmaf(nl2, N.x, -Ld.x), fmaf(nl2, N.y, -Ld.y), fmaf(nl2, N.z, -Ld.z));
                    const float rv = fmaxf(0.0f, v3_dot(R, *view));
                    const float sp = (g_phong_ks * (L->intensity / (FOUR_PI * r2))) * pow_uint(rv, g_phong_exp);
                    rgb.x += sp; rgb.y += sp; rgb.z += sp;
                }
            }
        }
    }
    return rgb;
}

static inline v3 bias_point(v3 P, v3 N, float bias)
{
    return v3_make(fmaf(N.x, bias, P.x), fmaf(N.y, bias, P.y), fmaf(N.z, bias, P.z));
}

/* mode 100: Lambert with every material treated as diffuse */
static v3 shade_lambert(const oracle_scene* s, const ray* r, const hit_rec* h, int brute, trav_count* c, uint64_t* n_shadow)
{
    surface sf;
    surface_at(s, r, h, &sf);
    const v3 view = v3_make(-r->d.x, -r->d.y, -r->d.z);
    return direct_light(s, bias_point(sf.P, sf.N, SHADOW_BIAS), sf.N, sf.albedo, &view, brute, c, n_shadow);
}

/* ------------------------------------------------------------------------------------------------
 * mode 200: path tracing, spp samples per pixel, up to max_bounces bounces (BASELINE.json configs[4]:
 * "4 spp path-traced (3 bounces)").  NOT IN THE REFERENCE (MaxTraceRecursionDepth = 1, one TraceRay per pixel,
 * R/DXRTRenderer.cpp:1172): build-defined, SURVEY.md section 8 row a13.  Counter-based RNG keyed by
 * (pixel, sample, seed) so CPU and GPU draw the same numbers in the same order.
 * ---------------------------------------------------------------------------------------------- */
static uint32_t g_path_spp = 4, g_path_bounces = 3, g_path_seed = 1234;
void oracle_set_path_params(uint32_t spp, uint32_t max_bounces, uint32_t seed)
{
    g_path_spp = spp ? spp : 1; g_path_bounces = max_bounces; g_path_seed = seed;
}

static inline uint32_t pcg_hash(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
static inline float rng_next(uint32_t* st) /* uniform in [0,1), 24 bits */
{
    *st = pcg_hash(*st);
    return (float)(*st >> 8) * 0x1p-24f;
}

static v3 trace_path(const oracle_scene* s, const float rot[9], v3 cam, uint32_t px, uint32_t py, float width, float height,
                     uint32_t pix, uint32_t sample, v3 miss, int brute, trav_count* c, uint64_t* n_closest, uint64_t* n_shadow,
                     hit_rec* first_hit)
{
    uint32_t rng = pcg_hash(pix ^ pcg_hash(sample + pcg_hash(g_path_seed)));
    const float jx = rng_next(&rng), jy = rng_next(&rng);
    ray r;
    ray_setup(&r, cam, ray_dir_j(rot, px, py, jx, jy, width, height));
    v3 L = v3_make(0.0f, 0.0f, 0.0f), thr = v3_make(1.0f, 1.0f, 1.0f);
    float tmin = RAY_TMIN;
    for (uint32_t bounce = 0;; bounce++) {
        hit_rec h;
        (*n_closest)++;
        if (brute) brute_closest(s, &r, tmin, RAY_TMAX, &h, c);
        else trace_closest(s, &r, tmin, RAY_TMAX, &h, c);
        if (bounce == 0 && first_hit) *first_hit = h;
        if (!h.hit) {
            L = v3_make(fmaf(thr.x, miss.x, L.x), fmaf(thr.y, miss.y, L.y), fmaf(thr.z, miss.z, L.z));
            break;
        }
        surface sf;
        surface_at(s, &r, &h, &sf);
        tmin = 0.0f; /* secondary rays start from biased origins */
        if (sf.mtype == 4u) { /* CONSTANT: emits its albedo, ends the path */
            L = v3_make(fmaf(thr.x, sf.albedo.x, L.x), fmaf(thr.y, sf.albedo.y, L.y), fmaf(thr.z, sf.albedo.z, L.z));
            break;
        }
        if (sf.mtype == 2u) { /* REFLECTIVE: perfect mirror */
            if (bounce == g_path_bounces) break;
            float k = 2.0f * v3_dot(r.d, sf.N);
            v3 d = v3_normalize(v3_make(fmaf(-k, sf.N.x, r.d.x), fmaf(-k, sf.N.y, r.d.y), fmaf(-k, sf.N.z, r.d.z)));
            thr = v3_make(thr.x * sf.albedo.x, thr.y * sf.albedo.y, thr.z * sf.albedo.z);
            ray_setup(&r, bias_point(sf.P, sf.N, SHADOW_BIAS), d);
            continue;
        }
        if (sf.mtype == 3u) { /* REFRACTIVE: Snell, total internal reflection when there is no transmitted ray */
            if (bounce == g_path_bounces) break;
            float eta = sf.entering ? 1.0f / sf.ior : sf.ior;
            float cosi = -v3_dot(r.d, sf.N);
            float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
            v3 d, o;
            if (k < 0.0f) {
                float m = 2.0f * v3_dot(r.d, sf.N);
                d = v3_make(fmaf(-m, sf.N.x, r.d.x), fmaf(-m, sf.N.y, r.d.y), fmaf(-m, sf.N.z, r.d.z));
                o = bias_point(sf.P, sf.N, SHADOW_BIAS);
            } else {
                float m = eta * cosi - sqrtf(k);
                d = v3_make(fmaf(m, sf.N.x, eta * r.d.x), fmaf(m, sf.N.y, eta * r.d.y), fmaf(m, sf.N.z, eta * r.d.z));
                o = bias_point(sf.P, sf.N, -SHADOW_BIAS);
            }
            ray_setup(&r, o, v3_normalize(d));
            continue;
        }
        /* DIFFUSE (and anything else): direct light now, then a cosine-weighted bounce */
        v3 Po = bias_point(sf.P, sf.N, SHADOW_BIAS);
        v3 Ld = direct_light(s, Po, sf.N, sf.albedo, NULL, brute, c, n_shadow);
        L = v3_make(fmaf(thr.x, Ld.x, L.x), fmaf(thr.y, Ld.y, L.y), fmaf(thr.z, Ld.z, L.z));
        if (bounce == g_path_bounces) break;
        float u1 = rng_next(&rng), u2 = rng_next(&rng);
        float rr = sqrtf(u1), phi = 6.28318530717958648f * u2;
        float lx = rr * oracle_sinf(phi + 1.57079632679489662f), ly = rr * oracle_sinf(phi), lz = sqrtf(fmaxf(0.0f, 1.0f - u1));
        /* orthonormal basis around N (Duff et al. 2017, branchless) */
        float sg = copysignf(1.0f, sf.N.z);
        float a = -1.0f / (sg + sf.N.z);
        float b = sf.N.x * sf.N.y * a;
        v3 T = v3_make(1.0f + sg * sf.N.x * sf.N.x * a, sg * b, -sg * sf.N.x);
        v3 B = v3_make(b, sg + sf.N.y * sf.N.y * a, -sf.N.y);
        v3 d = v3_make(fmaf(lz, sf.N.x, fmaf(ly, B.x, lx * T.x)), fmaf(lz, sf.N.y, fmaf(ly, B.y, lx * T.y)),
                       fmaf(lz, sf.N.z, fmaf(ly, B.z, lx * T.z)));
        thr = v3_make(thr.x * sf.albedo.x, thr.y * sf.albedo.y, thr.z * sf.albedo.z);
        ray_setup(&r, Po, v3_normalize(d));
    }
    return L;
}

static uint32_t* g_cost_sp;
static uint32_t *g_cost_pn, *g_cost_pt, *g_cost_sn, *g_cost_st;
int oracle_debug_max_sp(int reset) { int v = t_max_sp; if (reset) t_max_sp = 0; return v; } /* per-pixel fetch counts (analysis of lane utilisation) */
void oracle_set_cost_outputs(uint32_t* pn, uint32_t* pt, uint32_t* sn, uint32_t* st)
{
    g_cost_pn = pn; g_cost_pt = pt; g_cost_sn = sn; g_cost_st = st;
}
void oracle_set_stack_output(uint32_t* max_sp) { g_cost_sp = max_sp; }

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_render(const oracle_scene* s, const float pos[3], const float rot[9], uint32_t mode,
                  const float miss_rgb[3], uint32_t w, uint32_t h,
                  uint32_t y_begin, uint32_t y_end, uint32_t y_step,
                  uint8_t* rgba8, uint32_t* hit_inst, uint32_t* hit_prim, float* hit_t, float* rgb_f32,
                  oracle_stats* stats, int brute_force, int n_threads)
{
    if (!s || w == 0 || h == 0 || y_step == 0) return 1;
    if (y_end > h) y_end = h;
    const v3 o = v3_make(pos[0], pos[1], pos[2]);
    const float width = (float)w, height = (float)h;
    const v3 miss = miss_rgb ? v3_make(miss_rgb[0], miss_rgb[1], miss_rgb[2]) : v3_make(0.0f, 1.0f, 1.0f); /* hlsl:75 */
    uint64_t tot_nodes = 0, tot_tris = 0, tot_shadow = 0, tot_primary = 0, tot_pixels = 0;
    const long n_rows = y_begin < y_end ? (long)((y_end - y_begin + y_step - 1) / y_step) : 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : tot_nodes, tot_tris, tot_shadow, tot_primary, tot_pixels)
    for (long row = 0; row < n_rows; row++) {
        const uint32_t py = y_begin + (uint32_t)row * y_step;
        trav_count c = { 0, 0 };
        uint64_t n_shadow = 0, n_closest = 0;
        for (uint32_t px = 0; px < w; px++) {
            ray r;
            hit_rec hr;
            const size_t pix = (size_t)py * w + px;
            if (mode >= ORACLE_MODE_PATH) {
                /* path tracing: spp jittered samples averaged; hit outputs report sample 0's camera ray */
                v3 acc = v3_make(0.0f, 0.0f, 0.0f);
                hit_rec h0;
                memset(&h0, 0, sizeof(h0));
                for (uint32_t sm = 0; sm < g_path_spp; sm++) {
                    v3 Ls = trace_path(s, rot, o, px, py, width, height, (uint32_t)pix, sm, miss, brute_force, &c, &n_closest, &n_shadow,
                                       sm == 0 ? &h0 : NULL);
                    acc = v3_make(acc.x + Ls.x, acc.y + Ls.y, acc.z + Ls.z);
                }
                const float inv = 1.0f / (float)g_path_spp;
                v3 col = v3_make(acc.x * inv, acc.y * inv, acc.z * inv);
                if (rgba8) {
                    rgba8[4 * pix + 0] = oracle_unorm8(col.x); rgba8[4 * pix + 1] = oracle_unorm8(col.y);
                    rgba8[4 * pix + 2] = oracle_unorm8(col.z); rgba8[4 * pix + 3] = 255;
                }
                if (hit_inst) hit_inst[pix] = h0.hit ? s->tris[h0.tri].inst : ORACLE_MISS;
                if (hit_prim) hit_prim[pix] = h0.hit ? s->tris[h0.tri].prim : ORACLE_MISS;
                if (hit_t) hit_t[pix] = h0.hit ? h0.t : RAY_TMAX;
                if (rgb_f32) { rgb_f32[3 * pix] = col.x; rgb_f32[3 * pix + 1] = col.y; rgb_f32[3 * pix + 2] = col.z; }
                continue;
            }
            n_closest++;
            ray_setup(&r, o, ray_dir(rot, px, py, width, height));
            const trav_count c0 = c;
            if (g_cost_sp) t_max_sp = 0;
            if (brute_force) brute_closest(s, &r, RAY_TMIN, RAY_TMAX, &hr, &c);
            else trace_closest(s, &r, RAY_TMIN, RAY_TMAX, &hr, &c);
            const trav_count c1 = c;
            v3 col = miss;
            uint32_t inst = ORACLE_MISS, prim = ORACLE_MISS;
            if (hr.hit) {
                const oracle_tri* T = &s->tris[hr.tri];
                inst = T->inst; prim = T->prim;
                if (mode >= ORACLE_MODE_LAMBERT) col = shade_lambert(s, &r, &hr, brute_force, &c, &n_shadow);
                else col = shade_debug(mode, inst, prim, hr.t, hr.u, hr.v, r.o, r.d);
            }
            if (g_cost_sp) g_cost_sp[pix] = (uint32_t)t_max_sp;
            if (g_cost_pn) {
                g_cost_pn[pix] = (uint32_t)(c1.nodes - c0.nodes); g_cost_pt[pix] = (uint32_t)(c1.tris - c0.tris);
                g_cost_sn[pix] = (uint32_t)(c.nodes - c1.nodes); g_cost_st[pix] = (uint32_t)(c.tris - c1.tris);
            }
            if (rgba8) {
                rgba8[4 * pix + 0] = oracle_unorm8(col.x);
                rgba8[4 * pix + 1] = oracle_unorm8(col.y);
                rgba8[4 * pix + 2] = oracle_unorm8(col.z);
                rgba8[4 * pix + 3] = 255; /* alpha = 1 (hlsl:55,75,168) */
            }
            if (hit_inst) hit_inst[pix] = inst;
            if (hit_prim) hit_prim[pix] = prim;
            if (hit_t) hit_t[pix] = hr.hit ? hr.t : RAY_TMAX;
            if (rgb_f32) { rgb_f32[3 * pix] = col.x; rgb_f32[3 * pix + 1] = col.y; rgb_f32[3 * pix + 2] = col.z; }
        }
        tot_nodes += c.nodes; tot_tris += c.tris; tot_shadow += n_shadow; tot_primary += n_closest; tot_pixels += w;
    }
    if (stats) {
        stats->rays_primary = tot_primary; stats->rays_shadow = tot_shadow;
        stats->nodes_visited = tot_nodes; stats->tris_tested = tot_tris;
        stats->pixels = tot_pixels;
    }
    return 0;
}
