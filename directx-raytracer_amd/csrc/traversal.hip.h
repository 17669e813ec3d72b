// Device-side traversal shared by the render kernel (render_kernels.hip) and the path-tracing pipeline (path_kernels.hip):
// vector helpers, the ray, Moeller-Trumbore, the per-lane LDS stack, the quantised wide-node step and the wave-scheduled
// closest-hit / any-hit traversals.  Everything is internal to the including translation unit (anonymous namespace).
//
// Arithmetic contract: identical, operation for operation, to oracle/crt_oracle.c (compiled with
// -ffp-contract=off; fused multiply-adds only where fmaf()/fma() is written; correctly rounded / and sqrt).
#pragma once

#include "render_kernels.h"
#include "bvh_pack.h"

#include <hip/hip_runtime.h>

#include <climits>

namespace crt {
namespace {


constexpr float kTMin = 0.001f;   // hlsl:51
constexpr float kTMax = 10000.0f; // hlsl:52
constexpr float kDirEps = 1e-20f;
constexpr float kCullPad = 1.00000381469726562f; // 1 + 2^-18, see oracle trace_closest
constexpr float kShadowBias = 1e-3f;
constexpr float kFourPi = 12.566370614359172f;
constexpr int kBlock = 256;
constexpr uint32_t kBoostAfter = 300; // traversal-loop iterations after which a wavefront raises its issue priority
#ifndef NODE_STEPS
#define NODE_STEPS 2
#endif
#ifndef CRT_PROF
#define CRT_PROF 0
#endif
// diagnostics that change what a frame does or costs (per-workgroup timeline stamps, dropping the most expensive
// packets): only in the diagnostic builds of tools/diag_build.sh / tools/prof_build.sh, never in the product
#ifndef CRT_DIAG
#define CRT_DIAG CRT_PROF
#endif
// Register budget: the primary/shadow-ray variant is asked for 7 wavefronts per SIMD (<= 72 VGPRs; nothing spilled in the
// headline variant since the pixel position is recomputed after the traversal, render_kernels.hip).  With the 64-byte quantised nodes a node in flight is 16 registers instead of 28, and with the LDS
// stack at 16 entries (4 KB per wavefront) the CU holds those 28 wavefronts: 0.295 ms against 0.306 at 6 per SIMD;
// 8 per SIMD (64 VGPRs) spills inside the loops (0.37).  The path-tracing variant keeps the compiler's choice.
#ifndef CRT_WAVES_PER_EU
#define CRT_WAVES_PER_EU 7
#endif
#define CRT_OCCUPANCY_ATTR __attribute__((amdgpu_waves_per_eu(CRT_WAVES_PER_EU, 8)))
// scalar-cache fetches of records a whole wavefront shares (see loadNodeUniform): in the descent from the root, in any
// node step whose lanes agree, and in leaves
#ifndef UNIFORM_DESCENT
#define UNIFORM_DESCENT 1
#endif
#ifndef UNIFORM_STEP
#define UNIFORM_STEP 1
#endif
#ifndef UNIFORM_LEAF
#define UNIFORM_LEAF 1
#endif
#ifndef LEAF_PAIRS
#define LEAF_PAIRS 1
#endif
#ifndef EARLY_FETCH
#define EARLY_FETCH 1
#endif
// the next node's record is requested before the current step's pushes (nodeStepClosestAt)
// (EARLY_FETCH above); one copy of the traversal loops per direction octant for packets whose rays agree on it
#ifndef OCTANT_SPECIALISE
#define OCTANT_SPECIALISE 1
#endif
constexpr uint32_t kGroupMax = 16; // grid padding unit: tiles per XCD group never exceed this

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{ x, y, z }; }
__device__ __forceinline__ F3 sub3(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b)
{
    return f3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ F3 normalize3(F3 a)
{
    const float inv = 1.0f / sqrtf(dot3(a, a));
    return f3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ float frac1(float x) { return x - floorf(x); }
__device__ __forceinline__ float saturate1(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + t * (b - a); }

// sin() with the operation sequence of oracle_sinf: Cody-Waite reduction by 2*pi in double, odd Taylor
// polynomial to r^23 (Horner, fma), one rounding to float.  fp64 runs at full rate on CDNA4.
__device__ __forceinline__ float sinContract(float x)
{
    const double xd = static_cast<double>(x);
    const double k = rint(xd * 0x1.45f306dc9c883p-3);
    double r = fma(-k, 0x1.921fb54442d18p+2, xd);
    r = fma(-k, 0x1.1a62633145c07p-52, r);
    const double r2 = r * r;
    double p = -0x1.761b41316381ap-75;
    p = fma(p, r2, 0x1.71b8ef6dcf572p-66);
    p = fma(p, r2, -0x1.2f49b46814157p-57);
    p = fma(p, r2, 0x1.952c77030ad4ap-49);
    p = fma(p, r2, -0x1.ae7f3e733b81fp-41);
    p = fma(p, r2, 0x1.6124613a86d09p-33);
    p = fma(p, r2, -0x1.ae64567f544e4p-26);
    p = fma(p, r2, 0x1.71de3a556c734p-19);
    p = fma(p, r2, -0x1.a01a01a01a01ap-13);
    p = fma(p, r2, 0x1.1111111111111p-7);
    p = fma(p, r2, -0x1.5555555555555p-3);
    p = p * r2;
    return static_cast<float>(fma(p, r, r));
}
__device__ __forceinline__ float hashSin(float x, float k) { return frac1(sinContract(x) * k); }

__device__ __forceinline__ uint32_t unorm8(float c) { return static_cast<uint32_t>(saturate1(c) * 255.0f + 0.5f); }

struct Ray {
    F3 o, d;
    F3 idir, noid;
};

__device__ __forceinline__ float safeRcp(float d)
{
    const float ds = (fabsf(d) < kDirEps) ? copysignf(kDirEps, d) : d;
    return 1.0f / ds;
}

__device__ __forceinline__ Ray makeRay(F3 o, F3 d)
{
    Ray r;
    r.o = o;
    r.d = d;
    r.idir = f3(safeRcp(d.x), safeRcp(d.y), safeRcp(d.z));
    r.noid = f3(-(o.x * r.idir.x), -(o.y * r.idir.y), -(o.z * r.idir.z));
    return r;
}

// Moeller-Trumbore, two sided; u = weight of v1, v = weight of v2.  NaN/inf from det == 0 fail the compares.
__device__ __forceinline__ bool triTest(const Ray& r, const float4 a, const float4 b, const float4 c, float tmin,
                                        float& t, float& u, float& v)
{
    const F3 e1 = f3(b.x, b.y, b.z), e2 = f3(c.x, c.y, c.z);
    const F3 p = cross3(r.d, e2);
    const float det = dot3(e1, p);
    const float inv = 1.0f / det;
    const F3 s = sub3(r.o, f3(a.x, a.y, a.z));
    u = dot3(s, p) * inv;
    const F3 q = cross3(s, e1);
    v = dot3(r.d, q) * inv;
    t = dot3(e2, q) * inv;
    return (u >= 0.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > tmin);
}

// Per-lane traversal stack.  The first `cap` entries live in LDS (entry e of lane l at dword e*64+l: conflict free); cap
// is chosen so that the 28 wavefronts per CU the register budget allows fit its 160 KB (16 entries = 4 KB per wavefront).  No ray of the test scenes ever holds more than 15
// entries while the trees are 24..26 deep, but the builder allows depth 32, so deeper entries spill to a per-lane slice
// of a global arena that is never touched otherwise: any tree stays correct with the small LDS footprint.
struct Stack {
    int* lds;    // s_stack + lane
    int* spill;  // arena slice of this lane: kStackEntries - cap entries are ever needed, kStackEntries reserved
    int cap;     // wave-uniform
    int sp;
#if CRT_PROF // diagnostic build (tools/prof_build.sh): where a wavefront's cycles go, never compiled into the product
    unsigned long long tNode = 0, tLeaf = 0;
    uint32_t itNode = 0, itLeaf = 0, lanesNode = 0, lanesLeaf = 0;
    // divergent (per-lane fetched) steps: how many, lanes in them, runs of consecutive lanes on the same record, distinct records
    uint32_t dvN = 0, dvNLanes = 0, dvNRuns = 0, dvNDistinct = 0, dvL = 0, dvLLanes = 0, dvLRuns = 0, dvLDistinct = 0, unN = 0, unL = 0;
    __device__ __forceinline__ void divStats(int cur, uint32_t& steps, uint32_t& lanes, uint32_t& runs, uint32_t& distinct)
    {
        const unsigned long long act = __ballot(true);
        const int prev = __shfl_up(cur, 1, 64);
        const uint32_t lane = threadIdx.x & 63u;
        const bool prevActive = lane > 0 && ((act >> (lane - 1)) & 1ull);
        runs += __popcll(__ballot(!prevActive || prev != cur));
        steps++;
        lanes += __popcll(act);
        unsigned long long rest = act;
        while (rest) {
            const int first = __ffsll(static_cast<long long>(rest)) - 1;
            const int v = __shfl(cur, first, 64);
            rest &= ~__ballot(cur == v);
            distinct++;
        }
    }
#endif
    __device__ __forceinline__ void push(int v)
    {
        if (sp < cap) lds[sp * 64] = v;
        else spill[sp - cap] = v;
        sp++;
    }
    __device__ __forceinline__ int pop()
    {
        sp--;
        return sp < cap ? lds[sp * 64] : spill[sp - cap];
    }
};

struct Hit {
    float t, u, v;
    uint32_t tri; // triangle record: leaf-order index (legacy layout) or granule address in the packed buffer
    uint32_t gid;
};

__device__ __forceinline__ float ubyteToFloat(uint32_t w, int k) { return static_cast<float>((w >> (8 * k)) & 0xFFu); } // v_cvt_f32_ubyteK

typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v* ConstQuadPtr; // constant address space + wave-uniform address = scalar loads
__device__ __forceinline__ float4 quadOf(const f4v a) { return make_float4(a.x, a.y, a.z, a.w); }

__device__ __forceinline__ int pick4(const int4& v, uint32_t i) // v[i], i in 0..3, without dynamic register indexing
{
    const int lo = (i & 1u) ? v.y : v.x, hi = (i & 1u) ? v.w : v.z;
    return (i & 2u) ? hi : lo;
}

// =====================================================================================================================
// Tree layouts.  A layout says what a reference is, how a node record is fetched and how one node step turns it into the
// next current reference plus pushes; everything else (wave-level scheduling, leaves, the octant dispatch) is shared.
//
// ---- LayLegacy: 4-wide, node = crt_bvh_node4q, 64 bytes = four dwordx4 loads:
//   {lo.x lo.y lo.z s.x} {s.y s.z qlo_x qhi_x} {qlo_y qhi_y qlo_z qhi_z} {ref[4]}
// separate triangle array; references are signed (>= 0 node index, < 0 leaf ~((first << 3) | count)).
// ---- LayPacked<W>: the packed wide tree of bvh_pack.h, W = 4 (48-byte nodes, three loads) or 8 (80 bytes, five): nodes
// and triangles in ONE buffer, a node's children back to back, references (address << 3) | kind.
//
// In both the child planes are 8-bit offsets from the node's own minimum corner: plane = fma(q, s, lo).  A per-lane fetch
// request costs this kernel far more than vector arithmetic does (24 / 48 extra dependent VALU per step measured +9 % /
// +21 %, one extra 4-byte touch per pushed child +29 %), so records are kept to as few requests per lane as possible and
// decoded in registers.  The decode is folded into the slab test: t(q) = fma(q, s * idir, fma(lo, idir, -o * idir)),
// monotonic in q with the sign of idir, so for a known direction octant (OCT < 8) the near plane of each axis is a fixed
// member of the (qlo, qhi) pair and the min/max pairs of the generic form (OCT = 8) disappear -- bit for bit the same values.
// An unused child slot is the leaf of no triangles with a point box (qlo = qhi = 0): the slab test rejects it, there is no
// test of the reference.  (Not an inverted box: the min/max form of the mixed-octant path would turn that into the whole
// node and visit the empty leaf every time.)
// =====================================================================================================================

// the slab tests of children 2j and 2j + 1 of a node whose planes sit in bytes 2j, 2j + 1 of the six given words
// (Round 3, measured and not kept: one v_perm_b32 per plane PAIR building two halfs 1024 + q (0x6400 | q) that v_fma_mix_f32
// multiplies directly, instead of one v_cvt_f32_ubyte per plane: 12 instructions fewer per node step and 0.311 vs 0.284 ms.
// tools/micro/valu_cost.hip has the reason: v_fma_mix_f32 and v_perm_b32 issue at half the rate of v_fma_f32 -- as do
// v_cvt_f32_ubyte, float and integer min / max, compares, selects and everything VOP3-only or SDWA: 4.2 against 2.5 cycles per
// wavefront instruction -- so a conversion folded into a mixed-precision fma costs what the pair it replaces cost.)
template <int OCT>
__device__ __forceinline__ void slabPair(uint32_t lx, uint32_t hx, uint32_t ly, uint32_t hy, uint32_t lz, uint32_t hz, int pair, const Ray& r,
                                         float ax, float ay, float az, float bx, float by, float bz, float tmin, float tcull, float tn[2], bool hit[2])
{
    // near / far member of each (qlo, qhi) pair: fixed by the template octant, or picked per lane from the direction's sign
    // bits (mixed-octant wavefronts: bounce rays) -- six selects instead of the twelve min/max of the textbook form, and
    // the same values: t is monotonic in q with the sign of idir
    const bool sx = (OCT < 8) ? (OCT & 1) != 0 : (__float_as_uint(r.d.x) >> 31) != 0u;
    const bool sy = (OCT < 8) ? (OCT & 2) != 0 : (__float_as_uint(r.d.y) >> 31) != 0u;
    const bool sz = (OCT < 8) ? (OCT & 4) != 0 : (__float_as_uint(r.d.z) >> 31) != 0u;
    const uint32_t nxw = sx ? hx : lx, fxw = sx ? lx : hx, nyw = sy ? hy : ly, fyw = sy ? ly : hy, nzw = sz ? hz : lz, fzw = sz ? lz : hz;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int byte = 2 * pair + i;
        const float t_n = fmaxf(fmaxf(fmaf(ubyteToFloat(nxw, byte), ax, bx), fmaf(ubyteToFloat(nyw, byte), ay, by)), fmaxf(fmaf(ubyteToFloat(nzw, byte), az, bz), tmin));
        const float t_f = fminf(fminf(fmaf(ubyteToFloat(fxw, byte), ax, bx), fmaf(ubyteToFloat(fyw, byte), ay, by)), fminf(fmaf(ubyteToFloat(fzw, byte), az, bz), tcull));
        tn[i] = t_n;
        hit[i] = t_n <= t_f;
    }
}

struct LayLegacy {
    static constexpr int kWidth = 4;
    static constexpr int kDone = INT_MIN; // traversal finished (not a valid leaf reference)
    static constexpr int kRoot = 0;
    static constexpr int kStackPerLevel = 3;
    static constexpr int kWavesPerEu = CRT_WAVES_PER_EU;
    struct Node {
        float4 q0, q1, q2;
        int4 refs;
    };
    static __device__ __forceinline__ bool inner(int c) { return c >= 0; }
    static __device__ __forceinline__ bool leaf(int c) { return (c < 0) & (c != kDone); }
    static __device__ __forceinline__ Node load(const float4* __restrict__ nodes, int ref)
    {
        const float4* N = nodes + 4 * static_cast<size_t>(ref);
        Node nd;
        nd.q0 = N[0]; nd.q1 = N[1]; nd.q2 = N[2];
        nd.refs = *reinterpret_cast<const int4*>(N + 3);
        return nd;
    }
    static __device__ __forceinline__ Node loadUniform(const float4* nodes, int ref)
    {
        ConstQuadPtr C = (ConstQuadPtr)(reinterpret_cast<uintptr_t>(nodes + 4 * static_cast<size_t>(ref)));
        const f4v a = C[0], b = C[1], c = C[2], g = C[3];
        Node nd;
        nd.q0 = quadOf(a); nd.q1 = quadOf(b); nd.q2 = quadOf(c);
        nd.refs = make_int4(__float_as_int(g.x), __float_as_int(g.y), __float_as_int(g.z), __float_as_int(g.w));
        return nd;
    }
    // leaf reference -> first triangle and count; triangle i of the leaf: its record and the id kept in Hit::tri
    static __device__ __forceinline__ void leafRange(int c, uint32_t& first, uint32_t& cnt)
    {
        const uint32_t code = static_cast<uint32_t>(~c);
        first = code >> 3;
        cnt = code & 7u;
    }
    static __device__ __forceinline__ uint32_t triId(uint32_t first, uint32_t i) { return first + i; }
    static __device__ __forceinline__ const float4* triPtr(const float4* __restrict__ tris, uint32_t id) { return tris + 3 * static_cast<size_t>(id); }
    // shading / uv record of a hit: leaf order
    static __device__ __forceinline__ uint32_t shadeIndex(uint32_t id, uint32_t) { return id; }

    template <int OCT>
    static __device__ __forceinline__ void slab(const Node& nd, const Ray& r, float tmin, float tcull, float tn[4], bool hit[4])
    {
        const float ax = nd.q0.w * r.idir.x, ay = nd.q1.x * r.idir.y, az = nd.q1.y * r.idir.z;
        const float bx = fmaf(nd.q0.x, r.idir.x, r.noid.x), by = fmaf(nd.q0.y, r.idir.y, r.noid.y), bz = fmaf(nd.q0.z, r.idir.z, r.noid.z);
        const uint32_t lx = __float_as_uint(nd.q1.z), hx = __float_as_uint(nd.q1.w), ly = __float_as_uint(nd.q2.x), hy = __float_as_uint(nd.q2.y),
                       lz = __float_as_uint(nd.q2.z), hz = __float_as_uint(nd.q2.w);
#pragma unroll
        for (int j = 0; j < 2; j++) slabPair<OCT>(lx, hx, ly, hy, lz, hz, j, r, ax, ay, az, bx, by, bz, tmin, tcull, tn + 2 * j, hit + 2 * j);
    }

    // closest hit: visit the hit children nearest first.  Order key = (bits(t_near) & 0x7FFFFFFC) | slot: t_near >= 0 so its
    // bit pattern orders like the float, the two low bits hold the slot (keys are unique, order is total and identical in
    // the oracle); misses get 0xFFFFFFFF.  Five min/max pairs sort the four keys.
    // EARLY (the step is followed by another node step of the same scheduling decision): the nearest hit child -- or the
    // popped entry of a lane that hit nothing -- is known after two of the network's min levels, so its record is requested
    // right there into ndNext, and the rest of the ordering and the pushes of the other hit children run under that fetch
    // instead of in front of it.  Same stack operations per lane in the same order, so results and counters are unchanged.
    // (Round 3, measured and not kept: ordering only the NEAREST child and pushing the others in slot order -- no network, no
    // select chain per pushed child; 0.2 % more node visits on the 1M-triangle frame by the oracle's count.  Shaded frame
    // 0.2803 vs 0.2854 ms, icosphere soup 0.246 vs 0.255, but primary rays only 0.201 vs 0.192 and the 5M-triangle frame
    // 0.368 vs 0.361: the network runs under the early fetch, off the step's dependent chain, so removing it frees issue
    // slots nobody was waiting for.)
    template <bool COUNT, int OCT, bool EARLY>
    static __device__ __forceinline__ void closestStep(const Node& nd, const Ray& r, float tmin, float tcull, Stack& stack,
                                                       int& cur, uint32_t& cntNodes, const float4* __restrict__ nodes, Node& ndNext)
    {
        const int4 refs = nd.refs;
        if (COUNT) cntNodes++;
        float tn[4];
        bool hit[4];
        slab<OCT>(nd, r, tmin, tcull, tn, hit);
        uint32_t key[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            key[k] = hit[k] ? ((__float_as_uint(tn[k]) & 0x7FFFFFFCu) | static_cast<uint32_t>(k)) : 0xFFFFFFFFu;
        const uint32_t nearest = min(min(key[0], key[1]), min(key[2], key[3])); // = key[0] after the network
        const bool any = nearest != 0xFFFFFFFFu;
        cur = any ? pick4(refs, nearest & 3u) : (stack.sp == 0 ? kDone : stack.pop()); // a lane pops or pushes, never both
        if (EARLY) {
            if (cur >= 0) ndNext = load(nodes, cur);
        }
#define CRT_CSWAP(a, b) { const uint32_t lo = min(key[a], key[b]), hi = max(key[a], key[b]); key[a] = lo; key[b] = hi; }
        CRT_CSWAP(0, 1) CRT_CSWAP(2, 3) CRT_CSWAP(0, 2) CRT_CSWAP(1, 3) CRT_CSWAP(1, 2)
#undef CRT_CSWAP
        if (key[1] != 0xFFFFFFFFu) {
            if (key[3] != 0xFFFFFFFFu) stack.push(pick4(refs, key[3] & 3u)); // farthest first: the nearest pending child pops first
            if (key[2] != 0xFFFFFFFFu) stack.push(pick4(refs, key[2] & 3u));
            stack.push(pick4(refs, key[1] & 3u));
        }
    }

    // any hit: order independent, children taken in slot order
    template <bool COUNT, int OCT, bool EARLY>
    static __device__ __forceinline__ void anyStep(const Node& nd, const Ray& r, float tmin, float tcull, Stack& stack,
                                                   int& cur, uint32_t& cntNodes, const float4* __restrict__ nodes, Node& ndNext)
    {
        const int4 refs = nd.refs;
        if (COUNT) cntNodes++;
        float tn[4];
        bool hit[4];
        slab<OCT>(nd, r, tmin, tcull, tn, hit);
        const bool h0 = hit[0], h1 = hit[1], h2 = hit[2], h3 = hit[3];
        cur = h0 ? refs.x : (h1 ? refs.y : (h2 ? refs.z : (h3 ? refs.w : (stack.sp == 0 ? kDone : stack.pop()))));
        if (EARLY) {
            if (cur >= 0) ndNext = load(nodes, cur);
        }
        // first hit slot became current; later hit slots are pushed, last slot first
        if (h3 & (h0 | h1 | h2)) stack.push(refs.w);
        if (h2 & (h0 | h1)) stack.push(refs.z);
        if (h1 & h0) stack.push(refs.y);
    }
};

// The packed layouts are an experiment of round 3 that lost (48-byte 4-wide nodes +6 %, 80-byte 8-wide nodes
// +34 % on the 1M-triangle frame): compiled only with -DCRT_PACKED_LAYOUTS=1 (tools/variant_build.sh packed ...), host build only.
#ifndef CRT_PACKED_LAYOUTS
#define CRT_PACKED_LAYOUTS 0
#endif
#if CRT_PACKED_LAYOUTS
template <int W> struct PackedNode;
template <> struct PackedNode<4> { float4 q0, q1, q2; };
template <> struct PackedNode<8> { float4 q0, q1, q2, q3, q4; };

template <int W>
struct LayPacked {
    typedef PackFmt<W> F;
    static constexpr int kWidth = W;
    static constexpr int kDone = static_cast<int>(kRefDone);
    static constexpr int kRoot = static_cast<int>(kRefInner); // the root's record sits at address 0
    static constexpr int kStackPerLevel = W - 1;
    // register budget of the primary / shadow-ray kernel: an 8-wide node in flight is 20 registers, and the early fetch holds two
#ifndef CRT_WAVES_PER_EU_W8
#define CRT_WAVES_PER_EU_W8 5
#endif
    static constexpr int kWavesPerEu = W == 8 ? CRT_WAVES_PER_EU_W8 : CRT_WAVES_PER_EU;
    static constexpr uint32_t kQuads = F::kGranuleBytes / 16u;             // float4 per granule
    static constexpr uint32_t kPayload = (1u << F::kPayloadBits) - 1u;     // low key bits that carry a child designator
    typedef PackedNode<W> Node;
    static __device__ __forceinline__ bool inner(int c) { return (static_cast<uint32_t>(c) & 7u) == kRefInner; }
    static __device__ __forceinline__ bool leaf(int c) { return ((static_cast<uint32_t>(c) & 7u) != kRefInner) & (c != kDone); }
    static __device__ __forceinline__ const float4* at(const float4* buf, uint32_t address) { return buf + static_cast<size_t>(address) * kQuads; }
    static __device__ __forceinline__ Node load(const float4* __restrict__ nodes, int ref)
    {
        const float4* N = at(nodes, static_cast<uint32_t>(ref) >> 3);
        Node nd;
        nd.q0 = N[0]; nd.q1 = N[1]; nd.q2 = N[2];
        if constexpr (W == 8) { nd.q3 = N[3]; nd.q4 = N[4]; }
        return nd;
    }
    static __device__ __forceinline__ Node loadUniform(const float4* nodes, int ref)
    {
        ConstQuadPtr C = (ConstQuadPtr)(reinterpret_cast<uintptr_t>(at(nodes, static_cast<uint32_t>(ref) >> 3)));
        Node nd;
        nd.q0 = quadOf(C[0]); nd.q1 = quadOf(C[1]); nd.q2 = quadOf(C[2]);
        if constexpr (W == 8) { nd.q3 = quadOf(C[3]); nd.q4 = quadOf(C[4]); }
        return nd;
    }
    static __device__ __forceinline__ void leafRange(int c, uint32_t& first, uint32_t& cnt)
    {
        first = static_cast<uint32_t>(c) >> 3;
        cnt = static_cast<uint32_t>(c) & 7u;
    }
    static __device__ __forceinline__ uint32_t triId(uint32_t first, uint32_t i) { return first + i * F::kTriGranules; }
    static __device__ __forceinline__ const float4* triPtr(const float4* __restrict__ tris, uint32_t id) { return at(tris, id); }
    // shading / uv record of a hit: input order, by the gid the triangle record carries
    static __device__ __forceinline__ uint32_t shadeIndex(uint32_t, uint32_t gid) { return gid; }

    template <int OCT>
    static __device__ __forceinline__ void slab(const Node& nd, const Ray& r, float tmin, float tcull, float tn[W], bool hit[W])
    {
        const uint32_t sc = __float_as_uint(nd.q1.x);
        const float sx = __uint_as_float((sc & 0x3FFu) << 21), sy = __uint_as_float(((sc >> 10) & 0x3FFu) << 21), sz = __uint_as_float(sc & 0x7FE00000u);
        const float ax = sx * r.idir.x, ay = sy * r.idir.y, az = sz * r.idir.z;
        const float bx = fmaf(nd.q0.x, r.idir.x, r.noid.x), by = fmaf(nd.q0.y, r.idir.y, r.noid.y), bz = fmaf(nd.q0.z, r.idir.z, r.noid.z);
        if constexpr (W == 4) {
            const uint32_t lx = __float_as_uint(nd.q1.z), hx = __float_as_uint(nd.q1.w), ly = __float_as_uint(nd.q2.x), hy = __float_as_uint(nd.q2.y),
                           lz = __float_as_uint(nd.q2.z), hz = __float_as_uint(nd.q2.w);
#pragma unroll
            for (int j = 0; j < 2; j++) slabPair<OCT>(lx, hx, ly, hy, lz, hz, j, r, ax, ay, az, bx, by, bz, tmin, tcull, tn + 2 * j, hit + 2 * j);
        } else {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const uint32_t lx = __float_as_uint(half ? nd.q2.y : nd.q2.x), hx = __float_as_uint(half ? nd.q2.w : nd.q2.z);
                const uint32_t ly = __float_as_uint(half ? nd.q3.y : nd.q3.x), hy = __float_as_uint(half ? nd.q3.w : nd.q3.z);
                const uint32_t lz = __float_as_uint(half ? nd.q4.y : nd.q4.x), hz = __float_as_uint(half ? nd.q4.w : nd.q4.z);
#pragma unroll
                for (int j = 0; j < 2; j++)
                    slabPair<OCT>(lx, hx, ly, hy, lz, hz, j, r, ax, ay, az, bx, by, bz, tmin, tcull, tn + 4 * half + 2 * j, hit + 4 * half + 2 * j);
            }
        }
    }
    // the child designators m_k = (offset << 3) | kind, each in the low bits of a register (upper bits: whatever follows)
    static __device__ __forceinline__ void designators(const Node& nd, uint32_t m[W])
    {
        if constexpr (W == 4) {
            const uint32_t w = __float_as_uint(nd.q1.y);
            m[0] = w; m[1] = w >> 8; m[2] = w >> 16; m[3] = w >> 24;
        } else {
            const uint32_t a = __float_as_uint(nd.q1.y), b = __float_as_uint(nd.q1.z), c = __float_as_uint(nd.q1.w);
            m[0] = a; m[1] = a >> 10; m[2] = a >> 20; m[3] = b; m[4] = b >> 10; m[5] = b >> 20; m[6] = c; m[7] = c >> 10;
        }
    }

    // closest hit: visit the hit children nearest first.  Order key = (bits(t_near) & ~payload) | m_k: t_near >= 0 so its bit
    // pattern orders like the float; the low bits hold the child's designator, which makes keys unique AND is all a push
    // needs: reference = base8 + (key & payload) -- the ordering network carries the children, nothing is looked up by slot
    // afterwards.  Misses get 0xFFFFFFFF.  EARLY: see LayLegacy::closestStep.
    template <bool COUNT, int OCT, bool EARLY>
    static __device__ __forceinline__ void closestStep(const Node& nd, const Ray& r, float tmin, float tcull, Stack& stack,
                                                       int& cur, uint32_t& cntNodes, const float4* __restrict__ nodes, Node& ndNext)
    {
        if (COUNT) cntNodes++;
        float tn[W];
        bool hit[W];
        slab<OCT>(nd, r, tmin, tcull, tn, hit);
        uint32_t m[W], key[W];
        designators(nd, m);
#pragma unroll
        for (int k = 0; k < W; k++)
            key[k] = hit[k] ? ((__float_as_uint(tn[k]) & ~kPayload) | (m[k] & kPayload)) : 0xFFFFFFFFu;
        uint32_t nearest = min(min(key[0], key[1]), min(key[2], key[3]));
        if constexpr (W == 8) nearest = min(nearest, min(min(key[4], key[5]), min(key[6], key[7])));
        const bool any = nearest != 0xFFFFFFFFu;
        const uint32_t base8 = __float_as_uint(nd.q0.w);
        cur = any ? static_cast<int>(base8 + (nearest & kPayload)) : (stack.sp == 0 ? kDone : stack.pop()); // a lane pops or pushes, never both
        if (EARLY) {
            if (inner(cur)) ndNext = load(nodes, cur);
        }
#define CRT_CSWAP(a, b) { const uint32_t lo = min(key[a], key[b]), hi = max(key[a], key[b]); key[a] = lo; key[b] = hi; }
        if constexpr (W == 4) {
            CRT_CSWAP(0, 1) CRT_CSWAP(2, 3) CRT_CSWAP(0, 2) CRT_CSWAP(1, 3) CRT_CSWAP(1, 2)
            if (key[1] != 0xFFFFFFFFu) {
                if (key[3] != 0xFFFFFFFFu) stack.push(static_cast<int>(base8 + (key[3] & kPayload))); // farthest first: the nearest pending child pops first
                if (key[2] != 0xFFFFFFFFu) stack.push(static_cast<int>(base8 + (key[2] & kPayload)));
                stack.push(static_cast<int>(base8 + (key[1] & kPayload)));
            }
        } else {
            // Batcher's odd-even merge sort for eight keys: 19 compare-exchanges
            CRT_CSWAP(0, 1) CRT_CSWAP(2, 3) CRT_CSWAP(4, 5) CRT_CSWAP(6, 7)
            CRT_CSWAP(0, 2) CRT_CSWAP(1, 3) CRT_CSWAP(4, 6) CRT_CSWAP(5, 7)
            CRT_CSWAP(1, 2) CRT_CSWAP(5, 6)
            CRT_CSWAP(0, 4) CRT_CSWAP(1, 5) CRT_CSWAP(2, 6) CRT_CSWAP(3, 7)
            CRT_CSWAP(2, 4) CRT_CSWAP(3, 5)
            CRT_CSWAP(1, 2) CRT_CSWAP(3, 4) CRT_CSWAP(5, 6)
            if (key[1] != 0xFFFFFFFFu) {
                if (key[2] != 0xFFFFFFFFu) {
                    if (key[3] != 0xFFFFFFFFu) {
                        if (key[4] != 0xFFFFFFFFu) {
                            if (key[7] != 0xFFFFFFFFu) stack.push(static_cast<int>(base8 + (key[7] & kPayload)));
                            if (key[6] != 0xFFFFFFFFu) stack.push(static_cast<int>(base8 + (key[6] & kPayload)));
                            if (key[5] != 0xFFFFFFFFu) stack.push(static_cast<int>(base8 + (key[5] & kPayload)));
                            stack.push(static_cast<int>(base8 + (key[4] & kPayload)));
                        }
                        stack.push(static_cast<int>(base8 + (key[3] & kPayload)));
                    }
                    stack.push(static_cast<int>(base8 + (key[2] & kPayload)));
                }
                stack.push(static_cast<int>(base8 + (key[1] & kPayload)));
            }
        }
#undef CRT_CSWAP
    }

    // any hit: order independent, children taken in slot order: the first hit slot becomes current, later hit slots are
    // pushed, last slot first
    template <bool COUNT, int OCT, bool EARLY>
    static __device__ __forceinline__ void anyStep(const Node& nd, const Ray& r, float tmin, float tcull, Stack& stack,
                                                   int& cur, uint32_t& cntNodes, const float4* __restrict__ nodes, Node& ndNext)
    {
        if (COUNT) cntNodes++;
        float tn[W];
        bool hit[W];
        slab<OCT>(nd, r, tmin, tcull, tn, hit);
        uint32_t m[W];
        designators(nd, m);
        const uint32_t base8 = __float_as_uint(nd.q0.w);
        uint32_t first = 0;
        bool any = false;
#pragma unroll
        for (int k = W - 1; k >= 0; k--) {
            first = hit[k] ? m[k] : first;
            any |= hit[k];
        }
        cur = any ? static_cast<int>(base8 + (first & kPayload)) : (stack.sp == 0 ? kDone : stack.pop());
        if (EARLY) {
            if (inner(cur)) ndNext = load(nodes, cur);
        }
        bool before[W]; // some slot below k was hit
        before[0] = false;
#pragma unroll
        for (int k = 1; k < W; k++) before[k] = before[k - 1] | hit[k - 1];
#pragma unroll
        for (int k = W - 1; k >= 1; k--)
            if (hit[k] & before[k]) stack.push(static_cast<int>(base8 + (m[k] & kPayload)));
    }
};

#endif // CRT_PACKED_LAYOUTS

// Uniform descent: the rays of an 8x8 packet start at the root and usually agree on the first few nodes.  While every
// active lane stands on the SAME inner node its record is fetched once through the scalar cache (the node address is
// wave-uniform, so the loads become s_load) instead of 64 identical per-lane vector fetches; each lane still runs its own
// slab tests, ordering and pushes, so results and counters are exactly those of the per-lane loop that follows.
#if UNIFORM_DESCENT
#define CRT_UNIFORM_DESCENT(STEP)                                                                                              \
    for (;;) {                                                                                                                 \
        const int c0 = __builtin_amdgcn_readfirstlane(cur);                                                                    \
        if (!L::inner(c0) || __ballot(cur != c0) != 0ull) break;                                                               \
        typename L::Node ndUnused;                                                                                             \
        L::template STEP<COUNT, OCT, false>(L::loadUniform(nodes, c0), r, tmin, tcull, stack, cur, cntNodes, nodes, ndUnused);  \
    }
#else
#define CRT_UNIFORM_DESCENT(STEP)
#endif
#if CRT_PROF
#define CRT_DIV_STATS_NODE stack.divStats(cur, stack.dvN, stack.dvNLanes, stack.dvNRuns, stack.dvNDistinct);
#define CRT_DIV_STATS_LEAF stack.divStats(cur, stack.dvL, stack.dvLLanes, stack.dvLRuns, stack.dvLDistinct);
#else
#define CRT_DIV_STATS_NODE
#define CRT_DIV_STATS_LEAF
#endif
// One node step of the lanes standing on inner nodes (called with exactly those lanes active): through the scalar cache
// when they all stand on the same node, per lane otherwise.
// (Measured and rejected: fetching the DISTINCT nodes of a divergent step once each -- 6.6 distinct nodes among 50 wanting
// lanes on the 1M-triangle frame -- by the first lanes of the wavefront and handing them out through LDS: a scalar loop
// peels the distinct values, fetchers load and ds_write, every lane ds_reads its slot.  Bit-exact, a seventh of the
// per-lane requests, and 0.49 ms instead of 0.31: the peeling loop and two LDS round trips per step cost far more than
// the requests they save.)
// CRT_NODE_STEPS: the NODE_STEPS node steps of one scheduling decision.  The first fetches its record here; with
// EARLY_FETCH every step but the last requests the next record itself (see closestStep), per lane.
#if UNIFORM_STEP
#define CRT_FIRST_NODE_STEP(STEP, EARLY)                                                                                       \
    {                                                                                                                          \
        const int c0 = __builtin_amdgcn_readfirstlane(cur);                                                                    \
        if (__ballot(cur != c0) == 0ull) {                                                                                     \
            L::template STEP<COUNT, OCT, EARLY>(L::loadUniform(nodes, c0), r, tmin, tcull, stack, cur, cntNodes, nodes, ndNext); \
        } else {                                                                                                               \
            CRT_DIV_STATS_NODE                                                                                                 \
            L::template STEP<COUNT, OCT, EARLY>(L::load(nodes, cur), r, tmin, tcull, stack, cur, cntNodes, nodes, ndNext);      \
        }                                                                                                                      \
    }
#else
#define CRT_FIRST_NODE_STEP(STEP, EARLY) L::template STEP<COUNT, OCT, EARLY>(L::load(nodes, cur), r, tmin, tcull, stack, cur, cntNodes, nodes, ndNext);
#endif
#if EARLY_FETCH
#define CRT_NODE_STEPS(STEP)                                                                                                   \
    if (L::inner(cur)) {                                                                                                       \
        typename L::Node ndNext;                                                                                               \
        CRT_FIRST_NODE_STEP(STEP, (NODE_STEPS > 1))                                                                            \
        _Pragma("unroll") for (int rep = 1; rep < NODE_STEPS; rep++) {                                                         \
            if (L::inner(cur)) {                                                                                               \
                const typename L::Node ndCur = ndNext;                                                                         \
                if (rep + 1 < NODE_STEPS) L::template STEP<COUNT, OCT, true>(ndCur, r, tmin, tcull, stack, cur, cntNodes, nodes, ndNext);   \
                else L::template STEP<COUNT, OCT, false>(ndCur, r, tmin, tcull, stack, cur, cntNodes, nodes, ndNext);           \
            }                                                                                                                  \
        }                                                                                                                      \
    }
#else
#define CRT_NODE_STEPS(STEP)                                                                                                   \
    _Pragma("unroll") for (int rep = 0; rep < NODE_STEPS; rep++) {                                                             \
        if (L::inner(cur)) { typename L::Node ndNext; CRT_FIRST_NODE_STEP(STEP, false) }                                        \
    }
#endif

__device__ __forceinline__ void loadTriUniform(const float4* T, float4& a, float4& b, float4& c)
{
    ConstQuadPtr C = (ConstQuadPtr)(reinterpret_cast<uintptr_t>(T));
    a = quadOf(C[0]); b = quadOf(C[1]); c = quadOf(C[2]);
}

// Wave-level scheduling shared by both traversals.  Every lane walks its own ray in its own fixed order (so results
// and counters do not depend on what the other lanes do), but WHEN a lane's next step runs is decided per wavefront:
// node steps are issued while at least `innerMin` lanes still stand on inner nodes (or nobody waits at a leaf); then the
// lanes waiting at leaves intersect their triangles.  innerMin = 1 is the classic while-while loop (leaves wait until
// every lane has one: 47 % of the lanes active on the 1M-triangle frame); a fixed 32 was the setting of rounds 1 and 2 (first
// measurement: 0.67 vs 1.10 ms).  Round 3: the threshold follows the wavefront's LIVE lanes -- three quarters of them -- because
// late in a packet's life most rays have finished, fewer than 32 lanes are left on inner nodes whatever happens, and a fixed 32
// then serves every single lane that reaches a leaf at once, a whole pass of the wavefront for one or two lanes: primary rays only
// 0.193 -> 0.172 ms, soup 0.256 -> 0.231, the longest packets shorten most (lone launch of an 8-rank share of primary rays 141 -> 126 us).
// One scheduling decision of the closest-hit traversal for the whole wavefront: NODE_STEPS node steps of the lanes standing
// on inner nodes, or the leaf step of the lanes waiting at leaves.  Per-lane state (cur, stack, h, tcull) lives in the
// caller, so a caller may retire finished rays and start new ones between two calls (streamClosest).  Returns false when
// no lane has anything left to do.
template <bool COUNT, class L, int OCT>
__device__ __forceinline__ bool closestIteration(const float4* __restrict__ nodes, const float4* __restrict__ tris, const Ray& r, float tmin,
                                                 float& tcull, Stack& stack, int innerMin, Hit& h, int& cur, uint32_t& iters,
                                                 uint32_t& cntNodes, uint32_t& cntTris)
{
    const unsigned long long innerMask = __ballot(L::inner(cur));
    const unsigned long long leafMask = __ballot(L::leaf(cur));
    if ((innerMask | leafMask) == 0ull) return false;
    if (++iters == kBoostAfter) __builtin_amdgcn_s_setprio(3); // a wavefront on a long critical path stops queueing behind the others
    // innerMin > 0: node steps while at least that many lanes stand on inner nodes; innerMin <= 0 (adaptive): while at least
    // (live lanes * -innerMin) / 8 do (live = lanes with anything left to do), so that a wavefront whose rays have mostly finished
    // does not fall back to serving every single waiting leaf at once
    const int wantNode = innerMin > 0 ? innerMin : (static_cast<int>(__popcll(innerMask | leafMask)) * -innerMin + 7) / 8;
    if (innerMask != 0ull && (leafMask == 0ull || static_cast<int>(__popcll(innerMask)) >= wantNode)) {
#if CRT_PROF
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        CRT_NODE_STEPS(closestStep) // several node steps per scheduling decision: fewer ballots/branches
#if CRT_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stack.tNode += __builtin_amdgcn_s_memtime() - ts0; stack.itNode++; stack.lanesNode += __popcll(innerMask);
#endif
        return true;
    }
#if CRT_PROF
    const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
    stack.itLeaf++; stack.lanesLeaf += __popcll(leafMask);
#endif
    if (L::leaf(cur)) {
        uint32_t first, cnt;
        L::leafRange(cur, first, cnt);
#if UNIFORM_LEAF
        const int lc0 = __builtin_amdgcn_readfirstlane(cur);
        if (__ballot(cur != lc0) == 0ull) {
            // every waiting lane stands on the same leaf: its triangles come through the scalar cache, once per wavefront
            uint32_t ufirst, ucnt;
            L::leafRange(lc0, ufirst, ucnt);
            for (uint32_t i = 0; i < ucnt; i++) {
                const uint32_t id = L::triId(ufirst, i);
                float4 a, b, c;
                loadTriUniform(L::triPtr(tris, id), a, b, c);
                if (COUNT) cntTris++;
                float t, u, v;
                if (triTest(r, a, b, c, tmin, t, u, v)) {
                    const uint32_t gid = __float_as_uint(c.w);
                    if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                        h.t = t; h.u = u; h.v = v; h.tri = id; h.gid = gid;
                        tcull = t * kCullPad;
                    }
                }
            }
        } else
#endif
        {
            CRT_DIV_STATS_LEAF
#if LEAF_PAIRS
            // two triangles per memory round trip (same test order): the second record's loads overlap the first's
            for (uint32_t i = 0; i < cnt; i += 2) {
                const bool two = i + 1 < cnt;
                const uint32_t id = L::triId(first, i), id1 = L::triId(first, two ? i + 1 : i);
                const float4* T = L::triPtr(tris, id);
                const float4* T1 = L::triPtr(tris, id1);
                const float4 a = T[0], b = T[1], c = T[2];
                const float4 a1 = T1[0], b1 = T1[1], c1 = T1[2];
                if (COUNT) cntTris += two ? 2u : 1u;
                float t, u, v;
                if (triTest(r, a, b, c, tmin, t, u, v)) {
                    const uint32_t gid = __float_as_uint(c.w);
                    if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                        h.t = t; h.u = u; h.v = v; h.tri = id; h.gid = gid;
                        tcull = t * kCullPad;
                    }
                }
                if (two & triTest(r, a1, b1, c1, tmin, t, u, v)) {
                    const uint32_t gid = __float_as_uint(c1.w);
                    if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                        h.t = t; h.u = u; h.v = v; h.tri = id1; h.gid = gid;
                        tcull = t * kCullPad;
                    }
                }
            }
#else
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t id = L::triId(first, i);
                const float4* T = L::triPtr(tris, id);
                const float4 a = T[0], b = T[1], c = T[2];
                if (COUNT) cntTris++;
                float t, u, v;
                if (triTest(r, a, b, c, tmin, t, u, v)) {
                    const uint32_t gid = __float_as_uint(c.w);
                    if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                        h.t = t; h.u = u; h.v = v; h.tri = id; h.gid = gid;
                        tcull = t * kCullPad;
                    }
                }
            }
#endif
        }
        cur = stack.sp == 0 ? L::kDone : stack.pop();
    }
#if CRT_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stack.tLeaf += __builtin_amdgcn_s_memtime() - tl0;
#endif
    return true;
}

template <bool COUNT, class L, int OCT>
__device__ __forceinline__ void traceClosestOct(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                             Hit& h, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
    h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.gid = 0;
    int cur = n_nodes ? L::kRoot : L::kDone;
    stack.sp = 0;
    float tcull = tmax * kCullPad; // boxes are culled against best_t * pad; changes only when a hit is accepted
    CRT_UNIFORM_DESCENT(closestStep)
    while (closestIteration<COUNT, L, OCT>(nodes, tris, r, tmin, tcull, stack, innerMin, h, cur, iters, cntNodes, cntTris)) {}
}

// One scheduling decision of the any-hit traversal (see closestIteration); tmax / tcull / occluded are per-lane state of the caller
template <bool COUNT, class L, int OCT>
__device__ __forceinline__ bool anyIteration(const float4* __restrict__ nodes, const float4* __restrict__ tris, const Ray& r, float tmin, float tmax,
                                             float tcull, Stack& stack, int innerMin, bool& occluded, int& cur, uint32_t& iters,
                                             uint32_t& cntNodes, uint32_t& cntTris)
{
    const unsigned long long innerMask = __ballot(L::inner(cur));
    const unsigned long long leafMask = __ballot(L::leaf(cur));
    if ((innerMask | leafMask) == 0ull) return false;
    if (++iters == kBoostAfter) __builtin_amdgcn_s_setprio(3);
    // innerMin > 0: node steps while at least that many lanes stand on inner nodes; innerMin <= 0 (adaptive): while at least
    // (live lanes * -innerMin) / 8 do (live = lanes with anything left to do), so that a wavefront whose rays have mostly finished
    // does not fall back to serving every single waiting leaf at once
    const int wantNode = innerMin > 0 ? innerMin : (static_cast<int>(__popcll(innerMask | leafMask)) * -innerMin + 7) / 8;
    if (innerMask != 0ull && (leafMask == 0ull || static_cast<int>(__popcll(innerMask)) >= wantNode)) {
#if CRT_PROF
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        CRT_NODE_STEPS(anyStep) // several node steps per scheduling decision: fewer ballots/branches
#if CRT_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        stack.tNode += __builtin_amdgcn_s_memtime() - ts0; stack.itNode++; stack.lanesNode += __popcll(innerMask);
#endif
        return true;
    }
#if CRT_PROF
    const unsigned long long tl0 = __builtin_amdgcn_s_memtime();
    stack.itLeaf++; stack.lanesLeaf += __popcll(leafMask);
#endif
    if (L::leaf(cur)) {
        uint32_t first, cnt;
        L::leafRange(cur, first, cnt);
#if UNIFORM_LEAF
        const int lc0 = __builtin_amdgcn_readfirstlane(cur);
        if (__ballot(cur != lc0) == 0ull) {
            uint32_t ufirst, ucnt;
            L::leafRange(lc0, ufirst, ucnt);
            for (uint32_t i = 0; i < ucnt; i++) {
                float4 a, b, c;
                loadTriUniform(L::triPtr(tris, L::triId(ufirst, i)), a, b, c);
                if (COUNT) cntTris++;
                float t, u, v;
                if (triTest(r, a, b, c, tmin, t, u, v) & (t < tmax)) {
                    occluded = true;
                    break;
                }
            }
        } else
#endif
        for (uint32_t i = 0; i < cnt; i++) {
            const float4* T = L::triPtr(tris, L::triId(first, i));
            const float4 a = T[0], b = T[1], c = T[2];
            if (COUNT) cntTris++;
            float t, u, v;
            if (triTest(r, a, b, c, tmin, t, u, v) & (t < tmax)) {
                occluded = true;
                break;
            }
        }
        cur = (occluded | (stack.sp == 0)) ? L::kDone : stack.pop();
    }
#if CRT_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stack.tLeaf += __builtin_amdgcn_s_memtime() - tl0;
#endif
    return true;
}

template <bool COUNT, class L, int OCT>
__device__ __forceinline__ bool traceAnyOct(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                         uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                         uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
    bool occluded = false;
    int cur = n_nodes ? L::kRoot : L::kDone;
    stack.sp = 0;
    const float tcull = tmax * kCullPad;
    CRT_UNIFORM_DESCENT(anyStep)
    while (anyIteration<COUNT, L, OCT>(nodes, tris, r, tmin, tmax, tcull, stack, innerMin, occluded, cur, iters, cntNodes, cntTris)) {}
    return occluded;
}

// Pick the traversal loop specialised for the wavefront's direction octant when all its active lanes share one (nearly
// every 8x8 camera packet and every packet of shadow rays towards one light does); otherwise the generic loop.
__device__ __forceinline__ uint32_t octantOf(const Ray& r)
{
    return (__float_as_uint(r.d.x) >> 31) | ((__float_as_uint(r.d.y) >> 31) << 1) | ((__float_as_uint(r.d.z) >> 31) << 2);
}

template <bool COUNT, class L>
__device__ __forceinline__ void traceClosest(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                             Hit& h, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
#if OCTANT_SPECIALISE
    const uint32_t oct = octantOf(r);
    const uint32_t o0 = __builtin_amdgcn_readfirstlane(oct);
    if (__ballot(oct != o0) == 0ull) {
        switch (o0) {
#define CRT_CASE(k) case k: traceClosestOct<COUNT, L, k>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, h, iters, cntNodes, cntTris); return;
            CRT_CASE(0) CRT_CASE(1) CRT_CASE(2) CRT_CASE(3) CRT_CASE(4) CRT_CASE(5) CRT_CASE(6) CRT_CASE(7)
#undef CRT_CASE
        }
    }
#endif
    traceClosestOct<COUNT, L, 8>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, h, iters, cntNodes, cntTris);
}

template <bool COUNT, class L>
__device__ __forceinline__ bool traceAny(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                         uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                         uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
#if OCTANT_SPECIALISE
    const uint32_t oct = octantOf(r);
    const uint32_t o0 = __builtin_amdgcn_readfirstlane(oct);
    if (__ballot(oct != o0) == 0ull) {
        switch (o0) {
#define CRT_CASE(k) case k: return traceAnyOct<COUNT, L, k>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, iters, cntNodes, cntTris);
            CRT_CASE(0) CRT_CASE(1) CRT_CASE(2) CRT_CASE(3) CRT_CASE(4) CRT_CASE(5) CRT_CASE(6) CRT_CASE(7)
#undef CRT_CASE
        }
    }
#endif
    return traceAnyOct<COUNT, L, 8>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, iters, cntNodes, cntTris);
}

} // namespace
} // namespace crt
