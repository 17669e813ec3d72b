// HIP kernels for gfx950 (MI355X): the reference's per-pixel DXR program -- rayGen -> TraceRay -> miss /
// closestHit -> image store (R/HLSL/ray_tracing_shaders.hlsl:21-169, dispatched by DispatchRays at
// R/DXRTRenderer.cpp:1405) -- fused into one kernel, with the BVH traversal and the ray/triangle test the
// reference leaves to the DXR driver written out.
//
// Mapping: one 64-thread workgroup = one wavefront = one 8x8-pixel ray packet (the work unit); four units form the
// 16x16 macro tile that is the unit of multi-GPU ownership.  Every lane owns one ray and a private traversal stack
// in LDS (entry e of lane l at dword e*64+l: conflict free, 24 entries = 6 KB per wavefront; deeper entries spill
// to a global arena).  The tree is 4-wide; a node is fetched per lane with dwordx4 loads, or once per wavefront
// through the scalar cache when all lanes stand on the same node; triangles are 48-byte records.  Units are
// launched most-expensive-first from the cost the previous frame measured.
//
// Arithmetic contract: identical, operation for operation, to oracle/crt_oracle.c (compiled with
// -ffp-contract=off; fused multiply-adds only where fmaf()/fma() is written; correctly rounded / and sqrt).
#include "render_kernels.h"

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
constexpr int kDone = INT_MIN;    // traversal finished (not a valid leaf reference)
constexpr int kBlock = 256;
constexpr uint32_t kBoostAfter = 300;
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
// Register budget: the primary/shadow-ray variant is asked for 7 wavefronts per SIMD (<= 72 VGPRs; one register spilled
// outside the loops).  With the 64-byte quantised nodes a node in flight is 16 registers instead of 28, and with the LDS
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
#ifndef OCTANT_SPECIALISE
#define OCTANT_SPECIALISE 1
#endif // traversal-loop iterations after which a wavefront raises its issue priority
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
// is chosen so that 26 wavefronts per CU fit its 160 KB (24 entries = 6 KB per wavefront).  No ray of the test scenes ever holds more than 15
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
    uint32_t tri; // leaf-order triangle index
    uint32_t gid;
};

// ---- wide (4-child) node step.  Node = crt_bvh_node4q, 64 bytes = four dwordx4 loads:
//   {lo.x lo.y lo.z s.x} {s.y s.z qlo_x qhi_x} {qlo_y qhi_y qlo_z qhi_z} {ref[4]}
// The child planes are 8-bit offsets from the node's own minimum corner (byte k of a q word = child k): plane =
// fma(q, s, lo).  The vector memory pipe -- per-lane fetch requests -- is what bounds this kernel, not vector arithmetic
// (30 extra dependent VALU per step measured +0.5 %, one extra 4-byte touch per pushed child +29 %), so the record is
// kept to four requests per lane instead of the seven of a full-precision node and decoded in registers.  The decode is
// folded into the slab test: t(q) = fma(q, s * idir, fma(lo, idir, -o * idir)), monotonic in q with the sign of idir, so
// for a known direction octant (OCT < 8) the near plane of each axis is a fixed member of the (qlo, qhi) pair and the
// min/max pairs of the generic form (OCT = 8) disappear -- bit for bit the same values.
// One memory round trip yields four slab tests (15.6 instead of 29.7 steps per ray on the 1M-triangle frame).
// An unused child slot has ref CRT_BVH_EMPTY (tested explicitly).
constexpr int kEmptyRef = INT_MIN;

// one wide node in registers; fetched per lane (four dwordx4 vector loads) or, when the whole wavefront stands on the
// same node, once through the scalar cache (constant address space + wave-uniform address = one s_load_dwordx16)
struct NodeRegs {
    float4 q0, q1, q2;
    int4 refs;
};
constexpr size_t kNodeQuads = 4; // float4 per node record

__device__ __forceinline__ float ubyteToFloat(uint32_t w, int k) { return static_cast<float>((w >> (8 * k)) & 0xFFu); } // v_cvt_f32_ubyteK

template <int OCT>
__device__ __forceinline__ void slab4(const NodeRegs& nd, const Ray& r, float tmin, float tcull, float tn[4], bool hit[4])
{
    const float ax = nd.q0.w * r.idir.x, ay = nd.q1.x * r.idir.y, az = nd.q1.y * r.idir.z;
    const float bx = fmaf(nd.q0.x, r.idir.x, r.noid.x), by = fmaf(nd.q0.y, r.idir.y, r.noid.y), bz = fmaf(nd.q0.z, r.idir.z, r.noid.z);
    const uint32_t lx = __float_as_uint(nd.q1.z), hx = __float_as_uint(nd.q1.w), ly = __float_as_uint(nd.q2.x), hy = __float_as_uint(nd.q2.y),
                   lz = __float_as_uint(nd.q2.z), hz = __float_as_uint(nd.q2.w);
    const int rf[4] = { nd.refs.x, nd.refs.y, nd.refs.z, nd.refs.w };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float t_n, t_f;
        if (OCT < 8) {
            const float nx = ubyteToFloat((OCT & 1) ? hx : lx, k), fx = ubyteToFloat((OCT & 1) ? lx : hx, k);
            const float ny = ubyteToFloat((OCT & 2) ? hy : ly, k), fy = ubyteToFloat((OCT & 2) ? ly : hy, k);
            const float nz = ubyteToFloat((OCT & 4) ? hz : lz, k), fz = ubyteToFloat((OCT & 4) ? lz : hz, k);
            t_n = fmaxf(fmaxf(fmaf(nx, ax, bx), fmaf(ny, ay, by)), fmaxf(fmaf(nz, az, bz), tmin));
            t_f = fminf(fminf(fmaf(fx, ax, bx), fmaf(fy, ay, by)), fminf(fmaf(fz, az, bz), tcull));
        } else {
            const float x0 = fmaf(ubyteToFloat(lx, k), ax, bx), x1 = fmaf(ubyteToFloat(hx, k), ax, bx);
            const float y0 = fmaf(ubyteToFloat(ly, k), ay, by), y1 = fmaf(ubyteToFloat(hy, k), ay, by);
            const float z0 = fmaf(ubyteToFloat(lz, k), az, bz), z1 = fmaf(ubyteToFloat(hz, k), az, bz);
            t_n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
            t_f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tcull));
        }
        tn[k] = t_n;
        hit[k] = (t_n <= t_f) & (rf[k] != kEmptyRef);
    }
}

__device__ __forceinline__ NodeRegs loadNode(const float4* __restrict__ N)
{
    NodeRegs nd;
    nd.q0 = N[0]; nd.q1 = N[1]; nd.q2 = N[2];
    nd.refs = *reinterpret_cast<const int4*>(N + 3);
    return nd;
}
__device__ __forceinline__ __attribute__((unused)) NodeRegs loadNodeUniform(const float4* N)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4v* ConstPtr;
    ConstPtr C = (ConstPtr)(reinterpret_cast<uintptr_t>(N));
    const f4v a = C[0], b = C[1], c = C[2], g = C[3];
    NodeRegs nd;
    nd.q0 = make_float4(a.x, a.y, a.z, a.w); nd.q1 = make_float4(b.x, b.y, b.z, b.w); nd.q2 = make_float4(c.x, c.y, c.z, c.w);
    nd.refs = make_int4(__float_as_int(g.x), __float_as_int(g.y), __float_as_int(g.z), __float_as_int(g.w));
    return nd;
}

__device__ __forceinline__ __attribute__((unused)) void loadTriUniform(const float4* T, float4& a, float4& b, float4& c)
{
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4v* ConstPtr;
    ConstPtr C = (ConstPtr)(reinterpret_cast<uintptr_t>(T));
    const f4v x = C[0], y = C[1], z = C[2];
    a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w); c = make_float4(z.x, z.y, z.z, z.w);
}

__device__ __forceinline__ int pick4(const int4& v, uint32_t i) // v[i], i in 0..3, without dynamic register indexing
{
    const int lo = (i & 1u) ? v.y : v.x, hi = (i & 1u) ? v.w : v.z;
    return (i & 2u) ? hi : lo;
}

// closest hit: visit the hit children nearest first.  Order key = (bits(t_near) & 0x7FFFFFFC) | slot: t_near >= 0 so its
// bit pattern orders like the float, the two low bits hold the slot (keys are unique, order is total and identical in
// the oracle); misses get 0xFFFFFFFF.  Five min/max pairs sort the four keys.
template <bool COUNT, int BLOCK, int OCT>
__device__ __forceinline__ void nodeStepClosestAt(const NodeRegs& nd, const Ray& r, float tmin, float tcull, Stack& stack,
                                                  int& cur, uint32_t& cntNodes)
{
    const int4 refs = nd.refs;
    if (COUNT) cntNodes++;
    float tn[4];
    bool hit[4];
    slab4<OCT>(nd, r, tmin, tcull, tn, hit);
    uint32_t key[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        key[k] = hit[k] ? ((__float_as_uint(tn[k]) & 0x7FFFFFFCu) | static_cast<uint32_t>(k)) : 0xFFFFFFFFu;
#define CRT_CSWAP(a, b) { const uint32_t lo = min(key[a], key[b]), hi = max(key[a], key[b]); key[a] = lo; key[b] = hi; }
    CRT_CSWAP(0, 1) CRT_CSWAP(2, 3) CRT_CSWAP(0, 2) CRT_CSWAP(1, 3) CRT_CSWAP(1, 2)
#undef CRT_CSWAP
    if (key[0] == 0xFFFFFFFFu) {
        cur = stack.sp == 0 ? kDone : stack.pop();
    } else {
        if (key[3] != 0xFFFFFFFFu) stack.push(pick4(refs, key[3] & 3u)); // farthest first: the nearest pending child pops first
        if (key[2] != 0xFFFFFFFFu) stack.push(pick4(refs, key[2] & 3u));
        if (key[1] != 0xFFFFFFFFu) stack.push(pick4(refs, key[1] & 3u));
        cur = pick4(refs, key[0] & 3u);
    }
}

// any hit: order independent, children taken in slot order
template <bool COUNT, int BLOCK, int OCT>
__device__ __forceinline__ void nodeStepAnyAt(const NodeRegs& nd, const Ray& r, float tmin, float tcull, Stack& stack,
                                              int& cur, uint32_t& cntNodes)
{
    const int4 refs = nd.refs;
    if (COUNT) cntNodes++;
    float tn[4];
    bool hit[4];
    slab4<OCT>(nd, r, tmin, tcull, tn, hit);
    const bool h0 = hit[0], h1 = hit[1], h2 = hit[2], h3 = hit[3];
    if (!(h0 | h1 | h2 | h3)) {
        cur = stack.sp == 0 ? kDone : stack.pop();
    } else {
        // first hit slot becomes current; later hit slots are pushed, last slot first
        if (h3 & (h0 | h1 | h2)) stack.push(refs.w);
        if (h2 & (h0 | h1)) stack.push(refs.z);
        if (h1 & h0) stack.push(refs.y);
        cur = h0 ? refs.x : (h1 ? refs.y : (h2 ? refs.z : refs.w));
    }
}

// Uniform descent: the rays of an 8x8 packet start at the root and usually agree on the first few nodes.  While every
// active lane stands on the SAME inner node its record is fetched once through the scalar cache (the node address is
// wave-uniform, so the loads become s_load) instead of 64 identical per-lane vector fetches; each lane still runs its own
// slab tests, ordering and pushes, so results and counters are exactly those of the per-lane loop that follows.
#if UNIFORM_DESCENT
#define CRT_UNIFORM_DESCENT(STEP)                                                                                              \
    for (;;) {                                                                                                                 \
        const int c0 = __builtin_amdgcn_readfirstlane(cur);                                                                    \
        if (c0 < 0 || __ballot(cur != c0) != 0ull) break;                                                                      \
        STEP<COUNT, BLOCK, OCT>(loadNodeUniform(nodes + kNodeQuads * static_cast<size_t>(c0)), r, tmin, tcull, stack, cur, cntNodes);   \
    }
#else
#define CRT_UNIFORM_DESCENT(STEP)
#endif
// One node step of the lanes standing on inner nodes (called with exactly those lanes active): through the scalar cache
// when they all stand on the same node, per lane otherwise.
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
#if UNIFORM_STEP
#define CRT_NODE_STEP(STEP)                                                                                                    \
    {                                                                                                                          \
        const int c0 = __builtin_amdgcn_readfirstlane(cur);                                                                    \
        if (__ballot(cur != c0) == 0ull) {                                                                                     \
            STEP<COUNT, BLOCK, OCT>(loadNodeUniform(nodes + kNodeQuads * static_cast<size_t>(c0)), r, tmin, tcull, stack, cur, cntNodes); \
        } else {                                                                                                               \
            CRT_DIV_STATS_NODE                                                                                                 \
            STEP<COUNT, BLOCK, OCT>(loadNode(nodes + kNodeQuads * static_cast<size_t>(cur)), r, tmin, tcull, stack, cur, cntNodes);     \
        }                                                                                                                      \
    }
#else
#define CRT_NODE_STEP(STEP) STEP<COUNT, BLOCK, OCT>(loadNode(nodes + kNodeQuads * static_cast<size_t>(cur)), r, tmin, tcull, stack, cur, cntNodes);
#endif

// Wave-level scheduling shared by both traversals.  Every lane walks its own ray in its own fixed order (so results
// and counters do not depend on what the other lanes do), but WHEN a lane's next step runs is decided per wavefront:
// node steps are issued while at least `innerMin` lanes still stand on inner nodes (or nobody waits at a leaf); then the
// lanes waiting at leaves intersect their triangles.  innerMin = 1 is the classic while-while loop (leaves wait until
// every lane has one: 47 % of the lanes active on the 1M-triangle frame); 32 measured best (first measurement: 0.67 vs 1.10 ms; re-swept after every structural change).
// One scheduling decision of the closest-hit traversal for the whole wavefront: NODE_STEPS node steps of the lanes standing
// on inner nodes, or the leaf step of the lanes waiting at leaves.  Per-lane state (cur, stack, h, tcull) lives in the
// caller, so a caller may retire finished rays and start new ones between two calls (streamClosest).  Returns false when
// no lane has anything left to do.
template <bool COUNT, int BLOCK, int OCT>
__device__ __forceinline__ bool closestIteration(const float4* __restrict__ nodes, const float4* __restrict__ tris, const Ray& r, float tmin,
                                                 float& tcull, Stack& stack, int innerMin, Hit& h, int& cur, uint32_t& iters,
                                                 uint32_t& cntNodes, uint32_t& cntTris)
{
    const unsigned long long innerMask = __ballot(cur >= 0);
    const unsigned long long leafMask = __ballot((cur < 0) & (cur != kDone));
    if ((innerMask | leafMask) == 0ull) return false;
    if (++iters == kBoostAfter) __builtin_amdgcn_s_setprio(3); // a wavefront on a long critical path stops queueing behind the others
    if (innerMask != 0ull && (leafMask == 0ull || static_cast<int>(__popcll(innerMask)) >= innerMin)) {
#if CRT_PROF
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int rep = 0; rep < NODE_STEPS; rep++) { // several node steps per scheduling decision: fewer ballots/branches
            if (cur >= 0) CRT_NODE_STEP(nodeStepClosestAt)
        }
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
    if ((cur < 0) & (cur != kDone)) {
        const uint32_t code = static_cast<uint32_t>(~cur);
        const uint32_t first = code >> 3, cnt = code & 7u;
#if UNIFORM_LEAF
        const int lc0 = __builtin_amdgcn_readfirstlane(cur);
        if (__ballot(cur != lc0) == 0ull) {
            // every waiting lane stands on the same leaf: its triangles come through the scalar cache, once per wavefront
            const uint32_t ucode = static_cast<uint32_t>(~lc0);
            const uint32_t ufirst = ucode >> 3, ucnt = ucode & 7u;
            for (uint32_t i = ufirst; i < ufirst + ucnt; i++) {
                float4 a, b, c;
                loadTriUniform(tris + 3 * static_cast<size_t>(i), a, b, c);
                if (COUNT) cntTris++;
                float t, u, v;
                if (triTest(r, a, b, c, tmin, t, u, v)) {
                    const uint32_t gid = __float_as_uint(c.w);
                    if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                        h.t = t; h.u = u; h.v = v; h.tri = i; h.gid = gid;
                        tcull = t * kCullPad;
                    }
                }
            }
        } else
#endif
#if LEAF_PAIRS
        CRT_DIV_STATS_LEAF
        // two triangles per memory round trip (same test order): the second record's loads overlap the first's
        for (uint32_t i = first; i < first + cnt; i += 2) {
            const bool two = i + 1 < first + cnt;
            const float4* T = tris + 3 * static_cast<size_t>(i);
            const float4* T1 = two ? T + 3 : T;
            const float4 a = T[0], b = T[1], c = T[2];
            const float4 a1 = T1[0], b1 = T1[1], c1 = T1[2];
            if (COUNT) cntTris += two ? 2u : 1u;
            float t, u, v;
            if (triTest(r, a, b, c, tmin, t, u, v)) {
                const uint32_t gid = __float_as_uint(c.w);
                if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                    h.t = t; h.u = u; h.v = v; h.tri = i; h.gid = gid;
                    tcull = t * kCullPad;
                }
            }
            if (two & triTest(r, a1, b1, c1, tmin, t, u, v)) {
                const uint32_t gid = __float_as_uint(c1.w);
                if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                    h.t = t; h.u = u; h.v = v; h.tri = i + 1; h.gid = gid;
                    tcull = t * kCullPad;
                }
            }
        }
#else
        for (uint32_t i = first; i < first + cnt; i++) {
            const float4* T = tris + 3 * static_cast<size_t>(i);
            const float4 a = T[0], b = T[1], c = T[2];
            if (COUNT) cntTris++;
            float t, u, v;
            if (triTest(r, a, b, c, tmin, t, u, v)) {
                const uint32_t gid = __float_as_uint(c.w);
                if ((t < h.t) | ((t == h.t) & (gid < h.gid))) {
                    h.t = t; h.u = u; h.v = v; h.tri = i; h.gid = gid;
                    tcull = t * kCullPad;
                }
            }
        }
#endif
        cur = stack.sp == 0 ? kDone : stack.pop();
    }
#if CRT_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stack.tLeaf += __builtin_amdgcn_s_memtime() - tl0;
#endif
    return true;
}

template <bool COUNT, int BLOCK, int OCT>
__device__ __forceinline__ void traceClosestOct(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                             Hit& h, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
    h.t = tmax; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.gid = 0;
    int cur = n_nodes ? 0 : kDone;
    stack.sp = 0;
    float tcull = tmax * kCullPad; // boxes are culled against best_t * pad; changes only when a hit is accepted
    CRT_UNIFORM_DESCENT(nodeStepClosestAt)
    while (closestIteration<COUNT, BLOCK, OCT>(nodes, tris, r, tmin, tcull, stack, innerMin, h, cur, iters, cntNodes, cntTris)) {}
}

// One scheduling decision of the any-hit traversal (see closestIteration); tmax / tcull / occluded are per-lane state of the caller
template <bool COUNT, int BLOCK, int OCT>
__device__ __forceinline__ bool anyIteration(const float4* __restrict__ nodes, const float4* __restrict__ tris, const Ray& r, float tmin, float tmax,
                                             float tcull, Stack& stack, int innerMin, bool& occluded, int& cur, uint32_t& iters,
                                             uint32_t& cntNodes, uint32_t& cntTris)
{
    const unsigned long long innerMask = __ballot(cur >= 0);
    const unsigned long long leafMask = __ballot((cur < 0) & (cur != kDone));
    if ((innerMask | leafMask) == 0ull) return false;
    if (++iters == kBoostAfter) __builtin_amdgcn_s_setprio(3);
    if (innerMask != 0ull && (leafMask == 0ull || static_cast<int>(__popcll(innerMask)) >= innerMin)) {
#if CRT_PROF
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int rep = 0; rep < NODE_STEPS; rep++) {
            if (cur >= 0) CRT_NODE_STEP(nodeStepAnyAt)
        }
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
    if ((cur < 0) & (cur != kDone)) {
        const uint32_t code = static_cast<uint32_t>(~cur);
        const uint32_t first = code >> 3, cnt = code & 7u;
#if UNIFORM_LEAF
        const int lc0 = __builtin_amdgcn_readfirstlane(cur);
        if (__ballot(cur != lc0) == 0ull) {
            const uint32_t ucode = static_cast<uint32_t>(~lc0);
            const uint32_t ufirst = ucode >> 3, ucnt = ucode & 7u;
            for (uint32_t i = ufirst; i < ufirst + ucnt; i++) {
                float4 a, b, c;
                loadTriUniform(tris + 3 * static_cast<size_t>(i), a, b, c);
                if (COUNT) cntTris++;
                float t, u, v;
                if (triTest(r, a, b, c, tmin, t, u, v) & (t < tmax)) {
                    occluded = true;
                    break;
                }
            }
        } else
#endif
        for (uint32_t i = first; i < first + cnt; i++) {
            const float4* T = tris + 3 * static_cast<size_t>(i);
            const float4 a = T[0], b = T[1], c = T[2];
            if (COUNT) cntTris++;
            float t, u, v;
            if (triTest(r, a, b, c, tmin, t, u, v) & (t < tmax)) {
                occluded = true;
                break;
            }
        }
        cur = (occluded | (stack.sp == 0)) ? kDone : stack.pop();
    }
#if CRT_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stack.tLeaf += __builtin_amdgcn_s_memtime() - tl0;
#endif
    return true;
}

template <bool COUNT, int BLOCK, int OCT>
__device__ __forceinline__ bool traceAnyOct(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                         uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                         uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
    bool occluded = false;
    int cur = n_nodes ? 0 : kDone;
    stack.sp = 0;
    const float tcull = tmax * kCullPad;
    CRT_UNIFORM_DESCENT(nodeStepAnyAt)
    while (anyIteration<COUNT, BLOCK, OCT>(nodes, tris, r, tmin, tmax, tcull, stack, innerMin, occluded, cur, iters, cntNodes, cntTris)) {}
    return occluded;
}

// Pick the traversal loop specialised for the wavefront's direction octant when all its active lanes share one (nearly
// every 8x8 camera packet and every packet of shadow rays towards one light does); otherwise the generic loop.
__device__ __forceinline__ uint32_t octantOf(const Ray& r)
{
    return (__float_as_uint(r.d.x) >> 31) | ((__float_as_uint(r.d.y) >> 31) << 1) | ((__float_as_uint(r.d.z) >> 31) << 2);
}

template <bool COUNT, int BLOCK>
__device__ __forceinline__ void traceClosest(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                             uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                             Hit& h, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
#if OCTANT_SPECIALISE
    const uint32_t oct = octantOf(r);
    const uint32_t o0 = __builtin_amdgcn_readfirstlane(oct);
    if (__ballot(oct != o0) == 0ull) {
        switch (o0) {
#define CRT_CASE(k) case k: traceClosestOct<COUNT, BLOCK, k>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, h, iters, cntNodes, cntTris); return;
            CRT_CASE(0) CRT_CASE(1) CRT_CASE(2) CRT_CASE(3) CRT_CASE(4) CRT_CASE(5) CRT_CASE(6) CRT_CASE(7)
#undef CRT_CASE
        }
    }
#endif
    traceClosestOct<COUNT, BLOCK, 8>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, h, iters, cntNodes, cntTris);
}

template <bool COUNT, int BLOCK>
__device__ __forceinline__ bool traceAny(const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                         uint32_t n_nodes, const Ray& r, float tmin, float tmax, Stack& stack, int innerMin,
                                         uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
#if OCTANT_SPECIALISE
    const uint32_t oct = octantOf(r);
    const uint32_t o0 = __builtin_amdgcn_readfirstlane(oct);
    if (__ballot(oct != o0) == 0ull) {
        switch (o0) {
#define CRT_CASE(k) case k: return traceAnyOct<COUNT, BLOCK, k>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, iters, cntNodes, cntTris);
            CRT_CASE(0) CRT_CASE(1) CRT_CASE(2) CRT_CASE(3) CRT_CASE(4) CRT_CASE(5) CRT_CASE(6) CRT_CASE(7)
#undef CRT_CASE
        }
    }
#endif
    return traceAnyOct<COUNT, BLOCK, 8>(nodes, tris, n_nodes, r, tmin, tmax, stack, innerMin, iters, cntNodes, cntTris);
}

// rayGen (hlsl:21-55) with width/height as parameters instead of the literals 1920/1080 (hlsl:24-25)
__device__ __forceinline__ F3 rayDirJ(const float* rot, uint32_t px, uint32_t py, float jx, float jy, float width, float height)
{
    float x = static_cast<float>(px), y = static_cast<float>(py);
    x += jx; // 0.5 in the reference (hlsl:35-36); the path tracer jitters inside the pixel
    y += jy;
    x /= width;
    y /= height;
    x = (2.0f * x) - 1.0f;
    y = 1.0f - (2.0f * y);
    x *= width / height;
    const F3 dc = normalize3(f3(x, y, -1.0f));
    const F3 dw = f3(dot3(f3(rot[0], rot[1], rot[2]), dc), dot3(f3(rot[3], rot[4], rot[5]), dc),
                     dot3(f3(rot[6], rot[7], rot[8]), dc));
    return normalize3(dw);
}

__device__ __forceinline__ F3 rayDir(const float* rot, uint32_t px, uint32_t py, float width, float height)
{
    return rayDirJ(rot, px, py, 0.5f, 0.5f, width, height);
}

__device__ __forceinline__ F3 objectBaseColour(uint32_t inst) // hlsl:97-101,117-121
{
    const float f = static_cast<float>(inst);
    return f3(hashSin(f * 12.9898f, 43758.5453f), hashSin(f * 78.233f, 12345.6789f), hashSin(f * 39.425f, 34567.8901f));
}

// closestHit, modes 0..6 (hlsl:78-169)
__device__ __forceinline__ F3 shadeDebug(uint32_t mode, uint32_t inst, uint32_t prim, float t, float u, float v, F3 o, F3 d)
{
    const F3 wp = f3(o.x + d.x * t, o.y + d.y * t, o.z + d.z * t);
    if (mode == 0) {
        const float f = static_cast<float>(prim);
        return f3(hashSin(f * 12.9898f, 43758.5453f), hashSin(f * 78.233f, 43758.5453f), hashSin(f * 45.164f, 43758.5453f));
    }
    if (mode == 1) {
        const F3 base = objectBaseColour(inst);
        const int cx = static_cast<int>(floorf(wp.x / 2.0f)), cy = static_cast<int>(floorf(wp.y / 2.0f)),
                  cz = static_cast<int>(floorf(wp.z / 2.0f));
        const uint32_t hash = (static_cast<uint32_t>(cx) * 73856093u) ^ (static_cast<uint32_t>(cy) * 19349663u) ^
                              (static_cast<uint32_t>(cz) * 83492791u);
        const float variation = hashSin(static_cast<float>(hash) * 12.9898f, 43758.5453f);
        return f3(lerp1(base.x * 0.7f, base.x * 1.3f, variation), lerp1(base.y * 0.7f, base.y * 1.3f, variation),
                  lerp1(base.z * 0.7f, base.z * 1.3f, variation));
    }
    if (mode == 2) {
        const F3 base = objectBaseColour(inst);
        const float shade = hashSin(static_cast<float>(prim) * 12.9898f, 43758.5453f);
        const float k = lerp1(0.6f, 1.0f, shade);
        return f3(base.x * k, base.y * k, base.z * k);
    }
    if (mode == 3) return f3(1.0f - u - v, u, v);
    if (mode == 4) {
        const float h = saturate1((wp.y + 10.0f) / 20.0f);
        return f3(lerp1(0.1f, 0.9f, h), lerp1(0.2f, 0.9f, h), lerp1(0.6f, 0.9f, h));
    }
    if (mode == 5) {
        const float c = saturate1(t * 0.05f);
        return f3(c, c, c);
    }
    const int checker = (static_cast<int>(floorf(wp.x)) ^ static_cast<int>(floorf(wp.z))) & 1;
    const float c = checker ? 0.9f : 0.2f;
    return f3(c, c, c);
}

struct LightRec { float x, y, z, intensity; };
struct MaterialRec { float r, g, b; uint32_t type; uint32_t smooth; float ior; int texture; };

// CRTTexture::getColor restated (R/CRTTexture*.cpp; oracle: texture_color)
__device__ __forceinline__ F3 textureColor(const TextureRec& t, const unsigned char* texels, float u, float v)
{
    const F3 A = f3(t.a[0], t.a[1], t.a[2]), B = f3(t.b[0], t.b[1], t.b[2]);
    if (t.type == 1u) return (u < t.scalar || v < t.scalar || (1.0f - u - v) < t.scalar) ? A : B; // edges
    if (t.type == 2u) { // checker
        const int width = static_cast<int>(1.0f / t.scalar);
        const int u2 = static_cast<int>(floorf(u * static_cast<float>(width)));
        const int v2 = static_cast<int>(floorf(v * static_cast<float>(width)));
        return ((u2 + v2) % 2 == 0) ? A : B;
    }
    if (t.type == 3u) { // bitmap: nearest texel, v flipped
        if (t.channels < 3u || t.width == 0u) return f3(0.0f, 0.0f, 0.0f);
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        const int row = static_cast<int>((1.0f - v) * static_cast<float>(static_cast<int>(t.height) - 1));
        const int col = static_cast<int>(u * static_cast<float>(static_cast<int>(t.width) - 1));
        const unsigned char* px = texels + t.texel_offset + (static_cast<size_t>(row) * t.width + static_cast<size_t>(col)) * t.channels;
        return f3(static_cast<float>(px[0]) / 255.0f, static_cast<float>(px[1]) / 255.0f, static_cast<float>(px[2]) / 255.0f);
    }
    return A; // albedo texture
}

// Surface at a closest hit (oracle: surface_at): hit point, shading normal flipped to face the ray, material
struct Surface {
    F3 P, N, albedo;
    uint32_t mtype;
    bool entering;
    float ior;
};

__device__ __forceinline__ Surface surfaceAt(const RenderParams& p, const float4* tris, const Ray& r, const Hit& h)
{
    Surface sf;
    const float4* T = tris + 3 * static_cast<size_t>(h.tri);
    const float4 tb = T[1], tc = T[2];
    const float* S = reinterpret_cast<const float*>(p.shade) + 12 * static_cast<size_t>(h.tri);
    const uint32_t material = __float_as_uint(S[9]);
    sf.P = f3(r.o.x + r.d.x * h.t, r.o.y + r.d.y * h.t, r.o.z + r.d.z * h.t);
    sf.albedo = f3(1.0f, 1.0f, 1.0f);
    sf.mtype = 1u;
    sf.ior = 1.0f;
    bool smooth = false;
    if (material < p.n_mats) {
        const MaterialRec* M = reinterpret_cast<const MaterialRec*>(p.mats) + material;
        sf.albedo = f3(M->r, M->g, M->b);
        smooth = M->smooth != 0;
        sf.mtype = M->type;
        sf.ior = M->ior;
        if (M->texture >= 0 && static_cast<uint32_t>(M->texture) < p.n_textures) {
            // CRTMaterial::isTexture: albedo from the texture; edges on the hit's barycentrics, the rest on the mesh uvs
            const TextureRec tx = reinterpret_cast<const TextureRec*>(p.textures)[M->texture];
            float tu = h.u, tv = h.v;
            if (tx.type != 1u) {
                tu = 0.0f;
                tv = 0.0f;
                if (p.uvs) {
                    const float* U = reinterpret_cast<const float*>(p.uvs) + 6 * static_cast<size_t>(h.tri);
                    const float w = 1.0f - h.u - h.v;
                    tu = fmaf(U[4], h.v, fmaf(U[2], h.u, U[0] * w));
                    tv = fmaf(U[5], h.v, fmaf(U[3], h.u, U[1] * w));
                }
            }
            sf.albedo = textureColor(tx, p.texels, tu, tv);
        }
    }
    F3 N = cross3(f3(tb.x, tb.y, tb.z), f3(tc.x, tc.y, tc.z));
    if (smooth) {
        const float w = 1.0f - h.u - h.v;
        const F3 Ns = f3(fmaf(S[6], h.v, fmaf(S[3], h.u, S[0] * w)), fmaf(S[7], h.v, fmaf(S[4], h.u, S[1] * w)),
                         fmaf(S[8], h.v, fmaf(S[5], h.u, S[2] * w)));
        if (dot3(Ns, Ns) > 0.0f) N = Ns;
    }
    N = normalize3(N);
    sf.entering = true;
    if (dot3(N, r.d) > 0.0f) {
        N = f3(-N.x, -N.y, -N.z);
        sf.entering = false;
    }
    sf.N = N;
    return sf;
}

__device__ __forceinline__ F3 biasPoint(F3 P, F3 N, float bias)
{
    return f3(fmaf(N.x, bias, P.x), fmaf(N.y, bias, P.y), fmaf(N.z, bias, P.z));
}

// x^n by square and multiply in the oracle's order (pow_uint)
__device__ __forceinline__ float powUint(float x, uint32_t n)
{
    float result = 1.0f, base = x;
    while (n) {
        if (n & 1u) result *= base;
        base *= base;
        n >>= 1;
    }
    return result;
}

// direct light at Po: one any-hit shadow ray per light with a positive cosine (oracle: direct_light).  PHONG (mode 100
// only): plus the specular term ks * I / (4 pi r^2) * max(0, R . view)^n, R = the light direction mirrored about N.
template <bool COUNT, int BLOCK, bool PHONG>
__device__ __forceinline__ F3 directLight(const RenderParams& p, const float4* nodes, const float4* tris, F3 Po, F3 N, F3 albedo, F3 view,
                                          Stack& stack, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow)
{
    F3 rgb = f3(0.0f, 0.0f, 0.0f);
    const LightRec* lights = reinterpret_cast<const LightRec*>(p.lights);
    for (uint32_t li = 0; li < p.n_lights; li++) {
        const LightRec L = lights[li];
        const F3 Lv = sub3(f3(L.x, L.y, L.z), Po);
        const float r2 = dot3(Lv, Lv);
        const float dist = sqrtf(r2);
        const float invr = 1.0f / dist;
        const F3 Ld = f3(Lv.x * invr, Lv.y * invr, Lv.z * invr);
        const float cosv = fmaxf(0.0f, dot3(N, Ld));
        if (cosv > 0.0f) {
            const Ray sr = makeRay(Po, Ld);
            if (COUNT) cntShadow++;
            const bool occluded = traceAny<COUNT, BLOCK>(nodes, tris, p.n_nodes, sr, 0.0f, dist, stack, static_cast<int>(p.tune_inner_min), iters, cntNodes, cntTris);
            if (!occluded) {
                const float k = (L.intensity / (kFourPi * r2)) * cosv;
                rgb.x = fmaf(albedo.x, k, rgb.x);
                rgb.y = fmaf(albedo.y, k, rgb.y);
                rgb.z = fmaf(albedo.z, k, rgb.z);
                if (PHONG && p.phong_ks > 0.0f) {
                    const float nl2 = 2.0f * dot3(N, Ld);
                    const F3 R = f3(fmaf(nl2, N.x, -Ld.x), fmaf(nl2, N.y, -Ld.y), fmaf(nl2, N.z, -Ld.z));
                    const float rv = fmaxf(0.0f, dot3(R, view));
                    const float sp = (p.phong_ks * (L.intensity / (kFourPi * r2))) * powUint(rv, p.phong_exp);
                    rgb.x += sp; rgb.y += sp; rgb.z += sp;
                }
            }
        }
    }
    return rgb;
}

// mode 100: Lambert (+ optional Phong highlight) + one shadow ray per light, every material treated as diffuse (oracle: shade_lambert)
template <bool COUNT, int BLOCK, bool PHONG>
__device__ __forceinline__ F3 shadeLambert(const RenderParams& p, const float4* nodes, const float4* tris, const Ray& r,
                                           const Hit& h, Stack& stack, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow)
{
    const Surface sf = surfaceAt(p, tris, r, h);
    return directLight<COUNT, BLOCK, PHONG>(p, nodes, tris, biasPoint(sf.P, sf.N, kShadowBias), sf.N, sf.albedo, f3(-r.d.x, -r.d.y, -r.d.z), stack, iters, cntNodes, cntTris, cntShadow);
}

// ---- mode 200: path tracing (oracle: trace_path). Counter-based RNG keyed by (pixel, sample, seed).
__device__ __forceinline__ uint32_t pcgHash(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float rngNext(uint32_t& st)
{
    st = pcgHash(st);
    return static_cast<float>(st >> 8) * 0x1p-24f;
}

__device__ __forceinline__ uint32_t waveSum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// PHONG: the specular term of mode 100 is compiled into its own variant (chosen at launch when "phong_ks" is non-zero), so
// that the plain Lambert kernel keeps its register budget
template <bool COUNT, bool PHONG>
__global__ __launch_bounds__(64) CRT_OCCUPANCY_ATTR void renderKernel(const RenderParams p)
{
    extern __shared__ int s_stack[]; // stack_entries x 256 dwords, sized at launch from the BVH depth
    unsigned long long t_start = 0;
    if ((CRT_DIAG && p.timeline) || p.unit_cost) t_start = __builtin_amdgcn_s_memrealtime();

    // Work unit = one 8x8-pixel packet = one single-wavefront workgroup; unit u = 4 * j + sub, j = position of the 16x16
    // macro tile in this rank's tile list, sub = which 8x8 quarter.  Which unit a workgroup takes:
    //  * with a cost-sorted order from the previous frame (p.unit_order): the blockIdx-th most expensive unit, so the
    //    packets on the longest critical paths start first and the kernel's tail disappears (longest-processing-time
    //    first; the dispatcher hands out workgroups in blockIdx order);
    //  * otherwise: workgroup b runs on XCD b % 8 (observed round-robin dispatch); groups of xcd_group consecutive tiles
    //    of the list are dealt round-robin to the XCDs so all XCDs sweep the frame together.
    // Either way this is scheduling only: results never depend on it.
    // several frames in one launch (throughput mode): a launch's critical path -- its slowest packet -- is then shared by the
    // whole batch; workgroup index = position * n_batch + frame
    // (position-major: the b-th work units of all frames are neighbours in the launch, so with a cost-sorted order the
    // expensive packets of every frame of the batch start first)
    const uint32_t frame = p.n_batch > 1u ? blockIdx.x % p.n_batch : 0u;
    const uint32_t b = p.n_batch > 1u ? blockIdx.x / p.n_batch : blockIdx.x;
    const float* camPos = frame ? p.batch_pos[frame - 1u] : p.pos;
    const float* camRot = frame ? p.batch_rot[frame - 1u] : p.rot;
    uint32_t* outRgba8 = frame ? p.batch_rgba8[frame - 1u] : p.rgba8;
    uint32_t unit;
    // Split packets: a frame lasts as long as its slowest packets (grazing rays: hundreds of dependent steps, and a
    // wavefront serves its divergent lanes one group after the other), and the launch order already knows which they are.
    // The split_units most expensive packets are therefore rendered by four wavefronts each, one per 4x4 quarter with 16
    // live lanes: fewer rays share a wavefront's step sequence, so each quarter finishes sooner than the whole would, at the
    // price of idle lanes in a few hundred of the 32 640 wavefronts.  Scheduling only: every ray is traced as before.
    uint32_t quarter = 4u; // 4 = the whole 8x8 packet
    if (p.unit_order) {
        uint32_t pos = b;
        if (b < 4u * p.split_units) {
            pos = b >> 2;
            quarter = b & 3u;
        } else {
            pos = b - 3u * p.split_units;
        }
        unit = p.unit_order[pos];
        // the few packets with the longest critical paths bound the frame time even when they start first: let them
        // issue ahead of the other resident wavefronts
        if (pos < p.boost_units) __builtin_amdgcn_s_setprio(3);
#if CRT_DIAG
        if (pos < p.debug_skip_units) return; // drop the most expensive packets to see what bounds the frame
#endif
    } else {
        const uint32_t xcd = b & 7u, seq = b >> 3, i = seq >> 2;
        const uint32_t kGroup = p.xcd_group;
        unit = ((((i / kGroup) * 8u + xcd) * kGroup + (i % kGroup)) << 2) | (seq & 3u);
    }
    const uint32_t j = unit >> 2;
    uint32_t tile_x, tile_y;
    bool valid;
    if (p.n_ranks == 1) {
        // single GPU: the list walks 4x4-tile (64x64-pixel) blocks row-major, tiles row-major inside a block
        const uint32_t blocks_x = (p.tiles_x + 3u) >> 2;
        const uint32_t blk = j >> 4, within = j & 15u;
        tile_x = (blk % blocks_x) * 4u + (within & 3u);
        tile_y = (blk / blocks_x) * 4u + (within >> 2);
        valid = (tile_x < p.tiles_x) & (tile_y < p.tiles_y);
    } else {
        // N GPUs: macro tile k (row-major) belongs to rank k % N; this rank's list is k = j*N + rank
        const uint32_t k = j * p.n_ranks + p.rank;
        valid = k < p.tiles_x * p.tiles_y;
        tile_x = k % p.tiles_x;
        tile_y = k / p.tiles_x;
    }
    if (!valid) {
        if (p.unit_cost && frame == 0u && threadIdx.x == 0 && quarter >= 4u) p.unit_cost[unit] = 0;
        return;
    }

    const uint32_t tid = threadIdx.x, wave = unit & 3u, lane = tid & 63u;
    uint32_t lx = (wave & 1u) * 8u, ly = (wave >> 1) * 8u;
    if (quarter < 4u) { // lanes 0..15 walk the 4x4 quarter, the others have nothing to do
        lx += (quarter & 1u) * 4u + (lane & 3u);
        ly += (quarter >> 1) * 4u + ((lane >> 2) & 3u);
    } else {
        lx += lane & 7u;
        ly += lane >> 3;
    }
    const uint32_t px = tile_x * kTile + lx, py = tile_y * kTile + ly;
    const bool active = (px < p.width) & (py < p.height) & ((quarter == 4u) | (lane < 16u));

    uint32_t cntNodes = 0, cntTris = 0, cntShadow = 0, cntClosest = 0;
#if CRT_PROF
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    unsigned long long pTNode = 0, pTLeaf = 0; uint32_t pItN = 0, pItL = 0, pLaN = 0, pLaL = 0;
    uint32_t pDv[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    uint32_t iters = 0; // traversal-loop iterations of this wavefront = its critical path, fed back as next frame's cost
    if (active) {
        const float4* nodes = reinterpret_cast<const float4*>(p.nodes);
        const float4* tris = reinterpret_cast<const float4*>(p.tris);
        Stack stack;
        stack.lds = s_stack + tid;
        stack.spill = p.spill + ((static_cast<size_t>(frame) * p.units_per_frame + unit) * 64u + (quarter < 4u ? quarter * 16u + tid : tid)) * p.spill_stride;
        stack.cap = static_cast<int>(p.stack_entries);
        stack.sp = 0;

        constexpr int BLOCK = 64;
        F3 col;
        uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
        Hit h;
        {
            const F3 o = f3(camPos[0], camPos[1], camPos[2]);
            const Ray r = makeRay(o, rayDir(camRot, px, py, static_cast<float>(p.width), static_cast<float>(p.height)));
            if (COUNT) cntClosest++;
            traceClosest<COUNT, BLOCK>(nodes, tris, p.n_nodes, r, kTMin, kTMax, stack, static_cast<int>(p.tune_inner_min), h, iters, cntNodes, cntTris);
            col = f3(p.miss[0], p.miss[1], p.miss[2]); // miss shader (hlsl:72-76)
            if (h.t < kTMax) {
                const float4* T = tris + 3 * static_cast<size_t>(h.tri);
                if (p.mode >= 100u) col = shadeLambert<COUNT, BLOCK, PHONG>(p, nodes, tris, r, h, stack, iters, cntNodes, cntTris, cntShadow);
                else col = shadeDebug(p.mode, __float_as_uint(T[0].w), __float_as_uint(T[1].w), h.t, h.u, h.v, r.o, r.d);
            }
        }
        const bool hit = h.t < kTMax;
        if (hit) {
            const float4* T = tris + 3 * static_cast<size_t>(h.tri);
            inst = __float_as_uint(T[0].w);
            prim = __float_as_uint(T[1].w);
        }

#if CRT_PROF
        pTNode = stack.tNode; pTLeaf = stack.tLeaf; pItN = stack.itNode; pItL = stack.itLeaf; pLaN = stack.lanesNode; pLaL = stack.lanesLeaf;
        pDv[0] = stack.dvN; pDv[1] = stack.dvNLanes; pDv[2] = stack.dvNRuns; pDv[3] = stack.dvNDistinct;
        pDv[4] = stack.dvL; pDv[5] = stack.dvLLanes; pDv[6] = stack.dvLRuns; pDv[7] = stack.dvLDistinct;
#endif
        const uint32_t packed = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
        const size_t pix = static_cast<size_t>(py) * p.width + px;
        if (p.staging) outRgba8[static_cast<size_t>((tile_y * p.tiles_x + tile_x) / p.n_ranks) * (kTile * kTile) + ly * kTile + lx] = packed;
        else outRgba8[pix] = packed;
        if (p.hit_inst && frame == 0u) p.hit_inst[pix] = inst;
        if (p.hit_prim && frame == 0u) p.hit_prim[pix] = prim;
        if (p.hit_t && frame == 0u) p.hit_t[pix] = hit ? h.t : kTMax;
        if (p.rgb_f32 && frame == 0u) {
            p.rgb_f32[3 * pix + 0] = col.x;
            p.rgb_f32[3 * pix + 1] = col.y;
            p.rgb_f32[3 * pix + 2] = col.z;
        }
    }
    // cost fed back to order the next frame = this wavefront's lifetime in 0.64 us units (constant 100 MHz clock): it
    // sees what an iteration count does not (distant, incoherent packets are slow per iteration: cache misses)
    // (a split packet reports the sum of its quarters' lifetimes into its slot, which the host zeroes before a measuring launch)
    if (p.unit_cost && frame == 0u && threadIdx.x == 0) {
        const uint32_t life = static_cast<uint32_t>((__builtin_amdgcn_s_memrealtime() - t_start) >> 6);
        if (quarter < 4u) atomicAdd(&p.unit_cost[unit], life);
        else p.unit_cost[unit] = life;
    }
#if CRT_DIAG
    if (p.timeline && threadIdx.x == 0) {
        // wave lifetime on the constant 100 MHz clock, and which XCD ran it
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu; // HW_REG_XCC_ID[3:0]
        p.timeline[3 * static_cast<size_t>(blockIdx.x) + 0] = t_start;
        p.timeline[3 * static_cast<size_t>(blockIdx.x) + 1] = t_end;
        p.timeline[3 * static_cast<size_t>(blockIdx.x) + 2] = (static_cast<unsigned long long>(xcc) << 32) | (tile_y << 16) | tile_x;
    }
#endif
#if CRT_PROF
    if (p.counters) {
        const unsigned long long tk = __builtin_amdgcn_s_memtime() - tk0;
        // wave-uniform quantities: take them from the first active lane that has them (lane values are identical)
        const unsigned long long m = __ballot(pItN | pItL);
        if (m && lane == static_cast<uint32_t>(__ffsll(static_cast<long long>(m)) - 1)) {
            atomicAdd(&p.counters[4], tk); atomicAdd(&p.counters[5], pTNode); atomicAdd(&p.counters[6], pTLeaf);
            atomicAdd(&p.counters[7], static_cast<unsigned long long>(pItN)); atomicAdd(&p.counters[8], static_cast<unsigned long long>(pItL));
            atomicAdd(&p.counters[9], static_cast<unsigned long long>(pLaN)); atomicAdd(&p.counters[10], static_cast<unsigned long long>(pLaL));
            for (int i = 0; i < 8; i++) atomicAdd(&p.counters[11 + i], static_cast<unsigned long long>(pDv[i]));
        }
    }
#endif
    if (COUNT) {
        const uint32_t a = waveSum(cntNodes), c = waveSum(cntTris), s = waveSum(cntShadow), q = waveSum(cntClosest);
        if (lane == 0) {
            atomicAdd(&p.counters[0], static_cast<unsigned long long>(a));
            atomicAdd(&p.counters[1], static_cast<unsigned long long>(c));
            atomicAdd(&p.counters[2], static_cast<unsigned long long>(s));
            atomicAdd(&p.counters[3], static_cast<unsigned long long>(q));
        }
    }
}

// ---- mode 200: path tracing as a wavefront-private pipeline (oracle: trace_path; replaces the per-lane bounce loop, which
// kept 22 % of the lanes busy: a lane whose path had ended idled until the longest path of its wavefront ended).
//
// One workgroup = one wavefront = one pixel tile x `path_samples` samples = B paths: by default an 8x8 packet x up to 16
// samples (256 paths at 4 spp); option "path_tile" = 16 makes it a 16x16 macro tile x 4 samples (longer queues, fewer
// workgroups: slower at both 1080p and 4K).  The wavefront
// runs the whole pipeline for ITS paths by stages, 64 paths at a time, with two private queues in HBM scratch:
//   stage A  camera rays of the tile (coherent 8x8 packets, octant-specialised scalar-fetch descent), closest hit;
//            a miss finishes the path, a hit is appended to the shade queue;
//   stage B  every entry of the shade queue: surface, material, direct light (any-hit shadow rays), next direction; a path
//            that ends writes its radiance, one that goes on is appended to the trace queue;
//   stage C  every entry of the trace queue: closest hit of the bounce ray; miss -> finished, hit -> shade queue; back to B.
// Appending = wavefront ballot + prefix count (mbcnt) + a scalar running count: every stage works on dense 64-path
// chunks, no atomics, no cross-wavefront traffic, no kernel boundary, and the queues are streamed with coalesced dwordx4
// accesses (record i of a queue = one float4 per plane at index i).  Per path the arithmetic -- RNG stream, radiance
// updates, their order -- is the oracle's, so frames stay bit-exact; the sample average runs in sample order at the end.
// Samples beyond `path_samples` are further passes of the same wavefront over the same scratch.
constexpr uint32_t kShadePlanes = 3, kTracePlanes = 2;

struct PathScratch {
    float4* shade;   // kShadePlanes x B: {o, rng} {d, id | bounce << 16} {t, u, v, tri}
    float4* trace;   // kTracePlanes x B: {o, rng} {d, id | bounce << 16}
    float4* done;    // B, by path id (sample-in-pass * tile pixels + pixel-in-tile): the path's radiance so far, final when it ends
    float4* thr;     // B, by path id: its throughput (only the shade stage changes it; the queues carry the ray, not this)
    float4* accum;   // 256: running sum over the samples of earlier passes
    uint32_t B;
};

__device__ __forceinline__ uint32_t lanePrefix(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// Stage C as a stream: the bounce rays of the trace queue are incoherent and their traversals differ a lot in length, so
// a chunk-at-a-time loop leaves most lanes idle while the longest ray of each chunk finishes.  Here a lane that has finished
// retires its ray (miss -> radiance written, hit -> appended to the shade queue) and takes the next entry of the queue,
// as soon as at least `refillMin` lanes are idle: the wavefront stays full until the queue runs dry.  Every ray is still
// traced by one lane in its own fixed order, so results and fetch counts are those of the chunked loop.
#ifndef CRT_REFILL_MIN
#define CRT_REFILL_MIN 16
#endif
template <bool COUNT>
__device__ __forceinline__ void streamClosest(const float4* __restrict__ nodes, const float4* __restrict__ tris, uint32_t n_nodes,
                                              const PathScratch& q, uint32_t nTrace, uint32_t& nShade, F3 miss, Stack& stack, int innerMin,
                                              uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntClosest)
{
    constexpr int BLOCK = 64;
    Ray r = makeRay(f3(0.0f, 0.0f, 0.0f), f3(0.0f, 0.0f, 1.0f));
    Hit h;
    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.gid = 0;
    float tcull = kTMax * kCullPad;
    int cur = kDone;
    bool have = false;   // this lane holds a ray (being traced, or finished and not yet retired)
    uint32_t my = 0;     // its index in the trace queue
    uint32_t next = 0;   // wave-uniform: first queue entry not yet handed to a lane
    const unsigned long long all = __ballot(true);
    for (;;) {
        const bool idle = cur == kDone;
        const unsigned long long idleMask = __ballot(idle);
        if (idleMask == all || (next < nTrace && static_cast<uint32_t>(__popcll(idleMask)) >= static_cast<uint32_t>(CRT_REFILL_MIN))) {
            // retire the finished rays ...
            const bool retire = idle & have, isHit = retire & (h.t < kTMax);
            const unsigned long long mh = __ballot(isHit);
            if (retire) {
                const float4 a0 = q.trace[my], a1 = q.trace[q.B + my];
                if (isHit) {
                    const uint32_t k = nShade + lanePrefix(mh);
                    q.shade[k] = a0;
                    q.shade[q.B + k] = a1;
                    q.shade[2u * q.B + k] = make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
                } else {
                    const uint32_t id = __float_as_uint(a1.w) & 0xFFFFu;
                    const float4 a2 = q.thr[id], a3 = q.done[id];
                    q.done[id] = make_float4(fmaf(a2.x, miss.x, a3.x), fmaf(a2.y, miss.y, a3.y), fmaf(a2.z, miss.z, a3.z), 0.0f);
                }
            }
            nShade += static_cast<uint32_t>(__popcll(mh));
            // ... and hand the next queue entries to the idle lanes
            const uint32_t idx = next + lanePrefix(idleMask);
            if (idle) {
                have = idx < nTrace;
                if (have) {
                    my = idx;
                    const float4 a0 = q.trace[idx], a1 = q.trace[q.B + idx];
                    r = makeRay(f3(a0.x, a0.y, a0.z), f3(a1.x, a1.y, a1.z));
                    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.gid = 0;
                    tcull = kTMax * kCullPad;
                    stack.sp = 0;
                    cur = n_nodes ? 0 : kDone;
                    if (COUNT) cntClosest++;
                }
            }
            next += static_cast<uint32_t>(__popcll(idleMask));
            if (__ballot(have) == 0ull) break; // queue empty and every ray retired
        }
        closestIteration<COUNT, BLOCK, 8>(nodes, tris, r, 0.0f, tcull, stack, innerMin, h, cur, iters, cntNodes, cntTris);
    }
}

// Stage B as a stream (same idea as streamClosest): the entries of the shade queue are shaded by whichever lane is free.
// A lane's life with one entry: fetch (surface, material; mirror / glass / constant finish at once) -> for each light with a
// positive cosine, in light order: one any-hit shadow ray, its contribution added when unoccluded (oracle: direct_light) ->
// retire (radiance update, next direction drawn, appended to the trace queue or written out as finished).  Shadow rays end at
// their first hit, so their traversals differ even more in length than the bounce rays': refilling keeps the wavefront full.
template <bool COUNT>
__device__ __forceinline__ void streamShade(const RenderParams& p, const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                            const PathScratch& q, uint32_t nShade, uint32_t& nTrace, Stack& stack, int innerMin,
                                            uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow)
{
    constexpr int BLOCK = 64;
    const LightRec* lights = reinterpret_cast<const LightRec*>(p.lights);
    Ray sr = makeRay(f3(0.0f, 0.0f, 0.0f), f3(0.0f, 0.0f, 1.0f)); // the shadow ray in flight
    float dist = 0.0f, tcull = 0.0f, kcur = 0.0f;                  // its length, cull bound, and the light's weight if it arrives
    bool occluded = false;
    int cur = kDone;
    bool have = false, diffuse = false, tracing = false, alive = false;
    float thrMul = 0.0f;  // 1: throughput *= albedo when the path goes on; -1: CONSTANT (radiance += throughput * albedo)
    uint32_t my = 0, li = 0;
    F3 Po = f3(0.f, 0.f, 0.f), N = f3(0.f, 0.f, 1.f), albedo = f3(0.f, 0.f, 0.f);
    F3 aux = f3(0.f, 0.f, 0.f); // DIFFUSE: direct light gathered so far; REFLECTIVE / REFRACTIVE: the next direction
    uint32_t next = 0;
    const unsigned long long all = __ballot(true);
    for (;;) {
        const bool idle = cur == kDone;
        const unsigned long long idleMask = __ballot(idle);
        if (idleMask == all || (next < nShade && static_cast<uint32_t>(__popcll(idleMask)) >= static_cast<uint32_t>(CRT_REFILL_MIN))) {
            // 1. a shadow ray has come back: its light counts unless something is in the way
            if (idle & tracing) {
                if (!occluded) aux = f3(fmaf(albedo.x, kcur, aux.x), fmaf(albedo.y, kcur, aux.y), fmaf(albedo.z, kcur, aux.z));
                tracing = false;
                li++;
            }
            // 2. retire: the entries whose lights are all done (or that never had any to ask)
            const bool retire = idle & have & !(diffuse & (li < p.n_lights));
            F3 thr = f3(0.f, 0.f, 0.f), L = f3(0.f, 0.f, 0.f), nd = f3(0.f, 0.f, 0.f);
            uint32_t rng = 0, idb = 0;
            bool goesOn = false;
            if (retire) {
                rng = __float_as_uint(q.shade[my].w);
                idb = __float_as_uint(q.shade[q.B + my].w);
                const float4 a2 = q.thr[idb & 0xFFFFu], a3 = q.done[idb & 0xFFFFu];
                thr = f3(a2.x, a2.y, a2.z);
                L = f3(a3.x, a3.y, a3.z);
                goesOn = alive;
                nd = aux;
                if (thrMul < 0.0f) L = f3(fmaf(thr.x, albedo.x, L.x), fmaf(thr.y, albedo.y, L.y), fmaf(thr.z, albedo.z, L.z));
                if (diffuse) {
                    L = f3(fmaf(thr.x, aux.x, L.x), fmaf(thr.y, aux.y, L.y), fmaf(thr.z, aux.z, L.z));
                    if ((idb >> 16) != p.max_bounces) {
                        const float u1 = rngNext(rng), u2 = rngNext(rng);
                        const float rr = sqrtf(u1), phi = 6.28318530717958648f * u2;
                        const float lx = rr * sinContract(phi + 1.57079632679489662f), ly = rr * sinContract(phi), lz = sqrtf(fmaxf(0.0f, 1.0f - u1));
                        const float sg = copysignf(1.0f, N.z);
                        const float a = -1.0f / (sg + N.z);
                        const float b = N.x * N.y * a;
                        const F3 T = f3(1.0f + sg * N.x * N.x * a, sg * b, -sg * N.x);
                        const F3 Bv = f3(b, sg + N.y * N.y * a, -N.y);
                        const F3 d = f3(fmaf(lz, N.x, fmaf(ly, Bv.x, lx * T.x)), fmaf(lz, N.y, fmaf(ly, Bv.y, lx * T.y)),
                                        fmaf(lz, N.z, fmaf(ly, Bv.z, lx * T.z)));
                        nd = normalize3(d);
                        thrMul = 1.0f;
                        goesOn = true;
                    }
                }
                if (thrMul > 0.0f) {
                    thr = f3(thr.x * albedo.x, thr.y * albedo.y, thr.z * albedo.z);
                    if (goesOn) q.thr[idb & 0xFFFFu] = make_float4(thr.x, thr.y, thr.z, 0.0f);
                }
                q.done[idb & 0xFFFFu] = make_float4(L.x, L.y, L.z, 0.0f); // final if the path ends here, else the sum so far
                have = false;
            }
            const unsigned long long mOn = __ballot(goesOn);
            if (goesOn) {
                const uint32_t k = nTrace + lanePrefix(mOn);
                q.trace[k] = make_float4(Po.x, Po.y, Po.z, __uint_as_float(rng));
                q.trace[q.B + k] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(idb + 0x10000u)); // next bounce
            }
            nTrace += static_cast<uint32_t>(__popcll(mOn));
            // 3. fetch: the lanes without an entry take the next ones of the queue
            const bool wantNew = idle & !have;
            const unsigned long long mNew = __ballot(wantNew);
            const uint32_t idx = next + lanePrefix(mNew);
            if (wantNew && idx < nShade) {
                have = true;
                my = idx;
                const float4 a0 = q.shade[idx], a1 = q.shade[q.B + idx];
                const Ray r = makeRay(f3(a0.x, a0.y, a0.z), f3(a1.x, a1.y, a1.z));
                const uint32_t bounce = __float_as_uint(a1.w) >> 16;
                Hit h;
                const float4 a4 = q.shade[2u * q.B + idx];
                h.t = a4.x; h.u = a4.y; h.v = a4.z; h.tri = __float_as_uint(a4.w); h.gid = 0;
                const Surface sf = surfaceAt(p, tris, r, h);
                N = sf.N;
                albedo = sf.albedo;
                diffuse = false; alive = false; thrMul = 0.0f; li = 0;
                aux = f3(0.0f, 0.0f, 0.0f);
                Po = biasPoint(sf.P, sf.N, kShadowBias);
                if (sf.mtype == 4u) { // CONSTANT
                    thrMul = -1.0f;
                } else if (sf.mtype == 2u) { // REFLECTIVE
                    if (bounce != p.max_bounces) {
                        const float k = 2.0f * dot3(r.d, sf.N);
                        aux = normalize3(f3(fmaf(-k, sf.N.x, r.d.x), fmaf(-k, sf.N.y, r.d.y), fmaf(-k, sf.N.z, r.d.z)));
                        thrMul = 1.0f;
                        alive = true;
                    }
                } else if (sf.mtype == 3u) { // REFRACTIVE
                    if (bounce != p.max_bounces) {
                        const float eta = sf.entering ? 1.0f / sf.ior : sf.ior;
                        const float cosi = -dot3(r.d, sf.N);
                        const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
                        F3 d;
                        if (k < 0.0f) {
                            const float m2 = 2.0f * dot3(r.d, sf.N);
                            d = f3(fmaf(-m2, sf.N.x, r.d.x), fmaf(-m2, sf.N.y, r.d.y), fmaf(-m2, sf.N.z, r.d.z));
                        } else {
                            const float m2 = eta * cosi - sqrtf(k);
                            d = f3(fmaf(m2, sf.N.x, eta * r.d.x), fmaf(m2, sf.N.y, eta * r.d.y), fmaf(m2, sf.N.z, eta * r.d.z));
                            Po = biasPoint(sf.P, sf.N, -kShadowBias);
                        }
                        aux = normalize3(d);
                        alive = true;
                    }
                } else {
                    diffuse = true;
                }
            }
            next += static_cast<uint32_t>(__popcll(mNew));
            // 4. the next light of every diffuse entry that is not waiting for a shadow ray
            if ((cur == kDone) & have & diffuse & !tracing) {
                while (li < p.n_lights) {
                    const LightRec Lt = lights[li];
                    const F3 Lv = sub3(f3(Lt.x, Lt.y, Lt.z), Po);
                    const float r2 = dot3(Lv, Lv);
                    const float d1 = sqrtf(r2);
                    const float invr = 1.0f / d1;
                    const F3 Ldir = f3(Lv.x * invr, Lv.y * invr, Lv.z * invr);
                    const float cosv = fmaxf(0.0f, dot3(N, Ldir));
                    if (cosv > 0.0f) {
                        sr = makeRay(Po, Ldir);
                        dist = d1;
                        tcull = d1 * kCullPad;
                        kcur = (Lt.intensity / (kFourPi * r2)) * cosv;
                        occluded = false;
                        tracing = true;
                        stack.sp = 0;
                        cur = p.n_nodes ? 0 : kDone;
                        if (COUNT) cntShadow++;
                        break;
                    }
                    li++;
                }
            }
            if (__ballot(have) == 0ull) break;
        }
        anyIteration<COUNT, BLOCK, 8>(nodes, tris, sr, 0.0f, dist, tcull, stack, innerMin, occluded, cur, iters, cntNodes, cntTris);
    }
}

#ifndef CRT_PATH_WAVES_PER_EU
#define CRT_PATH_WAVES_PER_EU 5
#endif
template <bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_PATH_WAVES_PER_EU, 8))) void pathKernel(const RenderParams p)
{
    extern __shared__ int s_stack[];
    const uint32_t frame = p.n_batch > 1u ? blockIdx.x % p.n_batch : 0u;
    const uint32_t wg = p.n_batch > 1u ? blockIdx.x / p.n_batch : blockIdx.x;
    const bool big = p.path_tile == 16u;                  // workgroup = whole macro tile (four 8x8 packets per sample) or one 8x8 packet
    const uint32_t j = big ? wg : wg >> 2;                // position of the macro tile in this rank's list
    const uint32_t subFirst = big ? 0u : (wg & 3u), subCount = big ? 4u : 1u;
    const uint32_t tilePixels = subCount * 64u;
    const float* camPos = frame ? p.batch_pos[frame - 1u] : p.pos;
    const float* camRot = frame ? p.batch_rot[frame - 1u] : p.rot;
    uint32_t* outRgba8 = frame ? p.batch_rgba8[frame - 1u] : p.rgba8;
    uint32_t tile_x, tile_y;
    bool valid;
    if (p.n_ranks == 1) {
        const uint32_t blocks_x = (p.tiles_x + 3u) >> 2;
        const uint32_t blk = j >> 4, within = j & 15u;
        tile_x = (blk % blocks_x) * 4u + (within & 3u);
        tile_y = (blk / blocks_x) * 4u + (within >> 2);
        valid = (tile_x < p.tiles_x) & (tile_y < p.tiles_y);
    } else {
        const uint32_t k = j * p.n_ranks + p.rank;
        valid = k < p.tiles_x * p.tiles_y;
        tile_x = k % p.tiles_x;
        tile_y = k / p.tiles_x;
    }
    if (!valid) return;

    const uint32_t lane = threadIdx.x & 63u;
    const float4* nodes = reinterpret_cast<const float4*>(p.nodes);
    const float4* tris = reinterpret_cast<const float4*>(p.tris);
    Stack stack;
    stack.lds = s_stack + lane;
    stack.spill = p.spill + (static_cast<size_t>(blockIdx.x) * 64u + lane) * p.spill_stride;
    stack.cap = static_cast<int>(p.stack_entries);
    stack.sp = 0;
    constexpr int BLOCK = 64;
    const int innerMin = static_cast<int>(p.tune_inner_min);

    PathScratch q;
    q.B = tilePixels * p.path_samples;
    {
        float4* base = reinterpret_cast<float4*>(p.path_scratch + static_cast<size_t>(blockIdx.x) * p.path_region_bytes);
        q.shade = base;
        q.trace = q.shade + static_cast<size_t>(kShadePlanes) * q.B;
        q.done = q.trace + static_cast<size_t>(kTracePlanes) * q.B;
        q.thr = q.done + q.B;
        q.accum = q.thr + q.B;
    }
    const F3 miss = f3(p.miss[0], p.miss[1], p.miss[2]);
    uint32_t cntNodes = 0, cntTris = 0, cntShadow = 0, cntClosest = 0, iters = 0;

    for (uint32_t s0 = 0; s0 < p.spp; s0 += p.path_samples) {
        const uint32_t nS = min(p.path_samples, p.spp - s0);
        uint32_t nShade = 0; // wave-uniform queue lengths
        // ---- stage A: the tile's camera rays, one 8x8 packet of one sample at a time
        for (uint32_t sl = 0; sl < nS; sl++) {
            for (uint32_t sb = 0; sb < subCount; sb++) {
                const uint32_t sub = subFirst + sb;
                const uint32_t lx = (sub & 1u) * 8u + (lane & 7u), ly = (sub >> 1) * 8u + (lane >> 3);
                const uint32_t px = tile_x * kTile + lx, py = tile_y * kTile + ly;
                const bool active = (px < p.width) & (py < p.height);
                const uint32_t id = sl * tilePixels + sb * 64u + lane; // path id inside the workgroup
                bool isHit = false;
                Ray r;
                Hit h;
                uint32_t rng = 0;
                if (active) {
                    const uint32_t pixId = py * p.width + px;
                    rng = pcgHash(pixId ^ pcgHash((s0 + sl) + pcgHash(p.seed)));
                    const float jx = rngNext(rng), jy = rngNext(rng);
                    r = makeRay(f3(camPos[0], camPos[1], camPos[2]), rayDirJ(camRot, px, py, jx, jy, static_cast<float>(p.width), static_cast<float>(p.height)));
                    if (COUNT) cntClosest++;
                    traceClosest<COUNT, BLOCK>(nodes, tris, p.n_nodes, r, kTMin, kTMax, stack, innerMin, h, iters, cntNodes, cntTris);
                    isHit = h.t < kTMax;
                    // radiance so far: a miss ends the path with throughput (1) x miss colour; a hit starts from nothing, throughput 1
                    q.done[id] = isHit ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fmaf(1.0f, miss.x, 0.0f), fmaf(1.0f, miss.y, 0.0f), fmaf(1.0f, miss.z, 0.0f), 0.0f);
                    if (isHit) q.thr[id] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                    if (s0 + sl == 0u && frame == 0u) { // the hit outputs report sample 0's camera ray
                        const size_t pix = static_cast<size_t>(py) * p.width + px;
                        uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
                        if (isHit) {
                            const float4* T = tris + 3 * static_cast<size_t>(h.tri);
                            inst = __float_as_uint(T[0].w);
                            prim = __float_as_uint(T[1].w);
                        }
                        if (p.hit_inst) p.hit_inst[pix] = inst;
                        if (p.hit_prim) p.hit_prim[pix] = prim;
                        if (p.hit_t) p.hit_t[pix] = isHit ? h.t : kTMax;
                    }
                }
                const unsigned long long m = __ballot(isHit);
                if (isHit) {
                    const uint32_t i = nShade + lanePrefix(m);
                    q.shade[i] = make_float4(r.o.x, r.o.y, r.o.z, __uint_as_float(rng));
                    q.shade[q.B + i] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(id)); // bounce 0 in the upper half
                    q.shade[2u * q.B + i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
                }
                nShade += static_cast<uint32_t>(__popcll(m));
            }
        }
        // ---- stages B / C until no path is left
        while (nShade != 0u) {
            uint32_t nTrace = 0;
            streamShade<COUNT>(p, nodes, tris, q, nShade, nTrace, stack, innerMin, iters, cntNodes, cntTris, cntShadow); // stage B
            nShade = 0;
            streamClosest<COUNT>(nodes, tris, p.n_nodes, q, nTrace, nShade, miss, stack, innerMin, iters, cntNodes, cntTris, cntClosest); // stage C
        }
        // ---- this pass's samples join the running sums in sample order; after the last pass: average, quantise, store
        const bool last = s0 + nS >= p.spp;
        for (uint32_t sb = 0; sb < subCount; sb++) {
            const uint32_t sub = subFirst + sb;
            const uint32_t lx = (sub & 1u) * 8u + (lane & 7u), ly = (sub >> 1) * 8u + (lane >> 3);
            const uint32_t px = tile_x * kTile + lx, py = tile_y * kTile + ly;
            if ((px < p.width) & (py < p.height)) {
                const uint32_t pl = sb * 64u + lane;
                F3 acc = f3(0.0f, 0.0f, 0.0f);
                if (s0 != 0u) {
                    const float4 a = q.accum[pl];
                    acc = f3(a.x, a.y, a.z);
                }
                for (uint32_t sl = 0; sl < nS; sl++) {
                    const float4 Ls = q.done[sl * tilePixels + pl];
                    acc = f3(acc.x + Ls.x, acc.y + Ls.y, acc.z + Ls.z);
                }
                if (!last) {
                    q.accum[pl] = make_float4(acc.x, acc.y, acc.z, 0.0f);
                } else {
                    const float inv = 1.0f / static_cast<float>(p.spp);
                    const F3 col = f3(acc.x * inv, acc.y * inv, acc.z * inv);
                    const uint32_t packed = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
                    const size_t pix = static_cast<size_t>(py) * p.width + px;
                    if (p.staging) outRgba8[static_cast<size_t>((tile_y * p.tiles_x + tile_x) / p.n_ranks) * (kTile * kTile) + ly * kTile + lx] = packed;
                    else outRgba8[pix] = packed;
                    if (p.rgb_f32 && frame == 0u) {
                        p.rgb_f32[3 * pix + 0] = col.x;
                        p.rgb_f32[3 * pix + 1] = col.y;
                        p.rgb_f32[3 * pix + 2] = col.z;
                    }
                }
            }
        }
    }
    if (COUNT) {
        const uint32_t a = waveSum(cntNodes), c = waveSum(cntTris), sh = waveSum(cntShadow), cl = waveSum(cntClosest);
        if (lane == 0) {
            atomicAdd(&p.counters[0], static_cast<unsigned long long>(a));
            atomicAdd(&p.counters[1], static_cast<unsigned long long>(c));
            atomicAdd(&p.counters[2], static_cast<unsigned long long>(sh));
            atomicAdd(&p.counters[3], static_cast<unsigned long long>(cl));
        }
    }
}

__global__ __launch_bounds__(kBlock) void untileKernel(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame,
                                                       uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_ranks,
                                                       uint32_t rank_stride, uint32_t first_slot)
{
    const uint32_t px = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t py = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (px >= width || py >= height) return;
    const uint32_t k = (py / kTile) * tiles_x + px / kTile;
    const uint32_t rank = k % n_ranks, slot = k / n_ranks;
    // rank r's tiles start at slot r * rank_stride + first_slot (one frame: stride = slots, first = 0; frame f of a gathered
    // batch of B frames: stride = B * slots, first = f * slots)
    const size_t src = (static_cast<size_t>(rank) * rank_stride + first_slot + slot) * (kTile * kTile) + (py % kTile) * kTile + (px % kTile);
    frame[static_cast<size_t>(py) * width + px] = gathered[src];
}

} // namespace

uint32_t renderUnitCount(const RenderParams& p)
{
    uint32_t n = p.n_local_tiles;
    if (p.n_ranks == 1) n = ((p.tiles_x + 3u) / 4u) * ((p.tiles_y + 3u) / 4u) * 16u;
    n = (n + 8u * kGroupMax - 1u) / (8u * kGroupMax) * (8u * kGroupMax);
    return n * 4u;
}

// scratch the path-tracing pipeline needs per workgroup (one macro tile): the two queues, the finished-path radiances and
// the cross-pass sums; and how many workgroups launchRender starts for p in mode 200
size_t pathRegionBytes(uint32_t tile, uint32_t samples_per_pass)
{
    const size_t pixels = static_cast<size_t>(tile) * tile, B = pixels * samples_per_pass;
    return (kShadePlanes + kTracePlanes + 2u) * B * sizeof(float4) + pixels * sizeof(float4); // queues + radiance + throughput, + cross-pass sums
}
uint32_t pathWorkgroupCount(const RenderParams& p) { return renderUnitCount(p) / (p.path_tile == 16u ? 4u : 1u) * (p.n_batch ? p.n_batch : 1u); }

int launchRender(const RenderParams& p, bool counting, ihipStream_t* stream)
{
    if (p.n_local_tiles == 0) return 0;
    // list length: single GPU walks whole 4x4-tile blocks (padded at the frame edges); then padded to 8 XCDs x kGroupMax
    const uint32_t n = renderUnitCount(p) / 4u;
    const dim3 block(64);
    const size_t lds = static_cast<size_t>(p.stack_entries) * 64u * sizeof(int);
    if (p.mode >= 200u) { // one wavefront per macro tile carries all its paths through the pipeline
        const dim3 grid(pathWorkgroupCount(p));
        if (counting) hipLaunchKernelGGL((pathKernel<true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((pathKernel<false>), grid, block, lds, stream, p);
    } else {
        const dim3 grid((n * 4u + (p.unit_order ? 3u * p.split_units : 0u)) * (p.n_batch ? p.n_batch : 1u));
        const bool phong = p.mode >= 100u && p.phong_ks > 0.0f;
        if (counting && phong) hipLaunchKernelGGL((renderKernel<true, true>), grid, block, lds, stream, p);
        else if (counting) hipLaunchKernelGGL((renderKernel<true, false>), grid, block, lds, stream, p);
        else if (phong) hipLaunchKernelGGL((renderKernel<false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((renderKernel<false, false>), grid, block, lds, stream, p);
    }
    return static_cast<int>(hipGetLastError());
}

// Order the work units by descending cost (counting sort on min(cost, 1023); order inside a bucket is arbitrary).
// One 256-thread workgroup: small enough to be placed on a CU beside resident render wavefronts (a 1024-thread group
// waited for 16 free wave slots on one CU, i.e. for the next frame's render kernel to drain).
constexpr uint32_t kSortThreads = 256, kSortBuckets = 1024, kSortPerThread = kSortBuckets / kSortThreads;
__global__ __launch_bounds__(kSortThreads) void sortUnitsKernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n)
{
    __shared__ uint32_t hist[kSortBuckets];
    __shared__ uint32_t sums[kSortThreads];
    const uint32_t t = threadIdx.x;
    for (uint32_t b = t; b < kSortBuckets; b += kSortThreads) hist[b] = 0;
    __syncthreads();
    for (uint32_t i = t; i < n; i += kSortThreads) atomicAdd(&hist[min(cost[i], kSortBuckets - 1u)], 1u);
    __syncthreads();
    // most expensive bucket first: thread t owns the reversed buckets 4t..4t+3
    uint32_t local[kSortPerThread], total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSortPerThread; k++) {
        local[k] = hist[kSortBuckets - 1u - (t * kSortPerThread + k)];
        total += local[k];
    }
    sums[t] = total;
    __syncthreads();
    for (uint32_t d = 1; d < kSortThreads; d <<= 1) { // inclusive scan of the per-thread totals
        const uint32_t add = t >= d ? sums[t - d] : 0u;
        __syncthreads();
        sums[t] += add;
        __syncthreads();
    }
    uint32_t base = sums[t] - total;
#pragma unroll
    for (uint32_t k = 0; k < kSortPerThread; k++) { // hist[b] = first output slot of bucket b
        hist[kSortBuckets - 1u - (t * kSortPerThread + k)] = base;
        base += local[k];
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += kSortThreads) order[atomicAdd(&hist[min(cost[i], kSortBuckets - 1u)], 1u)] = i;
}

int launchSortUnits(const uint32_t* cost, uint32_t* order, uint32_t n, ihipStream_t* stream)
{
    hipLaunchKernelGGL(sortUnitsKernel, dim3(1), dim3(kSortThreads), 0, stream, cost, order, n);
    return static_cast<int>(hipGetLastError());
}

int launchUntile(const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height, uint32_t n_ranks,
                 uint32_t rank_stride, uint32_t first_slot, ihipStream_t* stream)
{
    const uint32_t tiles_x = (width + kTile - 1) / kTile;
    const dim3 grid((width + 63) / 64, (height + 3) / 4), block(kBlock);
    hipLaunchKernelGGL(untileKernel, grid, block, 0, stream, gathered, frame, width, height, tiles_x, n_ranks, rank_stride, first_slot);
    return static_cast<int>(hipGetLastError());
}

} // namespace crt
