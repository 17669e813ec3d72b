// GPU BVH build: stands in for BuildRaytracingAccelerationStructure, which the reference
// runs on the GPU at init (R/DXRTRenderer.cpp:548-806).  LBVH: 30-bit Morton codes of the quantised box centroids,
// radix sort, Karras 2012 hierarchy, bottom-up exact box fitting, ranges of <= 2 triangles collapsed to leaves, then --
// still on the device -- the collapse to the 4-wide tree (level by level, bvh_wide.h wideSlots) and its quantisation.
// Specification = oracle/crt_oracle.c build_lbvh(); the two produce byte-identical trees (tests/test_gpu_parity.py).
// The meshes go up as they are (vertices, indices, normals, uvs: 18 MB for 1M triangles instead of 130 MB of flattened
// records); triangle boxes, the leaf-ordered triangle / shading / uv records, the binary nodes, the wide nodes and their
// quantised form are produced on the device and STAY there (the context adopts the buffers; crt_bvh_export* read them back
// on demand).  Nothing but a few counters comes back to the host.
// Cubic Morton cells and leaves of at most 2 triangles (per-axis scaling and the SAH builder's 4-triangle leaves measured +36 %
// node fetches and +94 % triangle tests against the SAH tree on the 1M-triangle frame; now +18 % / -13 %).  It is the fast
// option (option "gpu_build": 8.7 ms against 360 ms at 1M triangles), not the default.
#include "bvh_build.h"
#include "bvh_wide.h"

#include <hip/hip_runtime.h>
#include "gpu_sort.hip.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstring>
#include <stdexcept>
#include <string>

namespace crt {
namespace {

#define GPU_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr " failed: ") + hipGetErrorString(e_)); \
    } while (0)

struct KNode { // Karras internal node
    int left, right;   // >= 0 internal index, < 0: ~sorted leaf position
    uint32_t lo, hi;   // range of sorted positions covered
};

// float <-> int such that signed int order == float order (for atomicMin/atomicMax on floats)
__device__ __forceinline__ int orderedInt(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float fromOrderedInt(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void initBoundsKernel(int* bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = INT_MAX;      // min
    else if (threadIdx.x < 6) bounds[threadIdx.x] = INT_MIN; // max
}

// centroid bounds: a fixed grid strides over the centroids, wave reduction, one atomic pair per wavefront and axis
// (one wavefront per 64 centroids meant 94 k atomics on the same six words at 1M triangles: 1.1 ms, most of it queueing)
constexpr uint32_t kBoundsBlocks = 256;
__global__ __launch_bounds__(256) void boundsKernel(const float* __restrict__ cent, uint32_t n, int* bounds)
{
    int mn[3] = { INT_MAX, INT_MAX, INT_MAX }, mx[3] = { INT_MIN, INT_MIN, INT_MIN };
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += kBoundsBlocks * 256u)
        for (int a = 0; a < 3; a++) {
            const int v = orderedInt(cent[3 * static_cast<size_t>(i) + a]);
            mn[a] = min(mn[a], v);
            mx[a] = max(mx[a], v);
        }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) {
            mn[a] = min(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = max(mx[a], __shfl_xor(mx[a], off, 64));
        }
    if ((threadIdx.x & 63u) == 0)
        for (int a = 0; a < 3; a++) {
            atomicMin(&bounds[a], mn[a]);
            atomicMax(&bounds[3 + a], mx[a]);
        }
}

__device__ __forceinline__ uint32_t expandBits10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t quant10(float f) { return f >= 0.0f ? (f < 1024.0f ? static_cast<uint32_t>(f) : 1023u) : 0u; }

__global__ __launch_bounds__(256) void mortonKernel(const float* __restrict__ cent, uint32_t n, const int* __restrict__ bounds,
                                                    unsigned long long* __restrict__ keys)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    float extm = 0.0f; // one cell size for all three axes (that of the longest extent): cubic cells
    for (int a = 0; a < 3; a++) {
        const float ext = fromOrderedInt(bounds[3 + a]) - fromOrderedInt(bounds[a]);
        if (ext > extm) extm = ext;
    }
    const float scale = extm > 0.0f ? 1024.0f / extm : 0.0f;
    for (int a = 0; a < 3; a++) q[a] = quant10((cent[3 * static_cast<size_t>(i) + a] - fromOrderedInt(bounds[a])) * scale);
    const uint32_t code = (expandBits10(q[0]) << 2) | (expandBits10(q[1]) << 1) | expandBits10(q[2]);
    keys[i] = (static_cast<unsigned long long>(code) << 32) | i;
}

__device__ __forceinline__ int delta64(const unsigned long long* __restrict__ keys, long long n, long long i, long long j)
{
    if (j < 0 || j >= n) return -1;
    return __clzll(static_cast<long long>(keys[i] ^ keys[j]));
}

// Karras 2012, one thread per internal node
__global__ __launch_bounds__(256) void hierarchyKernel(const unsigned long long* __restrict__ keys, uint32_t n, KNode* __restrict__ K,
                                                       int* __restrict__ parentOfInternal, int* __restrict__ parentOfLeaf)
{
    const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    const long long N = n;
    if (i >= N - 1) return;
    const int d = (delta64(keys, N, i, i + 1) - delta64(keys, N, i, i - 1)) < 0 ? -1 : 1;
    const int dmin = delta64(keys, N, i, i - d);
    long long lmax = 2;
    while (delta64(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    long long l = 0;
    for (long long t = lmax / 2; t >= 1; t /= 2)
        if (delta64(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const long long j = i + l * d;
    const int dnode = delta64(keys, N, i, j);
    long long sp = 0;
    for (long long t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta64(keys, N, i, i + (sp + t) * d) > dnode) sp += t;
        if (t == 1) break;
    }
    const long long gamma = i + sp * d + (d < 0 ? -1 : 0);
    const long long lo = i < j ? i : j, hi = i < j ? j : i;
    KNode k;
    k.lo = static_cast<uint32_t>(lo);
    k.hi = static_cast<uint32_t>(hi);
    if (lo == gamma) { k.left = ~static_cast<int>(gamma); parentOfLeaf[gamma] = static_cast<int>(i); }
    else { k.left = static_cast<int>(gamma); parentOfInternal[gamma] = static_cast<int>(i); }
    if (hi == gamma + 1) { k.right = ~static_cast<int>(gamma + 1); parentOfLeaf[gamma + 1] = static_cast<int>(i); }
    else { k.right = static_cast<int>(gamma + 1); parentOfInternal[gamma + 1] = static_cast<int>(i); }
    K[i] = k;
    if (i == 0) parentOfInternal[0] = -1;
}

struct Box6 { float mn[3], mx[3]; };

// bottom-up exact boxes: the second thread to arrive at a node unions its children and moves on.  A node whose triangle
// range lies inside the 256 leaves of one workgroup is only ever touched by that workgroup: arrival counter and fences at
// workgroup scope, which cost nothing.  A node spanning workgroups hands boxes between XCDs through memory, so those hops
// are fenced at agent scope on both sides (MI355X: per-XCD L2s are not coherent; cdna_hip_programming.md Guideline 16) --
// an agent-scope fence per hop for EVERY node was 6.4 of the build's 8.9 ms of kernel time at 1M triangles.
__global__ __launch_bounds__(256) void fitKernel(const KNode* __restrict__ K, const Box6* __restrict__ pbox, const unsigned long long* __restrict__ keys,
                                                 uint32_t n, const int* __restrict__ parentOfInternal, const int* __restrict__ parentOfLeaf,
                                                 Box6* nodeBox, unsigned int* flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t blockLo = blockIdx.x * 256u, blockHi = blockLo + 255u;
    int p = parentOfLeaf[i];
    while (p >= 0) {
        const KNode k = K[p];
        unsigned int earlier;
        if ((k.lo >= blockLo) & (k.hi <= blockHi)) {
            // release what this thread wrote below, announce arrival, acquire the sibling's boxes: all inside the workgroup
            earlier = __hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            earlier = __hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (earlier == 0u) return; // first arrival: the sibling subtree is not finished yet
        Box6 b;
        for (int c = 0; c < 2; c++) {
            const int ref = c == 0 ? k.left : k.right;
            Box6 cb;
            if (ref < 0) cb = pbox[static_cast<uint32_t>(keys[~ref] & 0xFFFFFFFFull)];
            else {
                const volatile Box6* src = nodeBox + ref; // written by another thread during this launch: do not cache in registers early
                for (int a = 0; a < 3; a++) { cb.mn[a] = src->mn[a]; cb.mx[a] = src->mx[a]; }
            }
            if (c == 0) b = cb;
            else
                for (int a = 0; a < 3; a++) {
                    b.mn[a] = b.mn[a] < cb.mn[a] ? b.mn[a] : cb.mn[a]; // same selects as the oracle's aabb_grow(l, r)
                    b.mx[a] = b.mx[a] > cb.mx[a] ? b.mx[a] : cb.mx[a];
                }
        }
        // write-through stores: an agent-scope release writes this XCD's dirty L2 lines back, and it is far cheaper when
        // the 24 MB of boxes this kernel produces are not sitting there dirty
        for (int a = 0; a < 3; a++) {
            __hip_atomic_store(&nodeBox[p].mn[a], b.mn[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&nodeBox[p].mx[a], b.mx[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        p = parentOfInternal[p];
    }
}

__global__ __launch_bounds__(256) void keptKernel(const KNode* __restrict__ K, uint32_t nInternal, uint32_t* __restrict__ kept)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nInternal) kept[i] = (K[i].hi - K[i].lo + 1u > static_cast<uint32_t>(kLbvhLeafMax)) ? 1u : 0u;
}

__device__ __forceinline__ int leafRefDev(uint32_t first, uint32_t count) { return ~static_cast<int>((first << 3) | count); }

__global__ __launch_bounds__(256) void emitKernel(const KNode* __restrict__ K, const uint32_t* __restrict__ kept, const uint32_t* __restrict__ rank,
                                                  const Box6* __restrict__ nodeBox, const Box6* __restrict__ pbox,
                                                  const unsigned long long* __restrict__ keys, uint32_t nInternal, crt_bvh_node* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nInternal || !kept[i]) return;
    const KNode k = K[i];
    Box6 b[2];
    int ref[2];
    for (int c = 0; c < 2; c++) {
        const int ch = c == 0 ? k.left : k.right;
        if (ch < 0) {
            b[c] = pbox[static_cast<uint32_t>(keys[~ch] & 0xFFFFFFFFull)];
            ref[c] = leafRefDev(static_cast<uint32_t>(~ch), 1u);
        } else {
            b[c] = nodeBox[ch];
            const uint32_t cnt = K[ch].hi - K[ch].lo + 1u;
            ref[c] = cnt <= static_cast<uint32_t>(kLbvhLeafMax) ? leafRefDev(K[ch].lo, cnt) : static_cast<int>(rank[ch]);
        }
    }
    crt_bvh_node N;
    N.lx0 = b[0].mn[0]; N.lx1 = b[0].mx[0]; N.ly0 = b[0].mn[1]; N.ly1 = b[0].mx[1]; N.lz0 = b[0].mn[2]; N.lz1 = b[0].mx[2];
    N.rx0 = b[1].mn[0]; N.rx1 = b[1].mx[0]; N.ry0 = b[1].mn[1]; N.ry1 = b[1].mx[1]; N.rz0 = b[1].mn[2]; N.rz1 = b[1].mx[2];
    N.left = ref[0]; N.right = ref[1]; N.pad0 = 0; N.pad1 = 0;
    out[rank[i]] = N;
}

// One entry per mesh (plus a terminator holding the totals): where its triangles and vertices start in the concatenated arrays
struct MeshEntry {
    uint32_t triStart, vertStart, nVerts, material;
    uint32_t hasNormals, hasUvs, pad0, pad1;
};

__device__ __forceinline__ uint32_t meshOf(const MeshEntry* __restrict__ table, uint32_t n_meshes, uint32_t g)
{
    uint32_t lo = 0, hi = n_meshes; // largest m with triStart[m] <= g (meshes without triangles are skipped by the <=)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (table[mid].triStart <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ float minSel(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float maxSel(float a, float b) { return a > b ? a : b; }

// per input triangle (global ordinal g): bounds and centroid, exactly as flattenMeshes computes them on the host
__global__ __launch_bounds__(256) void triBoxKernel(const MeshEntry* __restrict__ table, uint32_t n_meshes, const float* __restrict__ xyz,
                                                    const uint32_t* __restrict__ idx, uint32_t n, Box6* __restrict__ box, float* __restrict__ cent,
                                                    int* __restrict__ badIndex)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n) return;
    const MeshEntry M = table[meshOf(table, n_meshes, g)];
    const uint32_t i0 = idx[3 * static_cast<size_t>(g)], i1 = idx[3 * static_cast<size_t>(g) + 1], i2 = idx[3 * static_cast<size_t>(g) + 2];
    Box6 b;
    if (i0 >= M.nVerts || i1 >= M.nVerts || i2 >= M.nVerts) {
        *badIndex = 1;
        for (int k = 0; k < 3; k++) { b.mn[k] = 0.0f; b.mx[k] = 0.0f; cent[3 * static_cast<size_t>(g) + k] = 0.0f; }
        box[g] = b;
        return;
    }
    const float* A = xyz + 3 * static_cast<size_t>(M.vertStart + i0);
    const float* B = xyz + 3 * static_cast<size_t>(M.vertStart + i1);
    const float* C = xyz + 3 * static_cast<size_t>(M.vertStart + i2);
    for (int k = 0; k < 3; k++) {
        b.mn[k] = minSel(minSel(A[k], B[k]), C[k]);
        b.mx[k] = maxSel(maxSel(A[k], B[k]), C[k]);
        cent[3 * static_cast<size_t>(g) + k] = (b.mn[k] + b.mx[k]) * 0.5f;
    }
    box[g] = b;
}

// leaf position i <- input triangle keys[i] & 0xFFFFFFFF: the 48-byte triangle record, the shading record and (when any
// mesh has them) the uv record, written straight from the mesh arrays
__global__ __launch_bounds__(256) void gatherKernel(const unsigned long long* __restrict__ keys, uint32_t n, const MeshEntry* __restrict__ table,
                                                    uint32_t n_meshes, const float* __restrict__ xyz, const uint32_t* __restrict__ idx,
                                                    const float* __restrict__ normals, const float* __restrict__ uvsIn, crt_bvh_tri* __restrict__ tris,
                                                    crt_bvh_shade* __restrict__ shade, crt_bvh_uv* __restrict__ uvs)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = static_cast<uint32_t>(keys[i] & 0xFFFFFFFFull);
    const uint32_t m = meshOf(table, n_meshes, g);
    const MeshEntry M = table[m];
    const uint32_t v0 = M.vertStart + idx[3 * static_cast<size_t>(g)], v1 = M.vertStart + idx[3 * static_cast<size_t>(g) + 1],
                   v2 = M.vertStart + idx[3 * static_cast<size_t>(g) + 2];
    const float* A = xyz + 3 * static_cast<size_t>(v0);
    const float* B = xyz + 3 * static_cast<size_t>(v1);
    const float* C = xyz + 3 * static_cast<size_t>(v2);
    crt_bvh_tri T;
    for (int k = 0; k < 3; k++) {
        T.v0[k] = A[k];
        T.e1[k] = B[k] - A[k];
        T.e2[k] = C[k] - A[k];
    }
    T.inst = m;
    T.prim = g - M.triStart;
    T.gid = g;
    tris[i] = T;
    crt_bvh_shade S;
    for (int k = 0; k < 3; k++) { S.n0[k] = 0.0f; S.n1[k] = 0.0f; S.n2[k] = 0.0f; }
    S.material = M.material;
    S.pad[0] = 0; S.pad[1] = 0;
    if (M.hasNormals) {
        for (int k = 0; k < 3; k++) {
            S.n0[k] = normals[3 * static_cast<size_t>(v0) + k];
            S.n1[k] = normals[3 * static_cast<size_t>(v1) + k];
            S.n2[k] = normals[3 * static_cast<size_t>(v2) + k];
        }
    }
    shade[i] = S;
    if (uvs) {
        crt_bvh_uv U;
        U.uv0[0] = U.uv0[1] = U.uv1[0] = U.uv1[1] = U.uv2[0] = U.uv2[1] = 0.0f;
        if (M.hasUvs) {
            U.uv0[0] = uvsIn[3 * static_cast<size_t>(v0)]; U.uv0[1] = uvsIn[3 * static_cast<size_t>(v0) + 1];
            U.uv1[0] = uvsIn[3 * static_cast<size_t>(v1)]; U.uv1[1] = uvsIn[3 * static_cast<size_t>(v1) + 1];
            U.uv2[0] = uvsIn[3 * static_cast<size_t>(v2)]; U.uv2[1] = uvsIn[3 * static_cast<size_t>(v2) + 1];
        }
        uvs[i] = U;
    }
}

// ---- binary -> 4-wide collapse + quantisation on the device (the rules: bvh_wide.h; the host routine collapseBvh4 gives the
// same bytes).  Level by level from the root: every wide node of the level absorbs its binary nodes (wideSlots) and reserves
// places for its inner children in the next level (one atomic per wavefront: a wave prefix sum inside).  The result must be
// numbered in DFS pre-order like the host's: subtree sizes bottom-up, then ids top-down (id of child k = id of the parent
// + 1 + sizes of the children before it), then every node is written -- full precision and quantised -- at its id.
struct WideTmp {
    crt_bvh_node4 W;     // inner refs = binary indices
    uint32_t child[4];   // temporary index (level order) of the inner children, ~0u for leaf / empty slots
};

__device__ __forceinline__ uint32_t waveExclusiveSum(uint32_t v, uint32_t& total)
{
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= static_cast<uint32_t>(off)) incl += up;
    }
    total = __shfl(incl, 63, 64);
    return incl - v;
}

__global__ __launch_bounds__(256) void wideExpandKernel(const crt_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ frontier /* {binary, depth} pairs */,
                                                        uint32_t count, uint32_t base, uint32_t nextBase, WideTmp* __restrict__ tmp,
                                                        uint32_t* __restrict__ nextFrontier, uint32_t* __restrict__ nextCount, uint32_t* __restrict__ maxDepth)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t inner = 0;
    WideSlot sl[4];
    int n = 0;
    if (i < count) {
        uint32_t deep = 0;
        n = wideSlots(nodes, static_cast<int32_t>(frontier[2u * i]), frontier[2u * i + 1u], sl, &deep);
        atomicMax(maxDepth, deep);
        for (int k = 0; k < n; k++) inner += sl[k].ref >= 0 ? 1u : 0u;
    }
    uint32_t total = 0;
    const uint32_t before = waveExclusiveSum(inner, total);
    uint32_t waveBase = 0;
    if ((threadIdx.x & 63u) == 0u && total) waveBase = atomicAdd(nextCount, total);
    waveBase = __shfl(waveBase, 0, 64);
    if (i >= count) return;
    WideTmp& T = tmp[base + i];
    fillWide(sl, n, T.W);
    uint32_t at = waveBase + before;
    for (int k = 0; k < 4; k++) {
        T.child[k] = ~0u;
        if (k < n && sl[k].ref >= 0) {
            nextFrontier[2u * at] = static_cast<uint32_t>(sl[k].ref);
            nextFrontier[2u * at + 1u] = sl[k].depth;
            T.child[k] = nextBase + at;
            at++;
        }
    }
}

__global__ __launch_bounds__(256) void wideSizeKernel(const WideTmp* __restrict__ tmp, uint32_t base, uint32_t count, uint32_t* __restrict__ size)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const WideTmp& T = tmp[base + i];
    uint32_t s = 1;
    for (int k = 0; k < 4; k++)
        if (T.child[k] != ~0u) s += size[T.child[k]];
    size[base + i] = s;
}

__global__ __launch_bounds__(256) void wideIdKernel(const WideTmp* __restrict__ tmp, uint32_t base, uint32_t count, const uint32_t* __restrict__ size,
                                                    uint32_t* __restrict__ id)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const WideTmp& T = tmp[base + i];
    uint32_t run = id[base + i] + 1u; // (the root's id is set to 0 by the host)
    for (int k = 0; k < 4; k++)
        if (T.child[k] != ~0u) {
            id[T.child[k]] = run;
            run += size[T.child[k]];
        }
}

__global__ __launch_bounds__(256) void wideEmitKernel(const WideTmp* __restrict__ tmp, uint32_t total, const uint32_t* __restrict__ id,
                                                      crt_bvh_node4* __restrict__ nodes4, crt_bvh_node4q* __restrict__ nodes4q)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total) return;
    crt_bvh_node4 W = tmp[t].W;
    for (int k = 0; k < 4; k++)
        if (tmp[t].child[k] != ~0u) W.ref[k] = static_cast<int32_t>(id[tmp[t].child[k]]);
    nodes4[id[t]] = W;
    crt_bvh_node4q Q;
    quantizeNode4(W, Q);
    nodes4q[id[t]] = Q;
}

struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) { GPU_TRY(hipMalloc(&p, bytes ? bytes : 16)); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <class T> T* as() const { return static_cast<T*>(p); }
    void* release() { void* q = p; p = nullptr; return q; } // ownership leaves (the context adopts the buffer)
};

} // namespace

void buildBvhGpu(const crt_mesh_view* meshes, uint32_t n_meshes, Bvh& out, ihipStream_t* stream, double* device_ms)
{
    const bool timing = std::getenv("CRT_BUILD_TIMING") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tlast = tnow();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t = tnow();
        std::fprintf(stderr, "[build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tlast).count());
        tlast = t;
    };
    out = Bvh();
    if (device_ms) *device_ms = 0.0;
    uint64_t total = 0, totalVerts = 0;
    bool anyNormals = false, anyUvs = false;
    for (uint32_t m = 0; m < n_meshes; m++) {
        if (meshes[m].n_triangles && (!meshes[m].xyz || !meshes[m].idx)) throw std::runtime_error("mesh with triangles but null vertex/index pointer");
        total += meshes[m].n_triangles;
        totalVerts += meshes[m].n_vertices;
        anyNormals |= meshes[m].normals != nullptr && meshes[m].n_triangles > 0;
        anyUvs |= meshes[m].uvs != nullptr && meshes[m].n_triangles > 0;
    }
    if (total >= (1ull << 28)) throw std::runtime_error("too many triangles (limit 2^28 - 1)");
    if (totalVerts >= (1ull << 32)) throw std::runtime_error("too many vertices");
    const uint32_t n = static_cast<uint32_t>(total);
    if (n == 0) return;

    if (n <= static_cast<uint32_t>(kLeafMax)) {
        // a handful of triangles: one leaf, wrapped in a node whose right child is an empty leaf with the same box (as the SAH
        // path does); nothing worth a kernel launch -- the host does it with the same rules
        std::vector<crt_bvh_tri> inTri;
        std::vector<crt_bvh_shade> inShade;
        std::vector<float> pboxCent; // per triangle: 6 floats box + 3 floats centroid
        flattenMeshes(meshes, n_meshes, inTri, inShade, pboxCent);
        std::vector<crt_bvh_uv> inUv;
        flattenUvs(meshes, n_meshes, inUv);
        std::vector<Box6> hBox(n);
        std::vector<float> hCent(3 * static_cast<size_t>(n));
        for (uint32_t i = 0; i < n; i++) {
            std::memcpy(&hBox[i], &pboxCent[9 * static_cast<size_t>(i)], sizeof(Box6));
            std::memcpy(&hCent[3 * static_cast<size_t>(i)], &pboxCent[9 * static_cast<size_t>(i) + 6], 3 * sizeof(float));
        }
        out.tris.resize(n);
        out.shade.resize(n);
        Box6 root = hBox[0];
        for (uint32_t i = 1; i < n; i++)
            for (int a = 0; a < 3; a++) {
                root.mn[a] = root.mn[a] < hBox[i].mn[a] ? root.mn[a] : hBox[i].mn[a];
                root.mx[a] = root.mx[a] > hBox[i].mx[a] ? root.mx[a] : hBox[i].mx[a];
            }
        crt_bvh_node N;
        N.lx0 = N.rx0 = root.mn[0]; N.lx1 = N.rx1 = root.mx[0]; N.ly0 = N.ry0 = root.mn[1]; N.ly1 = N.ry1 = root.mx[1];
        N.lz0 = N.rz0 = root.mn[2]; N.lz1 = N.rz1 = root.mx[2];
        N.left = ~static_cast<int32_t>((0u << 3) | n);
        N.right = ~static_cast<int32_t>(0);
        N.pad0 = N.pad1 = 0;
        out.nodes.push_back(N);
        std::vector<unsigned long long> keys(n);
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = hCent[a]; hi[a] = hCent[a]; }
        for (uint32_t i = 1; i < n; i++)
            for (int a = 0; a < 3; a++) {
                const float c = hCent[3 * i + a];
                lo[a] = lo[a] < c ? lo[a] : c;
                hi[a] = hi[a] > c ? hi[a] : c;
            }
        auto expand = [](uint32_t v) { v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v; };
        for (uint32_t i = 0; i < n; i++) {
            uint32_t q[3];
            float extm = 0.0f;
            for (int a = 0; a < 3; a++) {
                const float ext = hi[a] - lo[a];
                if (ext > extm) extm = ext;
            }
            const float scale = extm > 0.0f ? 1024.0f / extm : 0.0f;
            for (int a = 0; a < 3; a++) {
                const float f = (hCent[3 * i + a] - lo[a]) * scale;
                q[a] = f >= 0.0f ? (f < 1024.0f ? static_cast<uint32_t>(f) : 1023u) : 0u;
            }
            keys[i] = (static_cast<unsigned long long>((expand(q[0]) << 2) | (expand(q[1]) << 1) | expand(q[2])) << 32) | i;
        }
        std::sort(keys.begin(), keys.end());
        for (uint32_t i = 0; i < n; i++) {
            out.tris[i] = inTri[keys[i] & 0xFFFFFFFFull];
            out.shade[i] = inShade[keys[i] & 0xFFFFFFFFull];
        }
        out.nTris = n;
        out.maxDepth = 1;
        collapseBvh4(out);
        reorderUvs(inUv, out);
        return;
    }

    // ---- the meshes as they are, back to back
    std::vector<MeshEntry> table(n_meshes + 1u);
    std::vector<float> hXyz(3 * static_cast<size_t>(totalVerts)), hNormals, hUvs;
    std::vector<uint32_t> hIdx(3 * static_cast<size_t>(n));
    if (anyNormals) hNormals.assign(3 * static_cast<size_t>(totalVerts), 0.0f);
    if (anyUvs) hUvs.assign(3 * static_cast<size_t>(totalVerts), 0.0f);
    {
        uint32_t t0 = 0, v0 = 0;
        for (uint32_t m = 0; m < n_meshes; m++) {
            const crt_mesh_view& M = meshes[m];
            MeshEntry& E = table[m];
            E.triStart = t0; E.vertStart = v0; E.nVerts = M.n_vertices; E.material = static_cast<uint32_t>(M.material_index);
            E.hasNormals = M.normals ? 1u : 0u; E.hasUvs = M.uvs ? 1u : 0u; E.pad0 = E.pad1 = 0;
            if (M.n_vertices && M.xyz) std::memcpy(&hXyz[3 * static_cast<size_t>(v0)], M.xyz, sizeof(float) * 3 * M.n_vertices); // (a mesh without triangles may come without arrays)
            if (M.n_triangles && M.idx) std::memcpy(&hIdx[3 * static_cast<size_t>(t0)], M.idx, sizeof(uint32_t) * 3 * static_cast<size_t>(M.n_triangles));
            if (M.normals && M.n_vertices) std::memcpy(&hNormals[3 * static_cast<size_t>(v0)], M.normals, sizeof(float) * 3 * M.n_vertices);
            if (M.uvs && M.n_vertices) std::memcpy(&hUvs[3 * static_cast<size_t>(v0)], M.uvs, sizeof(float) * 3 * M.n_vertices);
            t0 += M.n_triangles;
            v0 += M.n_vertices;
        }
        MeshEntry& E = table[n_meshes];
        E.triStart = t0; E.vertStart = v0; E.nVerts = 0; E.material = 0; E.hasNormals = E.hasUvs = E.pad0 = E.pad1 = 0;
    }
    lap("host concatenate");

    const uint32_t nInternal = n - 1;
    DevBuf dTable(sizeof(MeshEntry) * table.size()), dXyz(sizeof(float) * hXyz.size()), dIdx(sizeof(uint32_t) * hIdx.size());
    DevBuf dNormals(sizeof(float) * hNormals.size()), dUvsIn(sizeof(float) * hUvs.size()), dBad(sizeof(int));
    DevBuf dBox(sizeof(Box6) * n), dCent(sizeof(float) * 3 * n), dBounds(sizeof(int) * 6);
    DevBuf dKeysIn(sizeof(unsigned long long) * n), dKeys(sizeof(unsigned long long) * n);
    DevBuf dK(sizeof(KNode) * nInternal), dParI(sizeof(int) * nInternal), dParL(sizeof(int) * n);
    DevBuf dNodeBox(sizeof(Box6) * nInternal), dFlags(sizeof(unsigned int) * nInternal);
    DevBuf dKept(sizeof(uint32_t) * nInternal), dRank(sizeof(uint32_t) * nInternal);
    // the records the kernels will traverse (+64 bytes of slack for speculative wide loads of the last record)
    DevBuf dTris(sizeof(crt_bvh_tri) * n + 64), dShade(sizeof(crt_bvh_shade) * n + 64), dUvs(anyUvs ? sizeof(crt_bvh_uv) * n + 64 : 0);
    // scratch of the sort (digit counts per tile) and of the scan (tile sums); both written by this file's own kernels (gpu_sort.hip.h)
    DevBuf dSortCounts(sizeof(uint32_t) * gpusort::sortScratchWords(n)), dScanSums(gpusort::scanScratchBytes(nInternal));

    GPU_TRY(hipMemcpyAsync(dTable.p, table.data(), sizeof(MeshEntry) * table.size(), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipMemcpyAsync(dXyz.p, hXyz.data(), sizeof(float) * hXyz.size(), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipMemcpyAsync(dIdx.p, hIdx.data(), sizeof(uint32_t) * hIdx.size(), hipMemcpyHostToDevice, stream));
    if (anyNormals) GPU_TRY(hipMemcpyAsync(dNormals.p, hNormals.data(), sizeof(float) * hNormals.size(), hipMemcpyHostToDevice, stream));
    if (anyUvs) GPU_TRY(hipMemcpyAsync(dUvsIn.p, hUvs.data(), sizeof(float) * hUvs.size(), hipMemcpyHostToDevice, stream));
    GPU_TRY(hipMemsetAsync(dBad.p, 0, sizeof(int), stream));
    if (timing) { GPU_TRY(hipStreamSynchronize(stream)); }
    lap("alloc + H2D meshes");

    struct Events { // destroyed on every way out, exceptions included
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } ev;
    GPU_TRY(hipEventCreate(&ev.e0));
    GPU_TRY(hipEventCreate(&ev.e1));
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    GPU_TRY(hipEventRecord(e0, stream));
    const dim3 blk(256), grdN((n + 255) / 256), grdI((nInternal + 255) / 256);
    hipLaunchKernelGGL(triBoxKernel, grdN, blk, 0, stream, dTable.as<MeshEntry>(), n_meshes, dXyz.as<float>(), dIdx.as<uint32_t>(), n, dBox.as<Box6>(),
                       dCent.as<float>(), dBad.as<int>());
    hipLaunchKernelGGL(initBoundsKernel, dim3(1), dim3(64), 0, stream, dBounds.as<int>());
    hipLaunchKernelGGL(boundsKernel, dim3(kBoundsBlocks), blk, 0, stream, dCent.as<float>(), n, dBounds.as<int>());
    // keys = Morton code << 32 | ordinal, written in ordinal order: a stable sort on the four bytes of the high dword orders the full keys;
    // four passes ping-pong dKeys -> dKeysIn -> ... and end in dKeys
    hipLaunchKernelGGL(mortonKernel, grdN, blk, 0, stream, dCent.as<float>(), n, dBounds.as<int>(), dKeys.as<unsigned long long>());
    GPU_TRY(hipGetLastError()); // triBox / bounds / Morton launches
    hipError_t sortStatus = hipSuccess;
    unsigned long long* sorted = gpusort::sortKeysHigh(dKeys.as<unsigned long long>(), dKeysIn.as<unsigned long long>(), n, 4, dSortCounts.as<uint32_t>(), stream, &sortStatus);
    GPU_TRY(sortStatus);
    if (sorted != dKeys.as<unsigned long long>()) throw std::logic_error("sorted keys expected in the first buffer");
    hipLaunchKernelGGL(hierarchyKernel, grdI, blk, 0, stream, dKeys.as<unsigned long long>(), n, dK.as<KNode>(), dParI.as<int>(), dParL.as<int>());
    GPU_TRY(hipMemsetAsync(dFlags.p, 0, sizeof(unsigned int) * nInternal, stream));
    hipLaunchKernelGGL(fitKernel, grdN, blk, 0, stream, dK.as<KNode>(), dBox.as<Box6>(), dKeys.as<unsigned long long>(), n, dParI.as<int>(),
                       dParL.as<int>(), dNodeBox.as<Box6>(), dFlags.as<unsigned int>());
    hipLaunchKernelGGL(keptKernel, grdI, blk, 0, stream, dK.as<KNode>(), nInternal, dKept.as<uint32_t>());
    GPU_TRY(hipGetLastError()); // hierarchy / fit / kept launches
    GPU_TRY(gpusort::exclusiveSum(dKept.as<uint32_t>(), dRank.as<uint32_t>(), nInternal, dScanSums.as<uint32_t>(), stream));
    uint32_t lastKept = 0, lastRank = 0;
    int bad = 0;
    GPU_TRY(hipMemcpyAsync(&lastKept, dKept.as<uint32_t>() + (nInternal - 1), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&lastRank, dRank.as<uint32_t>() + (nInternal - 1), 4, hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipMemcpyAsync(&bad, dBad.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    if (bad) throw std::runtime_error("triangle index out of range");
    const uint32_t nKept = lastKept + lastRank;
    DevBuf dNodes(sizeof(crt_bvh_node) * nKept);
    hipLaunchKernelGGL(emitKernel, grdI, blk, 0, stream, dK.as<KNode>(), dKept.as<uint32_t>(), dRank.as<uint32_t>(), dNodeBox.as<Box6>(), dBox.as<Box6>(),
                       dKeys.as<unsigned long long>(), nInternal, dNodes.as<crt_bvh_node>());
    hipLaunchKernelGGL(gatherKernel, grdN, blk, 0, stream, dKeys.as<unsigned long long>(), n, dTable.as<MeshEntry>(), n_meshes, dXyz.as<float>(),
                       dIdx.as<uint32_t>(), dNormals.as<float>(), dUvsIn.as<float>(), dTris.as<crt_bvh_tri>(), dShade.as<crt_bvh_shade>(),
                       anyUvs ? dUvs.as<crt_bvh_uv>() : nullptr);
    GPU_TRY(hipGetLastError());
    if (timing) { GPU_TRY(hipStreamSynchronize(stream)); }
    lap("device build");
    // ---- collapse to the 4-wide tree and quantise, still on the device
    DevBuf dTmp(sizeof(WideTmp) * nKept), dSize(sizeof(uint32_t) * nKept), dId(sizeof(uint32_t) * nKept);
    DevBuf dFrontA(sizeof(uint32_t) * 2 * nKept), dFrontB(sizeof(uint32_t) * 2 * nKept), dScalars(sizeof(uint32_t) * 2); // [0] next count, [1] max depth
    std::vector<std::pair<uint32_t, uint32_t>> levels; // {first temporary index, count}
    {
        const uint32_t rootEntry[2] = { 0u, 0u };
        GPU_TRY(hipMemcpyAsync(dFrontA.p, rootEntry, sizeof(rootEntry), hipMemcpyHostToDevice, stream));
        GPU_TRY(hipMemsetAsync(dScalars.p, 0, sizeof(uint32_t) * 2, stream));
        uint32_t base = 0, count = 1;
        uint32_t* cur = dFrontA.as<uint32_t>();
        uint32_t* nxt = dFrontB.as<uint32_t>();
        while (count) {
            if (static_cast<uint64_t>(base) + count > nKept) throw std::runtime_error("wide collapse: more wide nodes than binary nodes");
            levels.emplace_back(base, count);
            GPU_TRY(hipMemsetAsync(dScalars.p, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(wideExpandKernel, dim3((count + 255) / 256), blk, 0, stream, dNodes.as<crt_bvh_node>(), cur, count, base, base + count,
                               dTmp.as<WideTmp>(), nxt, dScalars.as<uint32_t>(), dScalars.as<uint32_t>() + 1);
            GPU_TRY(hipGetLastError());
            uint32_t nextCount = 0;
            GPU_TRY(hipMemcpyAsync(&nextCount, dScalars.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            GPU_TRY(hipStreamSynchronize(stream));
            base += count;
            count = nextCount;
            std::swap(cur, nxt);
        }
    }
    const uint32_t nWide = levels.back().first + levels.back().second;
    for (size_t L = levels.size(); L-- > 0;)
        hipLaunchKernelGGL(wideSizeKernel, dim3((levels[L].second + 255) / 256), blk, 0, stream, dTmp.as<WideTmp>(), levels[L].first, levels[L].second, dSize.as<uint32_t>());
    GPU_TRY(hipMemsetAsync(dId.p, 0, sizeof(uint32_t), stream)); // the root's id
    for (size_t L = 0; L < levels.size(); L++)
        hipLaunchKernelGGL(wideIdKernel, dim3((levels[L].second + 255) / 256), blk, 0, stream, dTmp.as<WideTmp>(), levels[L].first, levels[L].second, dSize.as<uint32_t>(),
                           dId.as<uint32_t>());
    DevBuf dNodes4(sizeof(crt_bvh_node4) * nWide), dNodes4q(sizeof(crt_bvh_node4q) * nWide + 128);
    hipLaunchKernelGGL(wideEmitKernel, dim3((nWide + 255) / 256), blk, 0, stream, dTmp.as<WideTmp>(), nWide, dId.as<uint32_t>(), dNodes4.as<crt_bvh_node4>(),
                       dNodes4q.as<crt_bvh_node4q>());
    GPU_TRY(hipGetLastError());
    uint32_t maxDepth = 0;
    GPU_TRY(hipMemcpyAsync(&maxDepth, dScalars.as<uint32_t>() + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    GPU_TRY(hipEventRecord(e1, stream));
    GPU_TRY(hipStreamSynchronize(stream));
    float ms = 0.f;
    GPU_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (device_ms) *device_ms = ms;
    lap("device collapse + quantise");
    out.maxDepth = maxDepth;
    out.depth4 = static_cast<uint32_t>(levels.size());
    out.nNodes = nKept;
    out.nNodes4 = nWide;
    out.devNodes = dNodes.release();
    out.devNodes4 = dNodes4.release();
    out.devNodes4q = dNodes4q.release();
    // the leaf-ordered records stay in HBM: the caller adopts the buffers (and copies them out only if someone asks)
    out.nTris = n;
    out.devTris = dTris.release();
    out.devShade = dShade.release();
    out.devUvs = anyUvs ? dUvs.release() : nullptr;
}

} // namespace crt
