// Device-wide primitives of the GPU acceleration-structure build, written for wave64 / gfx950: a stable LSD radix sort of
// 64-bit keys on a bit range of their high dword, and an exclusive prefix sum of 32-bit counts.  Included by bvh_gpu.hip.
//
// Sort: the build's keys are (30-bit Morton code << 32) | input ordinal, and they arrive in ordinal order, so a STABLE sort
// on the Morton bits alone gives the order of the full 64-bit keys: four 8-bit passes over bits 32..63 instead of eight.
// One pass = three launches over tiles of 2048 keys (256 threads x 8):
//   histogram  per tile: 256 digit counts (LDS atomics) -> counts[digit][tile]
//   scan       exclusive sum of counts in (digit, tile) order (the prefix sum below) = where each tile's keys of each
//              digit start in the output
//   scatter    per tile: every wavefront walks its 512 keys 64 at a time in position order; lanes holding the same digit
//              find each other with eight ballots (one per digit bit), rank = earlier peers in the wavefront-chunk + the
//              digit's running count of this wavefront (LDS) + the counts of the tile's earlier wavefronts + the tile's
//              start from the scan.  Position order is kept at every level, so the pass is stable.
// Keys move 3 x 8 bytes per pass; at 1M..5M keys the sort is a small part of the build.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace crt {
namespace gpusort {

constexpr uint32_t kThreads = 256, kItems = 8, kTile = kThreads * kItems, kDigits = 256, kWaves = kThreads / 64;

__device__ __forceinline__ uint32_t digitOf(unsigned long long key, uint32_t shift) { return static_cast<uint32_t>(key >> shift) & 0xFFu; }

__global__ __launch_bounds__(kThreads) void histogramKernel(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t shift,
                                                           uint32_t nTiles, uint32_t* __restrict__ counts /* [digit][tile] */)
{
    __shared__ uint32_t hist[kDigits];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kTile;
#pragma unroll
    for (uint32_t j = 0; j < kItems; j++) {
        const uint32_t i = base + j * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&hist[digitOf(keys[i], shift)], 1u);
    }
    __syncthreads();
    counts[threadIdx.x * nTiles + blockIdx.x] = hist[threadIdx.x];
}

// exclusive sum of v[0..m) in place, one workgroup: for the few hundred tile sums of the prefix sum below
constexpr uint32_t kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void scanCountsKernel(uint32_t* __restrict__ v, uint32_t m)
{
    __shared__ uint32_t part[kScanThreads];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (m + kScanThreads - 1u) / kScanThreads;
    const uint32_t lo = t * per, hi = lo + per < m ? lo + per : m;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += v[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < kScanThreads; d <<= 1) {
        const uint32_t add = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t c = v[i];
        v[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(kThreads) void scatterKernel(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                         uint32_t n, uint32_t shift, uint32_t nTiles, const uint32_t* __restrict__ starts /* [digit][tile] */)
{
    __shared__ uint32_t run[kWaves][kDigits]; // per wavefront: keys of each digit seen so far
    __shared__ uint32_t tileStart[kDigits];
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    for (uint32_t w = 0; w < kWaves; w++) run[w][t] = 0;
    tileStart[t] = starts[t * nTiles + blockIdx.x];
    __syncthreads();
    const unsigned long long ltMask = lane ? (~0ull >> (64u - lane)) : 0ull;
    const uint32_t waveBase = blockIdx.x * kTile + wave * (64u * kItems);
    unsigned long long key[kItems];
    uint32_t rank[kItems]; // position among this wavefront's keys of the same digit
#pragma unroll
    for (uint32_t j = 0; j < kItems; j++) {
        const uint32_t i = waveBase + j * 64u + lane;
        const bool live = i < n;
        key[j] = live ? in[i] : 0ull;
        const uint32_t d = digitOf(key[j], shift);
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (uint32_t b = 0; b < 8u; b++) {
            const unsigned long long has = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? has : ~has;
        }
        uint32_t before = 0;
        if (live) {
            const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(peers))) - 1u;
            uint32_t old = 0;
            if (lane == leader) {
                old = run[wave][d];
                run[wave][d] = old + static_cast<uint32_t>(__popcll(peers));
            }
            old = __shfl(old, static_cast<int>(leader), 64);
            before = old + static_cast<uint32_t>(__popcll(peers & ltMask));
        }
        rank[j] = before;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kItems; j++) {
        const uint32_t i = waveBase + j * 64u + lane;
        if (i < n) {
            const uint32_t d = digitOf(key[j], shift);
            uint32_t pos = tileStart[d] + rank[j];
            for (uint32_t w = 0; w < wave; w++) pos += run[w][d];
            out[pos] = key[j];
        }
    }
}

inline uint32_t tilesFor(uint32_t n) { return (n + kTile - 1u) / kTile; }

// ---- exclusive prefix sum of 32-bit counts (compaction ranks): tile sums, one-workgroup scan of the sums, rescan of each tile
__global__ __launch_bounds__(kThreads) void tileSumKernel(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t part[kWaves];
    const uint32_t base = blockIdx.x * kTile + threadIdx.x * kItems;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kItems; j++) s += base + j < n ? v[base + j] : 0u;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < kWaves; w++) tot += part[w];
        sums[blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(kThreads) void tileScanKernel(const uint32_t* v, uint32_t n, const uint32_t* __restrict__ tileStarts,
                                                          uint32_t* out /* may be v: a thread reads its eight entries before it writes them */)
{
    __shared__ uint32_t part[kWaves];
    const uint32_t base = blockIdx.x * kTile + threadIdx.x * kItems;
    uint32_t x[kItems], s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kItems; j++) {
        x[j] = base + j < n ? v[base + j] : 0u;
        s += x[j];
    }
    uint32_t incl = s;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= static_cast<uint32_t>(off)) incl += up;
    }
    if ((threadIdx.x & 63u) == 63u) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = tileStarts[blockIdx.x] + incl - s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += part[w];
#pragma unroll
    for (uint32_t j = 0; j < kItems; j++) {
        if (base + j < n) out[base + j] = run;
        run += x[j];
    }
}

inline size_t scanScratchBytes(uint32_t n) { return sizeof(uint32_t) * tilesFor(n); }

// (both host routines return the first launch error of their kernels: hipSuccess, or what hipGetLastError reported right
// after the launch that failed)
inline hipError_t exclusiveSum(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tileSums, hipStream_t stream)
{
    const uint32_t nTiles = tilesFor(n);
    hipError_t e;
    hipLaunchKernelGGL(tileSumKernel, dim3(nTiles), dim3(kThreads), 0, stream, in, n, tileSums);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(scanCountsKernel, dim3(1), dim3(kScanThreads), 0, stream, tileSums, nTiles);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(tileScanKernel, dim3(nTiles), dim3(kThreads), 0, stream, in, n, tileSums, out);
    return hipGetLastError();
}

// scratch of the sort in uint32: digit counts per tile, then the tile sums of their scan
inline size_t sortScratchWords(uint32_t n) { return static_cast<size_t>(kDigits) * tilesFor(n) + tilesFor(kDigits * tilesFor(n)); }

// Stable sort of keys[0..n) on bits [32, 32 + 8 * passes): result in `keys` or `alt` (returned).  counts: sortScratchWords(n) uint32.
// (__shfl of a value held only by the leader, __ballot and the LDS counters all stay inside one wavefront: no barrier
// between the chunks of a wavefront is needed.)
inline unsigned long long* sortKeysHigh(unsigned long long* keys, unsigned long long* alt, uint32_t n, int passes, uint32_t* counts, hipStream_t stream,
                                        hipError_t* status)
{
    const uint32_t nTiles = tilesFor(n);
    unsigned long long* src = keys;
    unsigned long long* dst = alt;
    *status = hipSuccess;
    for (int p = 0; p < passes; p++) {
        const uint32_t shift = 32u + 8u * static_cast<uint32_t>(p);
        hipLaunchKernelGGL(histogramKernel, dim3(nTiles), dim3(kThreads), 0, stream, src, n, shift, nTiles, counts);
        if ((*status = hipGetLastError()) != hipSuccess) return src;
        if ((*status = exclusiveSum(counts, counts, kDigits * nTiles, counts + static_cast<size_t>(kDigits) * nTiles, stream)) != hipSuccess) return src; // in place
        hipLaunchKernelGGL(scatterKernel, dim3(nTiles), dim3(kThreads), 0, stream, src, dst, n, shift, nTiles, counts);
        if ((*status = hipGetLastError()) != hipSuccess) return src;
        unsigned long long* x = src; src = dst; dst = x;
    }
    return src;
}

} // namespace gpusort
} // namespace crt
