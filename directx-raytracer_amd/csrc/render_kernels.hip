// HIP kernels for gfx950 (MI355X): the reference's per-pixel DXR program -- rayGen -> TraceRay -> miss /
// closestHit -> image store (R/HLSL/ray_tracing_shaders.hlsl:21-169, dispatched by DispatchRays at
// R/DXRTRenderer.cpp:1405) -- fused into one kernel, with the BVH traversal and the ray/triangle test the
// reference leaves to the DXR driver written out.
//
// Mapping: one 64-thread workgroup = one wavefront = one 8x8-pixel ray packet (the work unit); four units form the
// 16x16 macro tile that is the unit of multi-GPU ownership.  Every lane owns one ray and a private traversal stack
// in LDS (entry e of lane l at dword e*64+l: conflict free, 16 entries = 4 KB per wavefront; deeper entries spill
// to a global arena).  The tree is 4-wide; a node is fetched per lane with dwordx4 loads, or once per wavefront
// through the scalar cache when all lanes stand on the same node; triangles are 48-byte records.  Units are
// launched most-expensive-first from the cost the previous frame measured.
//
// Arithmetic contract: identical, operation for operation, to oracle/crt_oracle.c (compiled with
// -ffp-contract=off; fused multiply-adds only where fmaf()/fma() is written; correctly rounded / and sqrt).
#include "split_packet.hip.h"

namespace crt {
namespace {

// PHONG: the specular term of mode 100 is compiled into its own variant (chosen at launch when "phong_ks" is non-zero), so
// that the plain Lambert kernel keeps its register budget
// pixel of a lane inside its 16x16 macro tile: 8x8 packet `wave`, or one 4x4 quarter of it (pixel = lane & 15: the lanes
// l, l + 16, l + 32, l + 48 of a split packet hold the four segments of the same ray)
__device__ __forceinline__ void lanePixel(uint32_t wave, uint32_t quarter, uint32_t lane, uint32_t& lx, uint32_t& ly)
{
    lx = (wave & 1u) * 8u;
    ly = (wave >> 1) * 8u;
    if (quarter < 4u) {
        lx += (quarter & 1u) * 4u + (lane & 3u);
        ly += (quarter >> 1) * 4u + ((lane >> 2) & 3u);
    } else {
        lx += lane & 7u;
        ly += lane >> 3;
    }
}

__device__ __forceinline__ uint32_t laneId() { return laneIndex(); }

// SPLIT: the variant that can render split packets (below); launched only when the option "split_units" is non-zero, so that
// the ordinary launch carries none of its code or registers (with it compiled into the one kernel, primary rays alone ran 5 %
// slower and the kernel spilled 14 registers).
// (The SPLIT variant is given 96 registers: the streams of a split quarter hold a ray in flight plus the hand-out state, and a
// launch that splits packets is one whose chip is not full -- an N-rank tile share -- so five wavefronts per SIMD cost it nothing.)
template <bool COUNT, bool PHONG, class L, bool SPLIT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SPLIT ? 5 : L::kWavesPerEu, 8))) void renderKernel(const RenderParams p)
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
    // Split packets (split_packet.hip.h): the split_units most expensive packets are rendered by 64 / R wavefronts each, one per
    // block of R = 1 << split_rays_log2 pixels, whose 64 lanes share the work of the block's rays.
    uint32_t quarter = 64u; // part of a split packet; 64 = the whole 8x8 packet
    if (p.unit_order) {
        uint32_t pos = b;
        if (SPLIT) {
            const uint32_t partsLog2 = 6u - p.split_rays_log2;
            if (b < (p.split_units << partsLog2)) {
                pos = b >> partsLog2;
                quarter = b & ((1u << partsLog2) - 1u);
            } else {
                pos = b - ((p.split_units << partsLog2) - p.split_units);
            }
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
        if (p.unit_cost && frame == 0u && threadIdx.x == 0 && quarter >= 64u) p.unit_cost[unit] = 0;
        return;
    }

    const uint32_t tid = threadIdx.x, wave = unit & 3u, lane = tid & 63u;
    uint32_t lx, ly;
    lanePixel(wave, 4u, lane, lx, ly);
    const uint32_t px = tile_x * kTile + lx, py = tile_y * kTile + ly;
    const bool active = (px < p.width) & (py < p.height);
    const bool split = SPLIT && quarter < 64u; // wave-uniform

    uint32_t cntNodes = 0, cntTris = 0, cntShadow = 0, cntClosest = 0;
#if CRT_PROF
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    unsigned long long pTNode = 0, pTLeaf = 0; uint32_t pItN = 0, pItL = 0, pLaN = 0, pLaL = 0;
    uint32_t pDv[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    uint32_t iters = 0; // traversal-loop iterations of this wavefront = its critical path, fed back as next frame's cost
    if (split) {
        Stack stack;
        stack.lds = s_stack + tid;
        // (a split packet's four wavefronts take the 64-lane slices behind those of the frame's ordinary units)
        stack.spill = p.spill + ((static_cast<size_t>(p.units_per_frame) * p.n_batch + static_cast<size_t>(b) * p.n_batch + frame) * 64u + tid) * p.spill_stride;
        stack.cap = static_cast<int>(p.stack_entries);
        stack.sp = 0;
        renderSplitQuarter<COUNT, PHONG, L>(p, camPos, camRot, outRgba8, frame, tile_x, tile_y, wave, quarter, p.split_rays_log2, p.split_segs_log2, stack, iters, cntNodes, cntTris, cntShadow, cntClosest);
    } else if (active) {
        const float4* nodes = reinterpret_cast<const float4*>(p.nodes);
        const float4* tris = reinterpret_cast<const float4*>(p.tris);
        Stack stack;
        stack.lds = s_stack + tid;
        stack.spill = p.spill + ((static_cast<size_t>(frame) * p.units_per_frame + unit) * 64u + tid) * p.spill_stride;
        stack.cap = static_cast<int>(p.stack_entries);
        stack.sp = 0;

        F3 col;
        uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
        Hit h;
        {
            const F3 o = f3(camPos[0], camPos[1], camPos[2]);
            const Ray r = makeRay(o, rayDir(camRot, px, py, static_cast<float>(p.width), static_cast<float>(p.height)));
            if (COUNT) cntClosest++;
            traceClosest<COUNT, L>(nodes, tris, p.n_nodes, r, kTMin, kTMax, stack, static_cast<int>(p.tune_inner_min), h, iters, cntNodes, cntTris);
            col = f3(p.miss[0], p.miss[1], p.miss[2]); // miss shader (hlsl:72-76)
            if (h.t < kTMax) {
                const float4* T = L::triPtr(tris, h.tri);
                if (p.mode >= 100u) col = shadeLambert<COUNT, L, PHONG>(p, nodes, tris, r, h, stack, iters, cntNodes, cntTris, cntShadow);
                else col = shadeDebug(p.mode, __float_as_uint(T[0].w), __float_as_uint(T[1].w), h.t, h.u, h.v, r.o, r.d);
            }
        }
        const bool hit = h.t < kTMax;
        if (hit) {
            const float4* T = L::triPtr(tris, h.tri);
            inst = __float_as_uint(T[0].w);
            prim = __float_as_uint(T[1].w);
        }

#if CRT_PROF
        pTNode = stack.tNode; pTLeaf = stack.tLeaf; pItN = stack.itNode; pItL = stack.itLeaf; pLaN = stack.lanesNode; pLaL = stack.lanesLeaf;
        pDv[0] = stack.dvN; pDv[1] = stack.dvNLanes; pDv[2] = stack.dvNRuns; pDv[3] = stack.dvNDistinct;
        pDv[4] = stack.dvL; pDv[5] = stack.dvLLanes; pDv[6] = stack.dvLRuns; pDv[7] = stack.dvLDistinct;
#endif
        const uint32_t packed = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
        // where the pixel goes: recomputed from a fresh lane index, so that nothing of it is held (it was spilled) during the traversal
        uint32_t lx, ly;
        lanePixel(wave, 4u, laneId(), lx, ly);
        const uint32_t px = tile_x * kTile + lx, py = tile_y * kTile + ly;
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
        if (quarter < 64u) atomicAdd(&p.unit_cost[unit], life);
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

int launchRender(const RenderParams& p, bool counting, ihipStream_t* stream)
{
    if (p.n_local_tiles == 0) return 0;
    // list length: single GPU walks whole 4x4-tile blocks (padded at the frame edges); then padded to 8 XCDs x kGroupMax
    const uint32_t n = renderUnitCount(p) / 4u;
    const dim3 block(64);
    const size_t lds = static_cast<size_t>(p.stack_entries) * 64u * sizeof(int);
    if (p.mode >= 200u) return launchPath(p, counting, stream); // path_kernels.hip
    {
        const uint32_t extra = p.unit_order && p.split_units ? ((p.split_units << (6u - p.split_rays_log2)) - p.split_units) : 0u; // the further parts of split packets
        const dim3 grid((n * 4u + extra) * (p.n_batch ? p.n_batch : 1u));
        const bool phong = p.mode >= 100u && p.phong_ks > 0.0f;
        const bool split = p.unit_order && p.split_units > 0u;
#define CRT_LAUNCH2(LAY, SPL)                                                                                          \
        if (counting && phong) hipLaunchKernelGGL((renderKernel<true, true, LAY, SPL>), grid, block, lds, stream, p);  \
        else if (counting) hipLaunchKernelGGL((renderKernel<true, false, LAY, SPL>), grid, block, lds, stream, p);     \
        else if (phong) hipLaunchKernelGGL((renderKernel<false, true, LAY, SPL>), grid, block, lds, stream, p);        \
        else hipLaunchKernelGGL((renderKernel<false, false, LAY, SPL>), grid, block, lds, stream, p);
#define CRT_LAUNCH(LAY) if (split) { CRT_LAUNCH2(LAY, true) } else { CRT_LAUNCH2(LAY, false) }
#if CRT_PACKED_LAYOUTS
        if (p.layout == 8u) { CRT_LAUNCH(LayPacked<8>) }
        else if (p.layout == 4u) { CRT_LAUNCH(LayPacked<4>) }
        else
#endif
        { CRT_LAUNCH(LayLegacy) }
#undef CRT_LAUNCH
#undef CRT_LAUNCH2
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

// XCD-affine launch order.  Workgroup b runs on XCD b % 8, and each XCD has its own L2: the unit list (which walks the frame
// in 64x64-pixel blocks) is cut into eight contiguous regions of equal total cost, each region is sorted by descending cost,
// and the j-th unit of region x is launched at position 8 j + x -- so an XCD keeps working inside one part of the frame (and
// of the scene) while every XCD still starts with its own most expensive packets.  Regions hold different numbers of units;
// the cheapest units of the longer regions fill the last rounds of the shorter ones, so the order stays a permutation of
// 0..n-1 (n is a multiple of 512); 'equal cost' therefore means equal cost above that of such a filler unit.  Region boundaries fall on multiples of n / 256 units.
__global__ __launch_bounds__(kSortThreads) void sortUnitsAffineKernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n)
{
    __shared__ uint32_t hist[8 * kSortBuckets];
    __shared__ uint32_t sums[kSortThreads];
    __shared__ uint32_t chunkRegion[kSortThreads];
    __shared__ uint32_t cnt[8], surplusBase[8], deficitBase[8], deficit[8];
    const uint32_t t = threadIdx.x;
    const uint32_t chunk = n / kSortThreads, rounds = n / 8u;
    __shared__ uint32_t fillCost;
    for (uint32_t b = t; b < 8u * kSortBuckets; b += kSortThreads) hist[b] = 0;
    __syncthreads();
    // Every XCD runs exactly n / 8 units, so a region with fewer units than that is topped up with the cheapest units of the
    // frame, of cost ~ fillCost each (the 1/16 quantile): what has to be equal across the regions is the cost above that.
    for (uint32_t i = t; i < n; i += kSortThreads) atomicAdd(&hist[min(cost[i], kSortBuckets - 1u)], 1u);
    __syncthreads();
    if (t == 0) {
        uint32_t acc = 0, b = 0;
        while (b + 1u < kSortBuckets && acc + hist[b] < n / 16u) acc += hist[b++];
        fillCost = b;
    }
    __syncthreads();
    for (uint32_t b = t; b < kSortBuckets; b += kSortThreads) hist[b] = 0;
    const uint32_t fc = fillCost;
    uint32_t mine = 0;
    for (uint32_t i = t * chunk; i < (t + 1u) * chunk; i++) mine += cost[i] > fc ? cost[i] - fc : 0u;
    sums[t] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < kSortThreads; d <<= 1) {
        const uint32_t add = t >= d ? sums[t - d] : 0u;
        __syncthreads();
        sums[t] += add;
        __syncthreads();
    }
    const uint32_t total = sums[kSortThreads - 1u];
    {
        const unsigned long long mid = static_cast<unsigned long long>(sums[t] - mine) + mine / 2u;
        const uint32_t r = total ? static_cast<uint32_t>(mid * 8ull / total) : t / (kSortThreads / 8u);
        chunkRegion[t] = r < 7u ? r : 7u;
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += kSortThreads) atomicAdd(&hist[chunkRegion[i / chunk] * kSortBuckets + min(cost[i], kSortBuckets - 1u)], 1u);
    __syncthreads();
    // per region: most expensive bucket first; 32 threads per region, 32 reversed buckets each
    const uint32_t reg = t >> 5, sub = t & 31u;
    uint32_t tot = 0;
    for (uint32_t k = 0; k < 32u; k++) tot += hist[reg * kSortBuckets + kSortBuckets - 1u - (sub * 32u + k)];
    __syncthreads();
    sums[t] = tot;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < sub; k++) base += sums[reg * 32u + k];
    if (sub == 31u) cnt[reg] = base + tot;
    for (uint32_t k = 0; k < 32u; k++) {
        uint32_t& h = hist[reg * kSortBuckets + kSortBuckets - 1u - (sub * 32u + k)];
        const uint32_t c = h;
        h = base;
        base += c;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t sb = 0, db = 0;
        for (uint32_t x = 0; x < 8u; x++) {
            surplusBase[x] = sb;
            deficitBase[x] = db;
            sb += cnt[x] > rounds ? cnt[x] - rounds : 0u;
            deficit[x] = cnt[x] < rounds ? rounds - cnt[x] : 0u;
            db += deficit[x];
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += kSortThreads) {
        const uint32_t x = chunkRegion[i / chunk];
        const uint32_t j = atomicAdd(&hist[x * kSortBuckets + min(cost[i], kSortBuckets - 1u)], 1u);
        uint32_t pos;
        if (j < rounds) {
            pos = j * 8u + x;
        } else {
            const uint32_t sidx = surplusBase[x] + (j - rounds);
            uint32_t y = 0;
            while (y < 7u && sidx >= deficitBase[y] + deficit[y]) y++;
            pos = (cnt[y] + (sidx - deficitBase[y])) * 8u + y;
        }
        order[pos] = i;
    }
}

int launchSortUnits(const uint32_t* cost, uint32_t* order, uint32_t n, bool xcdAffine, ihipStream_t* stream)
{
    if (xcdAffine && n % (2u * kSortThreads) == 0u) hipLaunchKernelGGL(sortUnitsAffineKernel, dim3(1), dim3(kSortThreads), 0, stream, cost, order, n);
    else hipLaunchKernelGGL(sortUnitsKernel, dim3(1), dim3(kSortThreads), 0, stream, cost, order, n);
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
