// Microbenchmark: what one vector-memory wave-instruction costs the L1 address/data path (TA/TCP) on gfx950, by width,
// by the number of distinct cache lines its lanes touch and by the number of active lanes.  All data L1/L2 resident (64 KB).
// Every CU runs `waves` wavefronts; each issues `iters` x 8 independent loads back to back (no dependent chain: throughput).
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/ta_cost.hip -o gpurun_out/ta_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WIDTH> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<3> { typedef float3 T; };
template <> struct Vec<4> { typedef float4 T; };

template <int WIDTH>
__global__ __launch_bounds__(64) void loadKernel(const float* __restrict__ data, int strideBytes, int groupLanes, int activeLanes, int iters, float* out)
{
    typedef typename Vec<WIDTH>::T V;
    const int lane = threadIdx.x;
    if (lane >= activeLanes) return;
    // lanes of a group share an address; groups are strideBytes apart
    // groupLanes > 0: neighbouring lanes share; < 0: lanes l, l + |g|, l + 2|g|, ... share (equal addresses far apart in the wavefront)
    const int grp = groupLanes > 0 ? lane / groupLanes : lane % (-groupLanes);
    const char* base = reinterpret_cast<const char*>(data) + grp * strideBytes;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const V v = *reinterpret_cast<const V*>(base + ((i * 8 + k) & 15) * 4096); // 16 different 4-KB pages: 64 KB, L1/L2 resident
#pragma unroll
            for (int c = 0; c < WIDTH; c++) acc += reinterpret_cast<const float*>(&v)[c]; // every component is used: the load keeps its width
            asm volatile("" ::: "memory");
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int WIDTH>
static double run(const float* d, int stride, int group, int active, float* out, int waves)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    loadKernel<WIDTH><<<256 * waves, 64>>>(d, stride, group, active, 10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loadKernel<WIDTH><<<256 * waves, 64>>>(d, stride, group, active, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // cycles per wave-instruction per CU: time * 2.4 GHz / (instructions per CU)
    const double instrPerCu = double(waves) * iters * 8;
    return ms * 1e-3 * 2.4e9 / instrPerCu;
}

int main()
{
    float* d; float* out;
    hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20); hipMalloc(&out, 64);
    const int waves = 16; // per CU: enough to saturate the L1 path
    printf("cycles per wave-instruction per CU (16 wavefronts per CU, independent loads, data cache resident)\n");
    printf("%-8s %-28s %8s\n", "width", "pattern", "cycles");
    struct P { const char* name; int stride, group, active; };
    const P pats[] = {
        { "all lanes same address", 0, 64, 64 },
        { "coalesced (16 B apart)", 16, 1, 64 },
        { "4 lanes/64-B record, 16 recs", 64, 4, 64 },
        { "8 groups of 8, 128 B apart", 128, 8, 64 },
        { "16 groups of 4, 128 B apart", 128, 4, 64 },
        { "64 lanes, 64 B apart", 64, 1, 64 },
        { "64 lanes, 128 B apart", 128, 1, 64 },
        { "64 lanes, 256 B apart", 256, 1, 64 },
        { "32 active, 128 B apart", 128, 1, 32 },
        { "16 active, 128 B apart", 128, 1, 16 },
        { "32 active, same address", 0, 64, 32 },
        { "8 addresses interleaved (l%8)", 128, -8, 64 },
        { "16 addresses interleaved (l%16)", 128, -16, 64 },
        { "19 runs ~ 3 lanes each, 64 B", 64, 3, 57 },
    };
    for (const P& p : pats) {
        printf("%-8s %-28s %8.1f\n", "dword", p.name, run<1>(d, p.stride, p.group, p.active, out, waves));
        printf("%-8s %-28s %8.1f\n", "dwordx2", p.name, run<2>(d, p.stride, p.group, p.active, out, waves));
        printf("%-8s %-28s %8.1f\n", "dwordx3", p.name, run<3>(d, p.stride, p.group, p.active, out, waves));
        printf("%-8s %-28s %8.1f\n", "dwordx4", p.name, run<4>(d, p.stride, p.group, p.active, out, waves));
    }
    return 0;
}
