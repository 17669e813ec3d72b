// C ABI of libcrt_hip.so (include/crt_hip.h): renderer context over the HIP runtime + scene-layer accessors.
// Each entry point cites the reference member it replaces in the header.  No CPU fallback exists here: every
// render path ends in launchRender() (render_kernels.hip; mode 200: path_kernels.hip).
#include "mem_util.h"
#include "../../include/crt_hip.h"

#include "bvh_build.h"
#include "bvh_wide.h"
#include "render_kernels.h"
#include "scene.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h> // types and prototypes only: the library is dlopen'ed by crt_comm_init (no link-time dependency)

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

using crt::RenderParams;

// Diagnostics that change what a frame does or costs ("timeline", "debug_skip_units", "debug_force_measure") exist only in
// the diagnostic builds (tools/diag_build.sh, tools/prof_build.sh: -DCRT_DIAG=1); the product rejects the option names.
#ifndef CRT_DIAG
#ifdef CRT_PROF
#define CRT_DIAG 1
#else
#define CRT_DIAG 0
#endif
#endif

struct crt_scene {
    crt::Scene scene;
    // uint32 copies of the int index vectors are not needed: std::vector<int> is reinterpreted like the
    // reference does for its index buffers (R/DXRTRenderer.cpp:314-315)
};

struct crt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    std::string error;

    crt::Bvh bvh; // host copy of what sits in HBM
    void* dNodes = nullptr;     // quantised wide nodes: what the kernels traverse
    void* dBinNodes = nullptr;  // gpu_build only: the binary tree and the full-precision wide tree as the builder left them in
    void* dWideNodes = nullptr; // HBM (no host copy exists; crt_bvh_export* read them back)
    void* dTris = nullptr;
    void* dShade = nullptr;
    void* dLights = nullptr;
    void* dMats = nullptr;
    void* dUvs = nullptr;
    void* dTextures = nullptr;
    void* dTexels = nullptr;
    uint32_t nTextures = 0;
    uint32_t nLights = 0, nMats = 0;
    bool haveScene = false;
    bool gpuBuild = false;      // option "gpu_build": LBVH on the device instead of the host SAH builder
    uint32_t bvhWidth = 0;      // option "bvh_width": 0 = legacy 64-byte 4-wide nodes, 4 / 8 = packed wide tree (bvh_pack.h), at the next upload
    double buildMs = 0.0;       // wall time of the last crt_upload_scene (build + upload)
    double buildDeviceMs = 0.0; // of which GPU kernels (gpu_build only)
    uint32_t sceneSerial = 0;
    float sceneLo[3] = { 0.f, 0.f, 0.f }, sceneHi[3] = { 0.f, 0.f, 0.f }; // box of the tree's root: where split rays are cut into segments

    float pos[3] = { 0.f, 0.f, 0.f };
    float rot[9] = { 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f };
    float miss[3] = { 0.f, 1.f, 1.f }; // hlsl:75
    uint32_t mode = 0;                 // R/DXRTRenderer.h:246 default shading mode
    bool counting = false;
    uint32_t pathSpp = 4, pathBounces = 3, pathSeed = 1234; // mode 200 (BASELINE.json configs[4]: 4 spp, 3 bounces)
    uint32_t phongKsPermille = 0, phongExp = 32;            // mode 100: specular term, off by default
    uint32_t tunePathTile = 0;     // mode 200 work split: 0 = default (8), 8 / 16 = pixel tile edge per workgroup
    uint32_t tunePathPipeline = 0; // mode 200: 0 = one persistent kernel with wavefront-private queues (default: 21.2 vs 23.4 ms on C5), 1 = the stages as separate launches over global queues
    uint32_t tunePathPassPaths = 1u << 24; // wavefront pipeline: paths one pass may carry (112 bytes of queue / radiance memory each)
    uint32_t tunePathRanges = 8;   // mode 200: 8 = every XCD works through its own contiguous part of the frame first, 1 = one shared work counter
    // wave scheduling threshold of the closest-hit traversal loop (traversal.hip.h closestIteration; an int: > 0 = node steps while
    // that many lanes stand on inner nodes, -k = while k eighths of the wavefront's LIVE lanes do) and of the any-hit loop.
    // -6 against round 2's fixed 32: primary rays only 0.193 -> 0.172 ms, icosphere soup 0.256 -> 0.231, 5M triangles 0.367 -> 0.352,
    // C3 with shadow rays 0.2866 -> 0.2847, path tracing C5 22.6 -> 22.0 ms, lone launch of an 8-rank share 181 -> 174 us
    uint32_t tuneInnerMin = static_cast<uint32_t>(-6);
    uint32_t tuneInnerMinAny = static_cast<uint32_t>(-6);
    uint32_t tuneStackEntries = 0; // 0 = from the BVH depth
    uint32_t tuneXcdGroup = 16;
    uint32_t tuneBoostUnits = 512;
    // Split packets (split_packet.hip.h): the n most expensive 8x8 packets are rendered by 64 / split_rays wavefronts each whose
    // lanes share the segments of the block's rays.  kSplitAuto (option value -1, the default): none for a whole frame on one GPU
    // -- the launch order keeps the chip full there and splitting only adds work (C3: 0.299 -> 0.336 ms with 64 split) -- and 64 /
    // 128 packets for a tile share of 2 / >= 4 ranks, whose lone launch lasts as long as its slowest wavefront: 221 -> 188, 199 -> 132,
    // 178 -> 118 us at 2 / 4 / 8 ranks on the C3 frame (tools/split_probe.py).
    static constexpr uint32_t kSplitAuto = 0xFFFFFFFFu;
    uint32_t tuneSplitUnits = kSplitAuto;
    uint32_t tuneSplitRaysLog2 = 2; // option "split_rays": 4 rays per wavefront of a split packet (16 wavefronts per packet)
    uint32_t tuneSplitSegsLog2 = 4; // option "split_segments": 16 pieces per split ray
    bool tuneXcdAffine = false;
    uint32_t debugSkipUnits = 0;
    hipStream_t lastRenderStream = nullptr;
    bool haveLastRenderStream = false;
    // launch order from measured costs of an earlier frame: 0 never, 1 always, 2 (default) only for frames issued on the
    // same stream as the frame before.  Such frames run one after another, and starting the packets on the longest
    // critical paths first shortens each of them (0.52 -> 0.43 ms on the 1M-triangle frame); frames issued on
    // alternating streams overlap, the tail of one fills with the head of the next, so the order has nothing left to
    // gain and its bookkeeping only adds cross-stream dependencies (0.43 vs 0.40 ms per frame with 4 in flight).
    int adaptiveOrder = 2;
    // Per-frame scratch lives in a ring of kRing slots: frame f uses slot f % kRing and first waits (on the GPU, never on
    // the host) for the frame that used the slot before it, so up to kRing frames issued on different streams run
    // concurrently without sharing a spill arena or a cost/order buffer.
    static constexpr int kRing = 4;
    uint32_t* dUnitCost[kRing] = {};
    uint32_t* dUnitOrder[kRing] = {};
    uint32_t unitCapacity = 0;
    uint64_t orderKey[kRing] = {};   // frame geometry each stored order belongs to; 0 = none
    bool sortPending[kRing] = {};    // evSort[slot] recorded (a sort of this slot's costs was issued)
    uint32_t orderView[kRing] = {};  // viewSerial the stored order was measured under
    uint32_t orderGen[kRing] = {};   // consecutive measurements of this frame geometry
    uint32_t orderFrame[kRing] = {}; // frameSerial of the last measurement
    bool debugForceMeasure = false;  // diagnostics: measure and sort at every frame even for an unchanged view
    uint32_t tuneRemeasureEvery = 1; // a changing view re-measures at every use of a slot: stale orders cost more than the measuring (tools/moving_camera.py)
    uint32_t viewSerial = 1;         // bumped when camera, mode or path settings change: costs must be measured again
    bool renderPending[kRing] = {};  // evRender[slot] recorded
    hipStream_t slotStream[kRing] = {}; // stream the slot's last frame ran on
    uint32_t frameSerial = 0;
    hipStream_t sideStream = nullptr; // sorts the costs of frame f while later frames render
    hipEvent_t evRender[kRing] = {}, evSort[kRing] = {};
    unsigned long long* dCounters = nullptr;
    int* dSpill[kRing] = {};          // traversal-stack spill arenas (traversal.hip.h Stack), one per ring slot
    size_t spillBytes[kRing] = {};
    // mode 200: scratch of the persistent path kernel (one region per RESIDENT workgroup + the launch's work counter).  An arena
    // belongs to the stream that last used it: frames issued on one stream run one after the other and share ONE arena; only
    // frames on different streams (up to kRing in flight) get arenas of their own.
    struct PathArena {
        unsigned char* mem = nullptr;
        size_t bytes = 0;
        hipStream_t stream = nullptr;
        bool used = false;       // `stream` is meaningful
        hipEvent_t lastUse = nullptr;
        bool pending = false;    // lastUse recorded
        uint32_t serial = 0;     // frameSerial of the last use (least recently used arena is taken over by a new stream)
    } pathArena[kRing];
    unsigned long long* dTimeline = nullptr; // diagnostic: 3 words per workgroup, counting variant only
    size_t timelineWords = 0;
    bool wantTimeline = false;

    // native multi-GPU frame assembly (crt_comm_init): RCCL communicator + per-ring-slot staging / gathered / frame buffers
    ncclComm_t comm = nullptr;
    struct HostExchange* hostComm = nullptr; // crt_comm_init_host: the tiles travel through shared host memory instead of RCCL
    uint32_t commRank = 0, commRanks = 0;
    void* dStage[kRing] = {};
    void* dGather[kRing] = {};
    void* dDistFrame[kRing] = {};
    size_t stageBytes[kRing] = {}, gatherBytes[kRing] = {}, distFrameBytes[kRing] = {};
    uint32_t distSerial = 0;
    hipEvent_t evDist[kRing] = {};       // end of the frame that last used a slot's staging / gathered / frame buffers
    hipStream_t distStream[kRing] = {};  // and the stream it ran on
    bool distPending[kRing] = {};

    // scratch frame buffers for the host-output path, grown on demand
    void* dFrame[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t dFrameBytes[5] = { 0, 0, 0, 0, 0 };
};

namespace {

thread_local std::string g_createError;

int fail(crt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    else g_createError = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail((ctx), CRT_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

void freeScene(crt_ctx* c)
{
    void** ptrs[] = { &c->dNodes, &c->dBinNodes, &c->dWideNodes, &c->dTris, &c->dShade, &c->dLights, &c->dMats, &c->dUvs };
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->haveScene = false;
}

int ensureFrame(crt_ctx* c, int slot, size_t bytes)
{
    if (c->dFrameBytes[slot] >= bytes) return CRT_OK;
    if (c->dFrame[slot]) (void)hipFree(c->dFrame[slot]);
    c->dFrame[slot] = nullptr;
    c->dFrameBytes[slot] = 0;
    HIP_TRY(c, hipMalloc(&c->dFrame[slot], bytes));
    c->dFrameBytes[slot] = bytes;
    return CRT_OK;
}

uint32_t tilesFor(uint32_t w, uint32_t h) { return ((w + crt::kTile - 1) / crt::kTile) * ((h + crt::kTile - 1) / crt::kTile); }

void fillParams(const crt_ctx* c, uint32_t w, uint32_t h, uint32_t rank, uint32_t nRanks, RenderParams& p)
{
    std::memset(&p, 0, sizeof(p));
    p.nodes = c->dNodes;
    p.tris = c->bvh.width ? c->dNodes : c->dTris;
    p.shade = c->dShade;
    p.lights = c->dLights;
    p.mats = c->dMats;
    p.uvs = c->dUvs;
    p.textures = c->dTextures;
    p.texels = static_cast<const unsigned char*>(c->dTexels);
    p.n_textures = c->nTextures;
    p.layout = c->bvh.width;
    p.n_nodes = c->bvh.width ? c->bvh.nWide : c->bvh.nNodes4;
    p.n_tris = c->bvh.nTris;
    p.n_lights = c->nLights;
    p.n_mats = c->nMats;
    crt::copyBytes(p.pos, c->pos, sizeof(p.pos));
    crt::copyBytes(p.rot, c->rot, sizeof(p.rot));
    crt::copyBytes(p.miss, c->miss, sizeof(p.miss));
    crt::copyBytes(p.scene_lo, c->sceneLo, sizeof(p.scene_lo));
    crt::copyBytes(p.scene_hi, c->sceneHi, sizeof(p.scene_hi));
    p.mode = c->mode;
    p.spp = c->pathSpp;
    p.max_bounces = c->pathBounces;
    p.seed = c->pathSeed;
    p.phong_ks = static_cast<float>(c->phongKsPermille) / 1000.0f;
    p.phong_exp = c->phongExp;
    p.width = w;
    p.height = h;
    p.tiles_x = (w + crt::kTile - 1) / crt::kTile;
    p.tiles_y = (h + crt::kTile - 1) / crt::kTile;
    p.rank = rank;
    p.n_ranks = nRanks;
    const uint32_t nTiles = p.tiles_x * p.tiles_y;
    p.n_local_tiles = rank < nTiles ? (nTiles - rank + nRanks - 1) / nRanks : 0;
    p.counters = c->dCounters;
    p.timeline = nullptr;
    p.unit_order = nullptr;
    p.spill = nullptr;
    p.unit_cost = nullptr;
    p.tune_inner_min = c->tuneInnerMin;
    p.tune_inner_min_any = c->tuneInnerMinAny;
    p.xcd_group = c->tuneXcdGroup;
    p.boost_units = c->tuneBoostUnits;
    p.split_units = 0; // set in runRender once a launch order is in use
    p.split_rays_log2 = c->tuneSplitRaysLog2;
    p.split_segs_log2 = c->tuneSplitSegsLog2;
    p.debug_skip_units = c->debugSkipUnits;
    // LDS part of the per-lane stack: 16 entries x 64 lanes x 4 B = 4 KB per wavefront, so that LDS never limits the 7
    // wavefronts per SIMD the kernel's register budget allows (12 .. 20 entries measured alike, 24 costs 4 %); no ray of the
    // test scenes holds more than 15 entries, deeper ones (possible up to the builder's depth 32) spill to the arena
    p.stack_entries = c->tuneStackEntries ? c->tuneStackEntries : 16u;
    p.n_batch = 1;
    p.units_per_frame = crt::renderUnitCount(p);
}

// enqueue one frame; when stats != nullptr, bracket with events, synchronise and fill the timers/counters
int runRender(crt_ctx* c, RenderParams& p, crt_frame_stats* stats)
{
    const bool counting = c->counting;
    HIP_TRY(c, hipSetDevice(c->device)); // the calling thread's current device may be another one (scratch hipMallocs below)
    if (counting) HIP_TRY(c, hipMemsetAsync(c->dCounters, 0, 32 * sizeof(unsigned long long), c->stream));
    if (c->wantTimeline && p.n_batch == 1) {
        const size_t words = 3 * (static_cast<size_t>(p.tiles_x + 4) * (p.tiles_y + 4) * 4 + 1024);
        if (c->timelineWords < words) {
            if (c->dTimeline) (void)hipFree(c->dTimeline);
            c->dTimeline = nullptr;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->dTimeline), words * sizeof(unsigned long long)));
            c->timelineWords = words;
        }
        HIP_TRY(c, hipMemsetAsync(c->dTimeline, 0, c->timelineWords * sizeof(unsigned long long), c->stream));
        p.timeline = c->dTimeline;
    }
    const uint32_t slot = c->frameSerial++ % crt_ctx::kRing;
    {
        // deepest stack a ray can build: three pending siblings per wide level; what does not fit the LDS part spills
        const uint32_t deepest = c->bvh.width ? (c->bvh.width - 1u) * c->bvh.depthWide + 1u : 3u * c->bvh.depth4 + 1u;
        p.spill_stride = deepest > p.stack_entries ? deepest - p.stack_entries : 1u;
        // (mode 200: one slice per resident workgroup of the persistent kernel)
        if (p.mode >= 200u) {
            p.path_tile = c->tunePathTile ? c->tunePathTile : 8u;
            // the stages as separate launches over global queues (8x8 work items, 64-byte nodes), or one persistent kernel
            p.path_wavefront = (c->tunePathPipeline == 1u && p.path_tile == 8u && c->bvh.width == 0u) ? 1u : 0u;
        }
        // (+ 64 slices for each of the four wavefronts of a split packet)
        const size_t groups = p.mode >= 200u ? static_cast<size_t>(crt::pathGridSize(p))
                                             : (static_cast<size_t>(crt::renderUnitCount(p)) + 16u * std::min(c->tuneSplitUnits == crt_ctx::kSplitAuto ? 128u : c->tuneSplitUnits, crt::renderUnitCount(p) / 4u)) * p.n_batch;
        const size_t need = groups * 64u * p.spill_stride * sizeof(int);
        if (c->spillBytes[slot] < need) {
            HIP_TRY(c, hipDeviceSynchronize());
            if (c->dSpill[slot]) (void)hipFree(c->dSpill[slot]);
            c->dSpill[slot] = nullptr;
            c->spillBytes[slot] = 0;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->dSpill[slot]), need));
            c->spillBytes[slot] = need;
        }
        p.spill = c->dSpill[slot];
    }
    crt_ctx::PathArena* arena = nullptr;
    if (p.mode >= 200u) {
        // path tracing: resident wavefronts stream the paths of one pixel tile after the other through private queues in HBM.  One
        // 8x8 packet x up to 16 samples per work item (256 paths at 4 spp) measured best at every frame size -- 6.6 vs 10.1 ms on C3
        // at 1080p, 27.3 vs 29.2 ms on C5 at 4K against one 16x16 macro tile x 4 samples
        p.path_tile = c->tunePathTile ? c->tunePathTile : 8u;
        p.path_samples = std::min<uint32_t>(p.path_tile == 16u ? 4u : 16u, std::max<uint32_t>(1u, p.spp));
        p.path_region_bytes = crt::pathRegionBytes(p.path_tile, p.path_samples);
        p.path_work_items = crt::pathWorkgroupCount(p);
        const size_t kHead = 512; // the work counters (one 64-byte line per range) live in front of the regions
        p.path_ranges = c->tunePathRanges;
        uint32_t wfItems = 0;
        if (p.path_wavefront) {
            wfItems = crt::pathWavefrontPassItems(p, c->tunePathPassPaths);
            p.wf_paths = wfItems * 64u * p.path_samples;
            crt::pathWavefrontLayout(p, wfItems, p.wf_chunk, p.wf_stride);
        }
        const size_t need = p.path_wavefront ? crt::pathWavefrontBytes(p, wfItems) : kHead + static_cast<size_t>(crt::pathGridSize(p)) * p.path_region_bytes;
        // the arena this stream used last; else an unused one; else the least recently used one of another stream
        for (auto& a : c->pathArena)
            if (a.used && a.stream == c->stream) arena = &a;
        if (!arena) {
            for (auto& a : c->pathArena)
                if (!arena || (!a.used && arena->used) || (a.used == arena->used && a.serial < arena->serial)) arena = &a;
            if (arena->pending && hipEventQuery(arena->lastUse) != hipSuccess) HIP_TRY(c, hipStreamWaitEvent(c->stream, arena->lastUse, 0));
            arena->stream = c->stream;
            arena->used = true;
        }
        if (arena->bytes < need) {
            HIP_TRY(c, hipDeviceSynchronize());
            if (arena->mem) (void)hipFree(arena->mem);
            arena->mem = nullptr;
            arena->bytes = 0;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&arena->mem), need));
            arena->bytes = need;
        }
        if (!arena->lastUse) HIP_TRY(c, hipEventCreateWithFlags(&arena->lastUse, hipEventDisableTiming));
        arena->serial = c->frameSerial;
        if (p.path_wavefront) {
            const size_t plane = static_cast<size_t>(p.wf_paths) * 16u, qplane = static_cast<size_t>(p.wf_stride) * 16u; // one float4 per path / queue entry
            unsigned char* at = arena->mem;
            p.wf_counts = reinterpret_cast<uint32_t*>(at); at += crt::kWfHeadBytes;
            p.wf_shade_q = at; at += 3u * qplane;
            p.wf_trace_q = at; at += 2u * qplane;
            p.wf_done = at; at += plane;
            p.wf_thr = at; at += plane;
            p.wf_accum = p.spp > p.path_samples ? at : nullptr;
        } else {
            p.path_counter = reinterpret_cast<uint32_t*>(arena->mem);
            p.path_scratch = arena->mem + kHead;
            HIP_TRY(c, hipMemsetAsync(p.path_counter, 0, kHead, c->stream));
        }
    }
    // Cost feedback: the lifetimes frame f's wavefronts report are sorted on a side stream while the next frames render and
    // order the launch of frame f + kRing (same ring slot), so neither the sort nor the dependency on an earlier frame
    // sits on a frame's critical path and kRing frames can be in flight.  Hint only: a stale or missing order changes
    // speed, never results.
    const uint32_t nUnits = crt::renderUnitCount(p);
    const uint64_t key = (static_cast<uint64_t>(p.width) << 40) ^ (static_cast<uint64_t>(p.height) << 20) ^
                         (static_cast<uint64_t>(p.n_ranks) << 8) ^ p.rank ^ (static_cast<uint64_t>(c->sceneSerial) << 52) ^ 1ull;
    bool feedback = false;
    const bool sameStream = c->haveLastRenderStream && c->lastRenderStream == c->stream;
    c->lastRenderStream = c->stream;
    c->haveLastRenderStream = true;
    if ((c->adaptiveOrder == 1 || (c->adaptiveOrder == 2 && sameStream)) && nUnits && p.mode < 200u) { // (the path pipeline has its own work split)
        if (c->unitCapacity < nUnits) {
            HIP_TRY(c, hipDeviceSynchronize()); // (frames of other streams may still use the buffers about to be replaced)
            for (int i = 0; i < crt_ctx::kRing; i++) {
                if (c->dUnitCost[i]) (void)hipFree(c->dUnitCost[i]);
                if (c->dUnitOrder[i]) (void)hipFree(c->dUnitOrder[i]);
                c->dUnitCost[i] = c->dUnitOrder[i] = nullptr;
                c->orderKey[i] = 0;
                c->sortPending[i] = false;
            }
            c->unitCapacity = 0;
            for (int i = 0; i < crt_ctx::kRing; i++) {
                HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->dUnitCost[i]), sizeof(uint32_t) * nUnits));
                HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->dUnitOrder[i]), sizeof(uint32_t) * nUnits));
            }
            c->unitCapacity = nUnits;
        }
        const bool usable = c->orderKey[slot] == key;
        p.unit_order = usable ? c->dUnitOrder[slot] : nullptr;
        {
            uint32_t want = c->tuneSplitUnits;
            if (want == crt_ctx::kSplitAuto) want = p.n_ranks >= 4u ? 128u : (p.n_ranks >= 2u ? 64u : 0u);
            p.split_units = usable ? std::min(want, nUnits / 4u) : 0u; // (at most a quarter of the packets: the spill arena below is sized for that)
        }
        // Costs are measured (and sorted) twice in a row -- the first measurement ran under an unordered launch -- and then:
        // an unchanged view keeps its order for good; a view that keeps changing (a moving camera) measures again every
        // remeasure_every-th use of the slot.  Default 1: the order ages fast -- with a camera turning 0.01 degrees per frame
        // an order 32 frames old cost 0.367 ms per frame, 128 frames old 0.423, against 0.347 when measured every frame.
        // (The feedback itself -- cost stores, the sort beside the next frame -- is 1.4 % of a frame: tools/moving_camera.py --wobble;
        //  the rest of that run's difference to a static view was the frame's own cost changing along the orbit.)
        const bool twice = usable && c->orderGen[slot] >= 2;
        const bool sameView = c->orderView[slot] == c->viewSerial;
        const bool recent = (c->frameSerial - c->orderFrame[slot]) < static_cast<uint32_t>(crt_ctx::kRing) * c->tuneRemeasureEvery;
        if (c->debugForceMeasure || !(twice && (sameView || recent))) {
            p.unit_cost = c->dUnitCost[slot];
            feedback = true;
            c->orderGen[slot] = usable ? c->orderGen[slot] + 1 : 1;
            c->orderView[slot] = c->viewSerial;
            c->orderFrame[slot] = c->frameSerial;
        }
    }
    // the previous user of this slot (frame f - kRing, possibly on another stream) and the sort of its costs must be done
    // before this frame touches the slot's spill arena, cost or order buffer
    // (a wait is only enqueued when it can matter: not for an event that has already completed, not for a frame that ran on
    // this same stream -- every barrier packet costs the stream a few microseconds)
    if (c->sortPending[slot] && hipEventQuery(c->evSort[slot]) != hipSuccess) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evSort[slot], 0));
    if (c->renderPending[slot] && c->slotStream[slot] != c->stream && hipEventQuery(c->evRender[slot]) != hipSuccess)
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evRender[slot], 0));
    c->sortPending[slot] = false;
    c->slotStream[slot] = c->stream;
    // split packets add their quarters' lifetimes into their cost slot: start from zero
    if (p.unit_cost && p.split_units) HIP_TRY(c, hipMemsetAsync(p.unit_cost, 0, sizeof(uint32_t) * nUnits, c->stream));
    if (stats) HIP_TRY(c, hipEventRecord(c->evStart, c->stream));
    const int rc = crt::launchRender(p, counting, c->stream);
    if (rc != 0) return fail(c, CRT_EHIP, "render kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
    if (stats) HIP_TRY(c, hipEventRecord(c->evStop, c->stream)); // kernel_ms = the render kernel alone
    HIP_TRY(c, hipEventRecord(c->evRender[slot], c->stream));
    c->renderPending[slot] = true;
    if (arena) {
        HIP_TRY(c, hipEventRecord(arena->lastUse, c->stream));
        arena->pending = true;
    }
    if (feedback) {
        hipStream_t ss = c->sideStream;
        HIP_TRY(c, hipStreamWaitEvent(ss, c->evRender[slot], 0));
        const int rs = crt::launchSortUnits(c->dUnitCost[slot], c->dUnitOrder[slot], nUnits, c->tuneXcdAffine, ss);
        if (rs != 0) return fail(c, CRT_EHIP, "sort kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rs)));
        HIP_TRY(c, hipEventRecord(c->evSort[slot], ss));
        c->sortPending[slot] = true;
        c->orderKey[slot] = key;
    }
    if (stats) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->evStart, c->evStop));
        std::memset(stats, 0, sizeof(*stats));
        stats->kernel_ms = ms;
        stats->rays_primary = 0;
        // pixels rendered by this launch
        uint64_t pix = 0;
        const uint32_t nTiles = p.tiles_x * p.tiles_y;
        for (uint32_t k = p.rank; k < nTiles; k += p.n_ranks) {
            const uint32_t tx = k % p.tiles_x, ty = k / p.tiles_x;
            const uint32_t w = std::min<uint32_t>(crt::kTile, p.width - tx * crt::kTile);
            const uint32_t h = std::min<uint32_t>(crt::kTile, p.height - ty * crt::kTile);
            pix += static_cast<uint64_t>(w) * h;
        }
        stats->rays_primary = pix;
        if (counting) {
            unsigned long long host[4] = { 0, 0, 0, 0 };
            HIP_TRY(c, hipMemcpy(host, c->dCounters, sizeof(host), hipMemcpyDeviceToHost));
            stats->nodes_visited = host[0];
            stats->tris_tested = host[1];
            stats->rays_shadow = host[2];
            stats->rays_primary = host[3]; // closest-hit rays: camera rays, plus bounce rays in mode 200
        }
    }
    return CRT_OK;
}

int checkRenderable(crt_ctx* c, uint32_t w, uint32_t h)
{
    if (!c) return CRT_EINVAL;
    if (!c->haveScene) return fail(c, CRT_ESTATE, "no scene uploaded: call crt_upload_scene first");
    if (w == 0 || h == 0 || w > 65536 || h > 65536) return fail(c, CRT_EINVAL, "bad frame size %ux%u", w, h);
    return CRT_OK;
}

} // namespace

extern "C" {

uint32_t crt_abi_version(void) { return CRT_ABI_VERSION; }

namespace {
// the 1-workgroup sort that orders a later frame runs beside the next frame's render kernel: highest priority, so it is
// dispatched at once instead of queueing behind 32 640 render workgroups
hipError_t createSideStream(crt_ctx* c)
{
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(&c->sideStream, hipStreamNonBlocking, greatest);
}
hipError_t createRingEvents(crt_ctx* c)
{
    for (int i = 0; i < crt_ctx::kRing; i++) {
        hipError_t e = hipEventCreateWithFlags(&c->evRender[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->evSort[i], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
} // namespace

int crt_create(crt_ctx** out, int device_id)
{
    if (!out) return fail(nullptr, CRT_EINVAL, "crt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, CRT_ENODEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= n) return fail(nullptr, CRT_EINVAL, "device_id %d out of range [0,%d)", device_id, n);
    crt_ctx* c = new (std::nothrow) crt_ctx();
    if (!c) return fail(nullptr, CRT_ENOMEM, "out of host memory");
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->evStart)) != hipSuccess || (e = hipEventCreate(&c->evStop)) != hipSuccess ||
        (e = createSideStream(c)) != hipSuccess ||
        (e = createRingEvents(c)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&c->dCounters), 32 * sizeof(unsigned long long))) != hipSuccess) {
        const int rc = fail(nullptr, CRT_ENODEVICE, "HIP initialisation failed on device %d: %s", device_id, hipGetErrorString(e));
        crt_destroy(c);
        return rc;
    }
    c->stream = c->ownStream;
    *out = c;
    return CRT_OK;
}

void crt_destroy(crt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize(); // frames may still be in flight on streams the caller set earlier, not only on the current one
    freeScene(c);
    for (int i = 0; i < 5; i++)
        if (c->dFrame[i]) (void)hipFree(c->dFrame[i]);
    (void)crt_comm_destroy(c);
    for (int i = 0; i < crt_ctx::kRing; i++) {
        if (c->dStage[i]) (void)hipFree(c->dStage[i]);
        if (c->dGather[i]) (void)hipFree(c->dGather[i]);
        if (c->dDistFrame[i]) (void)hipFree(c->dDistFrame[i]);
        if (c->evDist[i]) (void)hipEventDestroy(c->evDist[i]);
    }
    if (c->dCounters) (void)hipFree(c->dCounters);
    if (c->dTextures) (void)hipFree(c->dTextures);
    if (c->dTexels) (void)hipFree(c->dTexels);
    for (int i = 0; i < crt_ctx::kRing; i++) {
        if (c->dSpill[i]) (void)hipFree(c->dSpill[i]);
        if (c->pathArena[i].mem) (void)hipFree(c->pathArena[i].mem);
        if (c->pathArena[i].lastUse) (void)hipEventDestroy(c->pathArena[i].lastUse);
    }
    if (c->sideStream) (void)hipStreamSynchronize(c->sideStream);
    for (int i = 0; i < crt_ctx::kRing; i++) {
        if (c->dUnitCost[i]) (void)hipFree(c->dUnitCost[i]);
        if (c->dUnitOrder[i]) (void)hipFree(c->dUnitOrder[i]);
        if (c->evRender[i]) (void)hipEventDestroy(c->evRender[i]);
        if (c->evSort[i]) (void)hipEventDestroy(c->evSort[i]);
    }
    if (c->sideStream) (void)hipStreamDestroy(c->sideStream);
    if (c->dTimeline) (void)hipFree(c->dTimeline);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

const char* crt_last_error(const crt_ctx* c) { return c ? c->error.c_str() : g_createError.c_str(); }

int crt_bvh_build_host(const crt_mesh_view* meshes, uint32_t n_meshes, crt_bvh_node** nodes, uint32_t* n_nodes,
                       crt_bvh_tri** tris, crt_bvh_shade** shade, uint32_t* n_tris, uint32_t* max_depth)
{
    if ((!meshes && n_meshes) || !nodes || !n_nodes || !tris || !n_tris) return fail(nullptr, CRT_EINVAL, "crt_bvh_build_host: NULL argument");
    try {
        crt::Bvh b;
        crt::buildBvh(meshes, n_meshes, b);
        *n_nodes = static_cast<uint32_t>(b.nodes.size());
        *n_tris = static_cast<uint32_t>(b.tris.size());
        if (max_depth) *max_depth = b.maxDepth;
        *nodes = static_cast<crt_bvh_node*>(std::malloc(sizeof(crt_bvh_node) * (b.nodes.size() + 1)));
        *tris = static_cast<crt_bvh_tri*>(std::malloc(sizeof(crt_bvh_tri) * (b.tris.size() + 1)));
        if (shade) *shade = static_cast<crt_bvh_shade*>(std::malloc(sizeof(crt_bvh_shade) * (b.tris.size() + 1)));
        if (!*nodes || !*tris || (shade && !*shade)) return fail(nullptr, CRT_ENOMEM, "out of host memory");
        crt::copyBytes(*nodes, b.nodes.data(), sizeof(crt_bvh_node) * b.nodes.size());
        crt::copyBytes(*tris, b.tris.data(), sizeof(crt_bvh_tri) * b.tris.size());
        if (shade) crt::copyBytes(*shade, b.shade.data(), sizeof(crt_bvh_shade) * b.shade.size());
    } catch (const std::exception& ex) {
        return fail(nullptr, CRT_EINVAL, "BVH build failed: %s", ex.what());
    }
    return CRT_OK;
}

void crt_free(void* p) { std::free(p); }

void* crt_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void crt_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int crt_upload_scene(crt_ctx* c, const crt_mesh_view* meshes, uint32_t n_meshes, const crt_light* lights, uint32_t n_lights,
                     const crt_material* materials, uint32_t n_materials)
{
    if (!c) return CRT_EINVAL;
    if ((!meshes && n_meshes) || (!lights && n_lights) || (!materials && n_materials)) return fail(c, CRT_EINVAL, "NULL array with non-zero count");
    const auto tb0 = std::chrono::steady_clock::now();
    double deviceMs = 0.0;
    crt::Bvh built; // swapped into the context only once the build has succeeded: a failed build leaves the old scene intact
    try {
        if (c->gpuBuild) {
            HIP_TRY(c, hipSetDevice(c->device));
            crt::buildBvhGpu(meshes, n_meshes, built, c->stream, &deviceMs);
        } else {
            crt::buildBvh(meshes, n_meshes, built, static_cast<int>(c->bvhWidth));
        }
    } catch (const std::bad_alloc&) {
        return fail(c, CRT_ENOMEM, "out of host memory while building the BVH");
    } catch (const std::exception& ex) {
        return fail(c, CRT_EINVAL, "BVH build failed: %s", ex.what());
    }
    // records the GPU builder left in HBM: adopted below, or released here if anything fails before that
    struct DevRecords {
        crt::Bvh& b;
        ~DevRecords()
        {
            for (void** q : { &b.devTris, &b.devShade, &b.devUvs, &b.devNodes, &b.devNodes4, &b.devNodes4q }) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
        }
    } pending{ built };
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize()); // frames in flight on any stream the caller used still read the old scene's buffers
    freeScene(c); // from here on a failure leaves NO scene (haveScene = false): never the new host tree over old device buffers
    std::swap(c->bvh, built); // (pending now guards whatever the OLD tree still pointed at: nothing)
    c->buildDeviceMs = deviceMs;
    const bool recordsOnDevice = c->bvh.devTris != nullptr;
    if (recordsOnDevice) { // built on the GPU: the leaf-ordered records are already in HBM; adopted before anything else can fail
        c->dTris = c->bvh.devTris;
        c->dShade = c->bvh.devShade;
        c->dUvs = c->bvh.devUvs;
        c->dNodes = c->bvh.devNodes4q;
        c->dBinNodes = c->bvh.devNodes;
        c->dWideNodes = c->bvh.devNodes4;
        c->bvh.devTris = c->bvh.devShade = c->bvh.devUvs = c->bvh.devNodes = c->bvh.devNodes4 = c->bvh.devNodes4q = nullptr;
    }
    const size_t nb = sizeof(crt_bvh_node4q) * c->bvh.nodes4q.size(); // the quantised wide tree is what the kernels traverse
    const size_t tb = sizeof(crt_bvh_tri) * c->bvh.nTris;
    const size_t sb = sizeof(crt_bvh_shade) * c->bvh.nTris;
    const bool packed = c->bvh.width != 0;
    // +64 bytes of slack so that a speculative wide load of the last record stays inside the allocation
    if (packed && !c->dNodes) { // nodes and triangles in one buffer: dTris stays empty, the kernels get dNodes for both
        const size_t pb = c->bvh.packed.size() * sizeof(uint32_t);
        HIP_TRY(c, hipMalloc(&c->dNodes, pb + 128));
        if (pb) HIP_TRY(c, hipMemcpy(c->dNodes, c->bvh.packed.data(), pb, hipMemcpyHostToDevice));
    }
    if (!c->dNodes) { // (a tree collapsed on the device is already there)
        HIP_TRY(c, hipMalloc(&c->dNodes, nb + 128));
        if (nb) HIP_TRY(c, hipMemcpy(c->dNodes, c->bvh.nodes4q.data(), nb, hipMemcpyHostToDevice));
    }
    if (!recordsOnDevice) {
        if (!packed) HIP_TRY(c, hipMalloc(&c->dTris, tb + 64));
        HIP_TRY(c, hipMalloc(&c->dShade, sb + 64));
        if (tb && !packed) HIP_TRY(c, hipMemcpy(c->dTris, c->bvh.tris.data(), tb, hipMemcpyHostToDevice));
        if (sb) HIP_TRY(c, hipMemcpy(c->dShade, c->bvh.shade.data(), sb, hipMemcpyHostToDevice));
        if (!c->bvh.uvs.empty()) {
            const size_t ub = sizeof(crt_bvh_uv) * c->bvh.uvs.size();
            HIP_TRY(c, hipMalloc(&c->dUvs, ub));
            HIP_TRY(c, hipMemcpy(c->dUvs, c->bvh.uvs.data(), ub, hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(c, hipMalloc(&c->dLights, sizeof(crt_light) * (n_lights + 1)));
    HIP_TRY(c, hipMalloc(&c->dMats, sizeof(crt_material) * (n_materials + 1)));
    if (n_lights) HIP_TRY(c, hipMemcpy(c->dLights, lights, sizeof(crt_light) * n_lights, hipMemcpyHostToDevice));
    if (n_materials) HIP_TRY(c, hipMemcpy(c->dMats, materials, sizeof(crt_material) * n_materials, hipMemcpyHostToDevice));
    c->nLights = n_lights;
    c->nMats = n_materials;
    for (int a = 0; a < 3; a++) c->sceneLo[a] = c->sceneHi[a] = 0.0f;
    if (c->bvh.width == 0 && c->bvh.nNodes4 > 0) { // the root record as it sits in HBM (built here or on the device alike)
        crt_bvh_node4q root;
        HIP_TRY(c, hipMemcpy(&root, c->dNodes, sizeof(root), hipMemcpyDeviceToHost));
        for (int a = 0; a < 3; a++) {
            c->sceneLo[a] = root.lo[a];
            c->sceneHi[a] = crt::decodePlane(255u, root.s[a], root.lo[a]);
        }
    }
    c->buildMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
    c->haveScene = true;
    c->sceneSerial++;
    for (uint64_t& k : c->orderKey) k = 0;
    return CRT_OK;
}

int crt_set_textures(crt_ctx* c, const crt_texture* textures, uint32_t n)
{
    if (!c) return CRT_EINVAL;
    if (!textures && n) return fail(c, CRT_EINVAL, "crt_set_textures: NULL array with non-zero count");
    std::vector<crt::TextureRec> recs(n);
    std::vector<unsigned char> pool;
    for (uint32_t i = 0; i < n; i++) {
        const crt_texture& t = textures[i];
        if (t.type > 3u) return fail(c, CRT_EINVAL, "texture %u: unknown type %u", i, t.type);
        crt::TextureRec& r = recs[i];
        r.type = t.type;
        crt::copyBytes(r.a, t.color_a, 12);
        crt::copyBytes(r.b, t.color_b, 12);
        r.scalar = t.scalar;
        r.texel_offset = r.width = r.height = r.channels = 0;
        if (t.type == 3u) {
            if (!t.pixels || t.width == 0 || t.height == 0 || t.channels < 3)
                return fail(c, CRT_EINVAL, "texture %u: bitmap needs pixels, width, height and >= 3 channels", i);
            const size_t bytes = static_cast<size_t>(t.width) * t.height * t.channels;
            r.texel_offset = static_cast<uint32_t>(pool.size());
            r.width = t.width; r.height = t.height; r.channels = t.channels;
            pool.insert(pool.end(), t.pixels, t.pixels + bytes);
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->dTextures) (void)hipFree(c->dTextures);
    if (c->dTexels) (void)hipFree(c->dTexels);
    c->dTextures = c->dTexels = nullptr;
    c->nTextures = 0;
    HIP_TRY(c, hipMalloc(&c->dTextures, sizeof(crt::TextureRec) * (n + 1)));
    HIP_TRY(c, hipMalloc(&c->dTexels, pool.size() + 16));
    if (n) HIP_TRY(c, hipMemcpy(c->dTextures, recs.data(), sizeof(crt::TextureRec) * n, hipMemcpyHostToDevice));
    if (!pool.empty()) HIP_TRY(c, hipMemcpy(c->dTexels, pool.data(), pool.size(), hipMemcpyHostToDevice));
    c->nTextures = n;
    return CRT_OK;
}

int crt_bvh_export_uv(const crt_ctx* c, crt_bvh_uv* uvs, int* has_uvs)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (has_uvs) *has_uvs = c->dUvs ? 1 : 0;
    if (uvs && c->dUvs) { // (the host copy exists only for trees built on the host)
        if (!c->bvh.uvs.empty()) crt::copyBytes(uvs, c->bvh.uvs.data(), sizeof(crt_bvh_uv) * c->bvh.uvs.size());
        else if (hipMemcpy(uvs, c->dUvs, sizeof(crt_bvh_uv) * c->bvh.nTris, hipMemcpyDeviceToHost) != hipSuccess) return CRT_EHIP;
    }
    return CRT_OK;
}

int crt_set_camera(crt_ctx* c, const float pos[3], const float rot[9])
{
    if (!c) return CRT_EINVAL;
    if (!pos || !rot) return fail(c, CRT_EINVAL, "crt_set_camera: NULL argument");
    if (std::memcmp(c->pos, pos, sizeof(c->pos)) != 0 || std::memcmp(c->rot, rot, sizeof(c->rot)) != 0) c->viewSerial++;
    crt::copyBytes(c->pos, pos, sizeof(c->pos));
    crt::copyBytes(c->rot, rot, sizeof(c->rot));
    return CRT_OK;
}

int crt_set_shading_mode(crt_ctx* c, uint32_t mode)
{
    if (!c) return CRT_EINVAL;
    if (c->mode != mode) c->viewSerial++;
    c->mode = mode;
    return CRT_OK;
}

int crt_set_miss_color(crt_ctx* c, const float rgb[3])
{
    if (!c) return CRT_EINVAL;
    if (!rgb) return fail(c, CRT_EINVAL, "crt_set_miss_color: NULL argument");
    crt::copyBytes(c->miss, rgb, sizeof(c->miss));
    return CRT_OK;
}

int crt_set_counting(crt_ctx* c, int enabled)
{
    if (!c) return CRT_EINVAL;
    c->counting = enabled != 0;
    return CRT_OK;
}

int crt_set_option(crt_ctx* c, const char* name, int value)
{
    if (!c || !name) return CRT_EINVAL;
    if (std::strcmp(name, "gpu_build") == 0) {
        c->gpuBuild = value != 0;
        return CRT_OK;
    }
#if defined(CRT_PACKED_LAYOUTS) && CRT_PACKED_LAYOUTS
    // round-3 experiment (tools/variant_build.sh packed "-DCRT_PACKED_LAYOUTS=1", tools/width_ab.py): 0 = the product's 64-byte
    // 4-wide nodes, 4 / 8 = packed wide tree (bvh_pack.h), taken at the next crt_upload_scene; host build only
    if (std::strcmp(name, "bvh_width") == 0 && (value == 0 || value == 4 || value == 8)) {
        c->bvhWidth = static_cast<uint32_t>(value);
        return CRT_OK;
    }
#else
    if (std::strcmp(name, "bvh_width") == 0 && value == 0) return CRT_OK;
#endif
    if (std::strcmp(name, "spp") == 0 && value >= 1 && value <= 65536) {
        c->pathSpp = static_cast<uint32_t>(value);
        c->viewSerial++;
        return CRT_OK;
    }
    if (std::strcmp(name, "max_bounces") == 0 && value >= 0 && value <= 64) {
        c->pathBounces = static_cast<uint32_t>(value);
        c->viewSerial++;
        return CRT_OK;
    }
    if (std::strcmp(name, "phong_ks") == 0 && value >= 0 && value <= 100000) {
        c->phongKsPermille = static_cast<uint32_t>(value);
        c->viewSerial++;
        return CRT_OK;
    }
    if (std::strcmp(name, "phong_exponent") == 0 && value >= 1 && value <= 65536) {
        c->phongExp = static_cast<uint32_t>(value);
        c->viewSerial++;
        return CRT_OK;
    }
    if (std::strcmp(name, "seed") == 0) {
        c->pathSeed = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "path_pipeline") == 0 && (value == 0 || value == 1)) {
        c->tunePathPipeline = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "path_pass_paths") == 0 && value >= (1 << 16) && value <= (1 << 25)) {
        c->tunePathPassPaths = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "path_ranges") == 0 && (value == 1 || value == 8)) {
        c->tunePathRanges = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "path_tile") == 0 && (value == 0 || value == 8 || value == 16)) {
        c->tunePathTile = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "inner_min") == 0 && value >= -8 && value <= 65 && value != 0) { // negative: adaptive, eighths of the live lanes
        c->tuneInnerMin = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "inner_min_any") == 0 && value >= -8 && value <= 65 && value != 0) {
        c->tuneInnerMinAny = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "xcd_group") == 0 && (value == 1 || value == 2 || value == 4 || value == 8 || value == 16)) {
        c->tuneXcdGroup = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "xcd_affine_order") == 0 && (value == 0 || value == 1)) {
        c->tuneXcdAffine = value != 0;
        for (int i = 0; i < crt_ctx::kRing; i++) c->orderKey[i] = 0; // orders sorted the other way are stale
        return CRT_OK;
    }
    if (std::strcmp(name, "split_units") == 0 && value >= -1 && value <= 65536) {
        c->tuneSplitUnits = value < 0 ? crt_ctx::kSplitAuto : static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "split_rays") == 0 && (value == 4 || value == 8 || value == 16)) {
        c->tuneSplitRaysLog2 = value == 4 ? 2u : (value == 8 ? 3u : 4u);
        return CRT_OK;
    }
    if (std::strcmp(name, "split_segments") == 0 && (value == 4 || value == 8 || value == 16)) {
        c->tuneSplitSegsLog2 = value == 4 ? 2u : (value == 8 ? 3u : 4u);
        return CRT_OK;
    }
    if (std::strcmp(name, "boost_units") == 0 && value >= 0) {
        c->tuneBoostUnits = static_cast<uint32_t>(value);
        return CRT_OK;
    }
#if CRT_DIAG
    if (std::strcmp(name, "debug_skip_units") == 0 && value >= 0) {
        c->debugSkipUnits = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "timeline") == 0) {
        c->wantTimeline = value != 0;
        return CRT_OK;
    }
    if (std::strcmp(name, "debug_force_measure") == 0) {
        c->debugForceMeasure = value != 0;
        return CRT_OK;
    }
#else
    if (std::strcmp(name, "debug_skip_units") == 0 || std::strcmp(name, "timeline") == 0 || std::strcmp(name, "debug_force_measure") == 0)
        return fail(c, CRT_EINVAL, "option '%s' exists only in the diagnostic build (tools/diag_build.sh)", name);
#endif
    if (std::strcmp(name, "remeasure_every") == 0 && value >= 1 && value <= 1024) {
        c->tuneRemeasureEvery = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    if (std::strcmp(name, "adaptive_order") == 0) {
        if (value < 0 || value > 2) return fail(c, CRT_EINVAL, "adaptive_order takes 0 (off), 1 (on) or 2 (auto)");
        c->adaptiveOrder = static_cast<int>(value);
        for (uint64_t& k : c->orderKey) k = 0;
        return CRT_OK;
    }
    if (std::strcmp(name, "stack_entries") == 0 && (value == 0 || (value >= 1 && value <= crt::kStackEntries))) {
        c->tuneStackEntries = static_cast<uint32_t>(value);
        return CRT_OK;
    }
    return fail(c, CRT_EINVAL, "unknown option '%s' or value %d out of range", name, value);
}

int crt_debug_read_timeline(crt_ctx* c, unsigned long long* out, size_t max_words, size_t* n_words)
{
    if (!c || !out || !n_words) return CRT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = c->timelineWords < max_words ? c->timelineWords : max_words;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n) HIP_TRY(c, hipMemcpy(out, c->dTimeline, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *n_words = n;
    return CRT_OK;
}

int crt_debug_read_counters(crt_ctx* c, unsigned long long out[32])
{
    if (!c || !out) return CRT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->dCounters, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_set_stream(crt_ctx* c, void* hip_stream)
{
    if (!c) return CRT_EINVAL;
    c->stream = static_cast<hipStream_t>(hip_stream);
    return CRT_OK;
}

int crt_reset_stream(crt_ctx* c)
{
    if (!c) return CRT_EINVAL;
    c->stream = c->ownStream;
    return CRT_OK;
}

int crt_synchronize(crt_ctx* c)
{
    if (!c) return CRT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return CRT_OK;
}

int crt_render_frame_device(crt_ctx* c, uint32_t w, uint32_t h, void* d_rgba8, void* d_hit_inst, void* d_hit_prim,
                            void* d_hit_t, void* d_rgb_f32, crt_frame_stats* stats)
{
    int rc = checkRenderable(c, w, h);
    if (rc) return rc;
    if (!d_rgba8) return fail(c, CRT_EINVAL, "d_rgba8 is NULL");
    const auto t0 = std::chrono::steady_clock::now();
    RenderParams p;
    fillParams(c, w, h, 0, 1, p);
    p.rgba8 = static_cast<uint32_t*>(d_rgba8);
    p.hit_inst = static_cast<uint32_t*>(d_hit_inst);
    p.hit_prim = static_cast<uint32_t*>(d_hit_prim);
    p.hit_t = static_cast<float*>(d_hit_t);
    p.rgb_f32 = static_cast<float*>(d_rgb_f32);
    rc = runRender(c, p, stats);
    if (rc) return rc;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
}

int crt_render_frame(crt_ctx* c, uint32_t w, uint32_t h, uint8_t* rgba8, uint32_t* hit_inst, uint32_t* hit_prim, float* hit_t,
                     float* rgb_f32, crt_frame_stats* stats)
{
    int rc = checkRenderable(c, w, h);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = static_cast<size_t>(w) * h;
    void* host[5] = { rgba8, hit_inst, hit_prim, hit_t, rgb_f32 };
    const size_t bytes[5] = { n * 4, n * 4, n * 4, n * 4, n * 12 };
    for (int i = 0; i < 5; i++)
        if (host[i] || i == 0)
            if ((rc = ensureFrame(c, i, bytes[i])) != CRT_OK) return rc;
    RenderParams p;
    fillParams(c, w, h, 0, 1, p);
    p.rgba8 = static_cast<uint32_t*>(c->dFrame[0]);
    p.hit_inst = host[1] ? static_cast<uint32_t*>(c->dFrame[1]) : nullptr;
    p.hit_prim = host[2] ? static_cast<uint32_t*>(c->dFrame[2]) : nullptr;
    p.hit_t = host[3] ? static_cast<float*>(c->dFrame[3]) : nullptr;
    p.rgb_f32 = host[4] ? static_cast<float*>(c->dFrame[4]) : nullptr;
    crt_frame_stats local;
    rc = runRender(c, p, stats ? stats : &local); // synchronous like the reference's renderFrame
    if (rc) return rc;
    for (int i = 0; i < 5; i++)
        if (host[i]) HIP_TRY(c, hipMemcpyAsync(host[i], c->dFrame[i], bytes[i], hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
}

uint32_t crt_tile_count(uint32_t w, uint32_t h) { return tilesFor(w, h); }

uint32_t crt_tile_slots(uint32_t w, uint32_t h, uint32_t n_ranks)
{
    if (n_ranks == 0) return 0;
    return (tilesFor(w, h) + n_ranks - 1) / n_ranks;
}

int crt_render_tiles_device(crt_ctx* c, uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, void* d_staging, crt_frame_stats* stats)
{
    int rc = checkRenderable(c, w, h);
    if (rc) return rc;
    if (!d_staging || n_ranks == 0 || rank >= n_ranks) return fail(c, CRT_EINVAL, "bad tile arguments (rank %u of %u)", rank, n_ranks);
    const auto t0 = std::chrono::steady_clock::now();
    RenderParams p;
    fillParams(c, w, h, rank, n_ranks, p);
    p.staging = 1;
    p.rgba8 = static_cast<uint32_t*>(d_staging);
    rc = runRender(c, p, stats);
    if (rc) return rc;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
}

namespace {
// shared by the two batch entry points: frame 0 takes the place of the single frame, the others go to the batch arrays
int runBatch(crt_ctx* c, uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, bool staging, uint32_t n_frames, const float* cameras,
             void* const* d_out, crt_frame_stats* stats)
{
    int rc = checkRenderable(c, w, h);
    if (rc) return rc;
    if (n_frames == 0 || n_frames > static_cast<uint32_t>(crt::kMaxBatch)) return fail(c, CRT_EINVAL, "n_frames %u outside [1,%d]", n_frames, crt::kMaxBatch);
    if (!d_out || n_ranks == 0 || rank >= n_ranks) return fail(c, CRT_EINVAL, "bad batch arguments (rank %u of %u)", rank, n_ranks);
    for (uint32_t f = 0; f < n_frames; f++)
        if (!d_out[f]) return fail(c, CRT_EINVAL, "output pointer of frame %u is NULL", f);
    const auto t0 = std::chrono::steady_clock::now();
    RenderParams p;
    fillParams(c, w, h, rank, n_ranks, p);
    p.staging = staging ? 1u : 0u;
    p.n_batch = n_frames;
    p.rgba8 = static_cast<uint32_t*>(d_out[0]);
    if (cameras) {
        crt::copyBytes(p.pos, cameras, sizeof(p.pos));
        crt::copyBytes(p.rot, cameras + 3, sizeof(p.rot));
    }
    for (uint32_t f = 1; f < n_frames; f++) {
        const float* cam = cameras ? cameras + 12 * f : nullptr;
        crt::copyBytes(p.batch_pos[f - 1], cam ? cam : c->pos, sizeof(p.pos));
        crt::copyBytes(p.batch_rot[f - 1], cam ? cam + 3 : c->rot, sizeof(p.rot));
        p.batch_rgba8[f - 1] = static_cast<uint32_t*>(d_out[f]);
    }
    rc = runRender(c, p, stats);
    if (rc) return rc;
    if (stats) stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return CRT_OK;
}
} // namespace

int crt_render_frames_batch_device(crt_ctx* c, uint32_t w, uint32_t h, uint32_t n_frames, const float* cameras, void* const* d_rgba8,
                                   crt_frame_stats* stats)
{
    return runBatch(c, w, h, 0, 1, false, n_frames, cameras, d_rgba8, stats);
}

int crt_render_tiles_batch_device(crt_ctx* c, uint32_t w, uint32_t h, uint32_t rank, uint32_t n_ranks, uint32_t n_frames,
                                  const float* cameras, void* const* d_staging, crt_frame_stats* stats)
{
    return runBatch(c, w, h, rank, n_ranks, true, n_frames, cameras, d_staging, stats);
}

int crt_untile_batch_device(crt_ctx* c, uint32_t w, uint32_t h, uint32_t n_ranks, uint32_t n_frames, uint32_t frame, const void* d_gathered,
                            void* d_rgba8)
{
    if (!c) return CRT_EINVAL;
    if (!d_gathered || !d_rgba8 || n_ranks == 0 || w == 0 || h == 0 || n_frames == 0 || frame >= n_frames)
        return fail(c, CRT_EINVAL, "crt_untile_batch_device: bad argument");
    const uint32_t slots = crt_tile_slots(w, h, n_ranks);
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = crt::launchUntile(static_cast<const uint32_t*>(d_gathered), static_cast<uint32_t*>(d_rgba8), w, h, n_ranks,
                                     n_frames * slots, frame * slots, c->stream);
    if (rc != 0) return fail(c, CRT_EHIP, "untile kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
    return CRT_OK;
}

int crt_untile_device(crt_ctx* c, uint32_t w, uint32_t h, uint32_t n_ranks, const void* d_gathered, void* d_rgba8)
{
    return crt_untile_batch_device(c, w, h, n_ranks, 1, 0, d_gathered, d_rgba8);
}

int crt_bvh_info(const crt_ctx* c, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* max_depth)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (n_nodes) *n_nodes = c->bvh.nNodes;
    if (n_tris) *n_tris = c->bvh.nTris;
    if (max_depth) *max_depth = c->bvh.maxDepth;
    return CRT_OK;
}

int crt_build_stats(const crt_ctx* c, double* upload_ms, double* device_build_ms)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (upload_ms) *upload_ms = c->buildMs;
    if (device_build_ms) *device_build_ms = c->buildDeviceMs;
    return CRT_OK;
}

int crt_bvh_info4(const crt_ctx* c, uint32_t* n_nodes4, uint32_t* depth4)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (n_nodes4) *n_nodes4 = c->bvh.nNodes4;
    if (depth4) *depth4 = c->bvh.depth4;
    return CRT_OK;
}

int crt_bvh_export4(const crt_ctx* c, crt_bvh_node4* nodes4)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (nodes4) {
        if (!c->dWideNodes) crt::copyBytes(nodes4, c->bvh.nodes4.data(), sizeof(crt_bvh_node4) * c->bvh.nodes4.size());
        else if (hipMemcpy(nodes4, c->dWideNodes, sizeof(crt_bvh_node4) * c->bvh.nNodes4, hipMemcpyDeviceToHost) != hipSuccess) return CRT_EHIP;
    }
    return CRT_OK;
}

int crt_bvh_export4q(const crt_ctx* c, crt_bvh_node4q* nodes4q)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (nodes4q) {
        if (!c->dWideNodes) crt::copyBytes(nodes4q, c->bvh.nodes4q.data(), sizeof(crt_bvh_node4q) * c->bvh.nodes4q.size());
        else if (hipMemcpy(nodes4q, c->dNodes, sizeof(crt_bvh_node4q) * c->bvh.nNodes4, hipMemcpyDeviceToHost) != hipSuccess) return CRT_EHIP;
    }
    return CRT_OK;
}

int crt_bvh_quantize4(const crt_bvh_node4* nodes4, uint32_t n, crt_bvh_node4q* out)
{
    if ((!nodes4 || !out) && n) return fail(nullptr, CRT_EINVAL, "crt_bvh_quantize4: NULL argument");
    for (uint32_t i = 0; i < n; i++) crt::quantizeNode4(nodes4[i], out[i]);
    return CRT_OK;
}

int crt_bvh_build_host4(const crt_mesh_view* meshes, uint32_t n_meshes, crt_bvh_node4** nodes4, uint32_t* n_nodes4, uint32_t* depth4)
{
    if ((!meshes && n_meshes) || !nodes4 || !n_nodes4) return fail(nullptr, CRT_EINVAL, "crt_bvh_build_host4: NULL argument");
    try {
        crt::Bvh b;
        crt::buildBvh(meshes, n_meshes, b);
        *n_nodes4 = static_cast<uint32_t>(b.nodes4.size());
        if (depth4) *depth4 = b.depth4;
        *nodes4 = static_cast<crt_bvh_node4*>(std::malloc(sizeof(crt_bvh_node4) * (b.nodes4.size() + 1)));
        if (!*nodes4) return fail(nullptr, CRT_ENOMEM, "out of host memory");
        crt::copyBytes(*nodes4, b.nodes4.data(), sizeof(crt_bvh_node4) * b.nodes4.size());
    } catch (const std::exception& ex) {
        return fail(nullptr, CRT_EINVAL, "BVH build failed: %s", ex.what());
    }
    return CRT_OK;
}

int crt_bvh_export(const crt_ctx* c, crt_bvh_node* nodes, crt_bvh_tri* tris, crt_bvh_shade* shade)
{
    if (!c || !c->haveScene) return CRT_ESTATE;
    if (nodes) {
        if (!c->dBinNodes) crt::copyBytes(nodes, c->bvh.nodes.data(), sizeof(crt_bvh_node) * c->bvh.nodes.size());
        else if (hipMemcpy(nodes, c->dBinNodes, sizeof(crt_bvh_node) * c->bvh.nNodes, hipMemcpyDeviceToHost) != hipSuccess) return CRT_EHIP;
    }
    // a tree built on the GPU keeps its leaf-ordered records in HBM only: copy them out of there
    const bool onDevice = c->bvh.tris.empty() && c->bvh.nTris != 0;
    if (tris) {
        if (!onDevice) crt::copyBytes(tris, c->bvh.tris.data(), sizeof(crt_bvh_tri) * c->bvh.tris.size());
        else if (hipMemcpy(tris, c->dTris, sizeof(crt_bvh_tri) * c->bvh.nTris, hipMemcpyDeviceToHost) != hipSuccess) return CRT_EHIP;
    }
    if (shade) {
        if (!onDevice) crt::copyBytes(shade, c->bvh.shade.data(), sizeof(crt_bvh_shade) * c->bvh.shade.size());
        else if (hipMemcpy(shade, c->dShade, sizeof(crt_bvh_shade) * c->bvh.nTris, hipMemcpyDeviceToHost) != hipSuccess) return CRT_EHIP;
    }
    return CRT_OK;
}


// ------------------------------------------------------------------------------------------- native RCCL gather
// One process per GPU; every rank renders its macro tiles into a tile-major staging buffer, ONE ncclAllGather per frame moves
// them over xGMI (1 044 480 bytes per rank at 1080p / 8 GPUs, each rank's 7 inbound messages on 7 different links), the
// untile kernel rebuilds the row-major frame: all on the context's stream, no host synchronisation in between.
// RCCL is resolved at run time (dlopen): a process that already carries an RCCL (torch's librccl.so.1) shares that copy, a
// C++-only process takes /opt/rocm/lib's; a process that never calls crt_comm_* needs none.
namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*allGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*getErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// loaded once, whichever thread asks first (the initialiser of a function-local static runs exactly once; later callers wait for it)
RcclApi loadRccl()
{
    RcclApi api;
    for (const char* name : { "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" }) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) {
        const char* why = dlerror(); // may be NULL
        api.error = std::string("cannot load RCCL (librccl.so.1): ") + (why ? why : "unknown dlopen error");
        return api;
    }
    api.getUniqueId = reinterpret_cast<decltype(api.getUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
    api.commInitRank = reinterpret_cast<decltype(api.commInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
    api.commDestroy = reinterpret_cast<decltype(api.commDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
    api.allGather = reinterpret_cast<decltype(api.allGather)>(dlsym(api.handle, "ncclAllGather"));
    api.getErrorString = reinterpret_cast<decltype(api.getErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
    if (!api.getUniqueId || !api.commInitRank || !api.commDestroy || !api.allGather || !api.getErrorString) {
        api.error = "RCCL library lacks an expected entry point";
        api.handle = nullptr;
    }
    return api;
}

RcclApi& rccl()
{
    static RcclApi api = loadRccl();
    return api;
}

int ensureBuffer(crt_ctx* c, void** ptr, size_t* have, size_t need)
{
    if (*have >= need) return CRT_OK;
    HIP_TRY(c, hipDeviceSynchronize());
    if (*ptr) (void)hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    HIP_TRY(c, hipMalloc(ptr, need));
    *have = need;
    return CRT_OK;
}
} // namespace

int crt_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(nullptr, CRT_EINVAL, "crt_comm_unique_id: NULL argument");
    RcclApi& api = rccl();
    if (!api.handle) return fail(nullptr, CRT_ENODEVICE, "%s", api.error.c_str());
    ncclUniqueId id;
    const ncclResult_t r = api.getUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, CRT_EHIP, "ncclGetUniqueId failed: %s", api.getErrorString(r));
    static_assert(sizeof(id) == CRT_COMM_ID_BYTES, "crt_hip.h states the size of an RCCL unique id");
    std::memcpy(id_out, &id, sizeof(id));
    return CRT_OK;
}

int crt_comm_init(crt_ctx* c, uint32_t rank, uint32_t n_ranks, const void* unique_id)
{
    if (!c) return CRT_EINVAL;
    if (!unique_id || n_ranks == 0 || rank >= n_ranks) return fail(c, CRT_EINVAL, "crt_comm_init: bad arguments (rank %u of %u)", rank, n_ranks);
    if (c->comm) return fail(c, CRT_ESTATE, "crt_comm_init: the context already has a communicator (crt_comm_destroy first)");
    RcclApi& api = rccl();
    if (!api.handle) return fail(c, CRT_ENODEVICE, "%s", api.error.c_str());
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t r = api.commInitRank(&c->comm, static_cast<int>(n_ranks), id, static_cast<int>(rank));
    if (r != ncclSuccess) {
        c->comm = nullptr;
        return fail(c, CRT_EHIP, "ncclCommInitRank(rank %u of %u) failed: %s", rank, n_ranks, api.getErrorString(r));
    }
    c->commRank = rank;
    c->commRanks = n_ranks;
    return CRT_OK;
}

// ---- the same frame assembly with shared host memory as the transport: for ranks that share ONE GPU (RCCL refuses that), i.e. for
// rehearsing the multi-rank path on a one-GPU machine, and as a fallback where RCCL cannot be loaded.  A POSIX shared-memory object
// holds a header (arrival counter + generation of a sense-reversing barrier) and the ranks' tile slices; per frame: copy the own
// slice in, barrier, copy all slices out, barrier.  Never a measurement of anything.
struct HostExchange {
    struct Header {
        std::atomic<uint32_t> magic, arrive, generation;
        uint32_t nRanks;
        uint64_t capacity;
    };
    static constexpr uint32_t kMagic = 0x43525431u;
    static constexpr size_t kDataAt = 4096;
    int fd = -1;
    void* map = nullptr;
    size_t bytes = 0;
    std::string name;
    bool owner = false;
    Header* header() const { return static_cast<Header*>(map); }
    unsigned char* data() const { return static_cast<unsigned char*>(map) + kDataAt; }
    bool barrier(uint32_t n) const // false: a peer did not arrive within a minute
    {
        Header* h = header();
        const uint32_t gen = h->generation.load(std::memory_order_acquire);
        if (h->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            h->arrive.store(0, std::memory_order_relaxed);
            h->generation.fetch_add(1, std::memory_order_acq_rel);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (h->generation.load(std::memory_order_acquire) == gen) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        return true;
    }
    void close()
    {
        if (map) munmap(map, bytes);
        if (fd >= 0) ::close(fd);
        if (owner && !name.empty()) shm_unlink(name.c_str());
        map = nullptr;
        fd = -1;
    }
};

int crt_comm_init_host(crt_ctx* c, uint32_t rank, uint32_t n_ranks, const char* name)
{
    if (!c) return CRT_EINVAL;
    if (!name || name[0] != '/' || n_ranks == 0 || rank >= n_ranks) return fail(c, CRT_EINVAL, "crt_comm_init_host: bad arguments (rank %u of %u, name must start with '/')", rank, n_ranks);
    if (c->comm || c->hostComm) return fail(c, CRT_ESTATE, "crt_comm_init_host: the context already has a communicator (crt_comm_destroy first)");
    std::unique_ptr<HostExchange> x(new HostExchange);
    x->name = name;
    x->bytes = HostExchange::kDataAt + (size_t(1) << 28); // room for a 8K RGBA8 frame; pages are only taken when touched
    if (rank == 0) {
        x->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (x->fd < 0) return fail(c, CRT_EIO, "crt_comm_init_host: cannot create shared memory '%s': %s", name, std::strerror(errno));
        x->owner = true;
        if (ftruncate(x->fd, static_cast<off_t>(x->bytes)) != 0) {
            const int e = errno;
            x->close();
            return fail(c, CRT_EIO, "crt_comm_init_host: cannot size '%s': %s", name, std::strerror(e));
        }
    } else {
        for (int tries = 0; tries < 6000 && x->fd < 0; tries++) { // up to 60 s for rank 0 to come up
            x->fd = shm_open(name, O_RDWR, 0600);
            if (x->fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (x->fd < 0) return fail(c, CRT_EIO, "crt_comm_init_host: rank %u found no shared memory '%s'", rank, name);
        struct stat st;
        for (int tries = 0; tries < 6000; tries++) { // ... and to size it
            if (fstat(x->fd, &st) == 0 && static_cast<size_t>(st.st_size) >= x->bytes) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    x->map = mmap(nullptr, x->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, x->fd, 0);
    if (x->map == MAP_FAILED) {
        const int e = errno;
        x->map = nullptr;
        x->close();
        return fail(c, CRT_EIO, "crt_comm_init_host: cannot map '%s': %s", name, std::strerror(e));
    }
    HostExchange::Header* h = x->header();
    if (rank == 0) {
        h->arrive.store(0);
        h->generation.store(0);
        h->nRanks = n_ranks;
        h->capacity = x->bytes - HostExchange::kDataAt;
        h->magic.store(HostExchange::kMagic, std::memory_order_release);
    } else {
        int tries = 0;
        while (h->magic.load(std::memory_order_acquire) != HostExchange::kMagic && tries++ < 6000) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (h->magic.load(std::memory_order_acquire) != HostExchange::kMagic || h->nRanks != n_ranks) {
            x->close();
            return fail(c, CRT_EIO, "crt_comm_init_host: '%s' is not this launch's exchange (%u ranks expected)", name, n_ranks);
        }
    }
    if (!x->barrier(n_ranks)) { // collective, like crt_comm_init: everybody is attached before anybody goes on (and before rank 0 may unlink)
        x->close();
        return fail(c, CRT_EIO, "crt_comm_init_host: not all %u ranks arrived within a minute", n_ranks);
    }
    c->hostComm = x.release();
    c->commRank = rank;
    c->commRanks = n_ranks;
    return CRT_OK;
}

int crt_comm_destroy(crt_ctx* c)
{
    if (!c) return CRT_EINVAL;
    if (c->hostComm) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        c->hostComm->close();
        delete c->hostComm;
        c->hostComm = nullptr;
        c->commRanks = 0;
        return CRT_OK;
    }
    if (!c->comm) return CRT_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    const ncclResult_t r = rccl().commDestroy(c->comm);
    c->comm = nullptr;
    c->commRanks = 0;
    return r == ncclSuccess ? CRT_OK : fail(c, CRT_EHIP, "ncclCommDestroy failed: %s", rccl().getErrorString(r));
}

int crt_comm_info(const crt_ctx* c, uint32_t* rank, uint32_t* n_ranks)
{
    if (!c) return CRT_EINVAL;
    if (rank) *rank = c->commRank;
    if (n_ranks) *n_ranks = c->commRanks; // 0: no communicator
    return CRT_OK;
}

int crt_render_frame_distributed(crt_ctx* c, uint32_t w, uint32_t h, void* d_rgba8, uint8_t* host_rgba8, crt_frame_stats* stats)
{
    int rc = checkRenderable(c, w, h);
    if (rc) return rc;
    if (!c->comm && !c->hostComm) return fail(c, CRT_ESTATE, "crt_render_frame_distributed: no communicator (crt_comm_init first)");
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t n = c->commRanks, slots = crt_tile_slots(w, h, n);
    const size_t per = static_cast<size_t>(slots) * crt::kTile * crt::kTile * 4; // bytes each rank contributes
    const uint32_t k = c->distSerial++ % crt_ctx::kRing;
    if ((rc = ensureBuffer(c, &c->dStage[k], &c->stageBytes[k], per)) != CRT_OK) return rc;
    if ((rc = ensureBuffer(c, &c->dGather[k], &c->gatherBytes[k], per * n)) != CRT_OK) return rc;
    void* frame = d_rgba8;
    if (!frame) {
        if ((rc = ensureBuffer(c, &c->dDistFrame[k], &c->distFrameBytes[k], static_cast<size_t>(w) * h * 4)) != CRT_OK) return rc;
        frame = c->dDistFrame[k];
    }
    // the slot's previous frame (4 frames ago) may have been issued on another stream: order behind it on the GPU
    if (c->distPending[k] && c->distStream[k] != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evDist[k], 0));
    crt_frame_stats local;
    rc = crt_render_tiles_device(c, w, h, c->commRank, n, c->dStage[k], stats ? &local : nullptr);
    if (rc) return rc;
    if (c->comm) {
        const ncclResult_t r = rccl().allGather(c->dStage[k], c->dGather[k], per, ncclUint8, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, CRT_EHIP, "ncclAllGather failed: %s", rccl().getErrorString(r));
    } else { // through shared host memory: own slice in, everybody waits, all slices out, everybody waits again before the next frame's writes
        HostExchange* x = c->hostComm;
        if (per * n > x->header()->capacity) return fail(c, CRT_EINVAL, "crt_render_frame_distributed: frame too large for the host exchange");
        HIP_TRY(c, hipMemcpyAsync(x->data() + per * c->commRank, c->dStage[k], per, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!x->barrier(n)) return fail(c, CRT_EIO, "crt_render_frame_distributed: a rank did not deliver its tiles within a minute");
        HIP_TRY(c, hipMemcpyAsync(c->dGather[k], x->data(), per * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!x->barrier(n)) return fail(c, CRT_EIO, "crt_render_frame_distributed: a rank did not collect the tiles within a minute");
    }
    rc = crt_untile_device(c, w, h, n, c->dGather[k], frame);
    if (rc) return rc;
    if (host_rgba8) HIP_TRY(c, hipMemcpyAsync(host_rgba8, frame, static_cast<size_t>(w) * h * 4, hipMemcpyDeviceToHost, c->stream));
    if (!c->evDist[k]) HIP_TRY(c, hipEventCreateWithFlags(&c->evDist[k], hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->evDist[k], c->stream));
    c->distStream[k] = c->stream;
    c->distPending[k] = true;
    if (stats || host_rgba8) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (stats) {
        *stats = local; // kernel_ms and the counters describe this rank's tile launch
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return CRT_OK;
}

// ------------------------------------------------------------------------------------------- scene layer
int crt_scene_load(const char* path, crt_scene** out, char* err, size_t err_len)
{
    if (!path || !out) return CRT_EINVAL;
    *out = nullptr;
    crt_scene* s = new (std::nothrow) crt_scene();
    if (!s) return CRT_ENOMEM;
    try {
        s->scene.parseSceneFile(path);
    } catch (const std::exception& ex) {
        if (err && err_len) snprintf(err, err_len, "%s", ex.what());
        delete s;
        return std::strstr(ex.what(), "cannot open") ? CRT_EIO : CRT_EPARSE;
    }
    *out = s;
    return CRT_OK;
}

int crt_scene_save(const crt_scene* s, const char* path, char* err, size_t err_len)
{
    if (!s || !path) return CRT_EINVAL;
    try {
        crt::SceneParser::saveBinary(path, s->scene);
    } catch (const std::exception& ex) {
        if (err && err_len) snprintf(err, err_len, "%s", ex.what());
        return CRT_EIO;
    }
    return CRT_OK;
}

int crt_scene_new(crt_scene** out)
{
    if (!out) return CRT_EINVAL;
    *out = new (std::nothrow) crt_scene();
    return *out ? CRT_OK : CRT_ENOMEM;
}

void crt_scene_free(crt_scene* s) { delete s; }

int crt_scene_add_mesh(crt_scene* s, const float* xyz, uint32_t nv, const uint32_t* idx, uint32_t nt, int32_t material_index)
{
    if (!s || (!xyz && nv) || (!idx && nt)) return CRT_EINVAL;
    for (uint32_t i = 0; i < 3 * nt; i++)
        if (idx[i] >= nv) return CRT_EINVAL;
    crt::Mesh& m = s->scene.addObject();
    m.reserve(nv, 3 * static_cast<size_t>(nt));
    for (uint32_t i = 0; i < nv; i++) m.addVertex(crt::Vector(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    for (uint32_t i = 0; i < 3 * nt; i++) m.addIndex(static_cast<int>(idx[i]));
    m.setMaterialIndex(material_index);
    m.calculateVertexNormals();
    return CRT_OK;
}

int crt_scene_add_light(crt_scene* s, const float pos[3], float intensity)
{
    if (!s || !pos) return CRT_EINVAL;
    s->scene.addLight(crt::Light(crt::Vector(pos[0], pos[1], pos[2]), intensity));
    return CRT_OK;
}

int crt_scene_add_material(crt_scene* s, const crt_material* m)
{
    if (!s || !m) return CRT_EINVAL;
    crt::Material mat;
    mat.setType(static_cast<crt::MaterialType>(m->type <= 4 ? m->type : 0));
    mat.setAlbedo(crt::Vector(m->albedo[0], m->albedo[1], m->albedo[2]));
    mat.setSmoothShading(m->smooth != 0);
    mat.setIor(m->ior);
    s->scene.addMaterial(mat);
    return CRT_OK;
}

uint32_t crt_scene_mesh_count(const crt_scene* s) { return s ? static_cast<uint32_t>(s->scene.getObjects().size()) : 0; }

int crt_scene_mesh(const crt_scene* s, uint32_t i, crt_mesh_view* out)
{
    if (!s || !out || i >= s->scene.getObjects().size()) return CRT_EINVAL;
    const crt::Mesh& m = s->scene.getObjects()[i];
    out->xyz = m.getVertices().empty() ? nullptr : m.getVertices().data()->data();
    out->idx = reinterpret_cast<const uint32_t*>(m.getIndices().data());
    out->normals = m.getVertexNormals().size() == m.getVertices().size() && !m.getVertexNormals().empty()
                       ? m.getVertexNormals().data()->data() : nullptr;
    out->uvs = (m.getUV().size() == m.getVertices().size() && !m.getUV().empty()) ? m.getUV().data()->data() : nullptr;
    out->n_vertices = static_cast<uint32_t>(m.getVertices().size());
    out->n_triangles = static_cast<uint32_t>(m.getIndices().size() / 3);
    out->material_index = m.getMaterialIndex();
    return CRT_OK;
}

uint32_t crt_scene_light_count(const crt_scene* s) { return s ? static_cast<uint32_t>(s->scene.getLights().size()) : 0; }

int crt_scene_light(const crt_scene* s, uint32_t i, crt_light* out)
{
    if (!s || !out || i >= s->scene.getLights().size()) return CRT_EINVAL;
    const crt::Light& l = s->scene.getLights()[i];
    crt::copyBytes(out->pos, l.getPosition().data(), 12);
    out->intensity = l.getIntensity();
    return CRT_OK;
}

uint32_t crt_scene_material_count(const crt_scene* s) { return s ? static_cast<uint32_t>(s->scene.getMaterials().size()) : 0; }

int crt_scene_material(const crt_scene* s, uint32_t i, crt_material* out)
{
    if (!s || !out || i >= s->scene.getMaterials().size()) return CRT_EINVAL;
    const crt::Material& m = s->scene.getMaterials()[i];
    crt::copyBytes(out->albedo, m.getAlbedo().data(), 12);
    out->type = static_cast<uint32_t>(m.getType());
    out->smooth = m.isSmoothShading() ? 1u : 0u;
    out->ior = m.getIor();
    out->texture = m.isTexture() ? s->scene.textureIndexByName(m.getTextureName()) : -1; // getTextureByName, R/CRTScene.cpp:52-63
    return CRT_OK;
}

uint32_t crt_scene_texture_count(const crt_scene* s) { return s ? static_cast<uint32_t>(s->scene.getTextures().size()) : 0; }

int crt_scene_texture_color(const crt_scene* s, uint32_t i, float u, float v, float out_rgb[3])
{
    if (!s || !out_rgb || i >= s->scene.getTextures().size()) return CRT_EINVAL;
    const crt::Vector c = s->scene.getTextures()[i].getColor(u, v);
    crt::copyBytes(out_rgb, c.data(), 12);
    return CRT_OK;
}

int crt_scene_add_texture(crt_scene* s, const char* name, const char* type, const float color_a[3], const float color_b[3], float scalar,
                          const char* file_path)
{
    if (!s || !name || !type) return CRT_EINVAL;
    crt::TextureDesc t;
    t.name = name;
    t.type = type;
    if (t.type != "albedo" && t.type != "edges" && t.type != "checker" && t.type != "bitmap") return CRT_EINVAL;
    if (color_a) t.colorA = crt::Vector(color_a[0], color_a[1], color_a[2]);
    if (color_b) t.colorB = crt::Vector(color_b[0], color_b[1], color_b[2]);
    t.scalar = scalar;
    if (file_path) t.filePath = file_path;
    if (t.typeCode() == 3u) {
        try {
            t.loadBitmap(std::string());
        } catch (const std::exception&) {
            return CRT_EIO;
        }
    }
    s->scene.addTexture(t);
    return CRT_OK;
}

int crt_scene_set_material_texture(crt_scene* s, uint32_t material, const char* texture_name)
{
    if (!s || !texture_name || material >= s->scene.getMaterials().size()) return CRT_EINVAL;
    s->scene.materialsRef()[material].setTextureName(texture_name);
    return CRT_OK;
}

int crt_scene_set_mesh_uvs(crt_scene* s, uint32_t mesh, const float* uvs)
{
    if (!s || !uvs || mesh >= s->scene.getObjects().size()) return CRT_EINVAL;
    crt::Mesh& m = s->scene.objectsRef()[mesh];
    std::vector<crt::Vector> uv(m.getVertices().size());
    for (size_t i = 0; i < uv.size(); i++) uv[i] = crt::Vector(uvs[3 * i], uvs[3 * i + 1], uvs[3 * i + 2]);
    m.setUVs(std::move(uv));
    return CRT_OK;
}

int crt_scene_settings(const crt_scene* s, uint32_t* width, uint32_t* height, float background_rgb[3])
{
    if (!s) return CRT_EINVAL;
    const crt::Settings& st = s->scene.getSettings();
    if (width) *width = static_cast<uint32_t>(st.imageWidth);
    if (height) *height = static_cast<uint32_t>(st.imageHeight);
    if (background_rgb) crt::copyBytes(background_rgb, st.backgroundColor.data(), 12);
    return CRT_OK;
}

int crt_scene_camera_get(const crt_scene* s, float pos[3], float rot[9])
{
    if (!s) return CRT_EINVAL;
    if (pos) crt::copyBytes(pos, s->scene.getCamera().getPosition().data(), 12);
    if (rot) crt::copyBytes(rot, s->scene.getCamera().getRotationMatrix().data(), 36);
    return CRT_OK;
}

int crt_scene_camera_set(crt_scene* s, const float pos[3], const float rot[9])
{
    if (!s) return CRT_EINVAL;
    if (pos) s->scene.getCamera().setPosition(crt::Vector(pos[0], pos[1], pos[2]));
    if (rot) s->scene.getCamera().setRotationMatrix(crt::Matrix(rot[0], rot[1], rot[2], rot[3], rot[4], rot[5], rot[6], rot[7], rot[8]));
    return CRT_OK;
}

#define CAMERA_OP(name, call)                        \
    int name                                         \
    {                                                \
        if (!s) return CRT_EINVAL;                   \
        s->scene.getCamera().call;                   \
        return CRT_OK;                               \
    }
CAMERA_OP(crt_scene_camera_rotate(crt_scene* s, float dyaw, float dpitch), rotate(dyaw, dpitch))
CAMERA_OP(crt_scene_camera_zoom(crt_scene* s, float amount), zoom(amount))
CAMERA_OP(crt_scene_camera_move_forward(crt_scene* s, float d), moveForward(d))
CAMERA_OP(crt_scene_camera_move_right(crt_scene* s, float d), moveRight(d))
CAMERA_OP(crt_scene_camera_pan(crt_scene* s, float deg), pan(deg))
CAMERA_OP(crt_scene_camera_tilt(crt_scene* s, float deg), tilt(deg))
CAMERA_OP(crt_scene_camera_roll(crt_scene* s, float deg), roll(deg))
#undef CAMERA_OP

int crt_scene_camera_pan_around_target(crt_scene* s, float degrees, const float target[3])
{
    if (!s || !target) return CRT_EINVAL;
    s->scene.getCamera().panAroundTarget(degrees, crt::Vector(target[0], target[1], target[2]));
    return CRT_OK;
}

int crt_upload_scene_from(crt_ctx* c, const crt_scene* s)
{
    if (!c) return CRT_EINVAL;
    if (!s) return fail(c, CRT_EINVAL, "scene is NULL");
    const uint32_t nm = crt_scene_mesh_count(s), nl = crt_scene_light_count(s), nmat = crt_scene_material_count(s);
    std::vector<crt_mesh_view> meshes(nm);
    std::vector<crt_light> lights(nl);
    std::vector<crt_material> mats(nmat);
    for (uint32_t i = 0; i < nm; i++) crt_scene_mesh(s, i, &meshes[i]);
    for (uint32_t i = 0; i < nl; i++) crt_scene_light(s, i, &lights[i]);
    for (uint32_t i = 0; i < nmat; i++) crt_scene_material(s, i, &mats[i]);
    int rc = crt_upload_scene(c, meshes.data(), nm, lights.data(), nl, mats.data(), nmat);
    if (rc) return rc;
    std::vector<crt_texture> tex;
    for (const crt::TextureDesc& t : s->scene.getTextures()) {
        crt_texture x{};
        x.type = t.typeCode();
        crt::copyBytes(x.color_a, t.colorA.data(), 12);
        crt::copyBytes(x.color_b, t.colorB.data(), 12);
        x.scalar = t.scalar;
        x.pixels = t.pixels.empty() ? nullptr : t.pixels.data();
        x.width = static_cast<uint32_t>(t.width);
        x.height = static_cast<uint32_t>(t.height);
        x.channels = static_cast<uint32_t>(t.channels);
        tex.push_back(x);
    }
    rc = crt_set_textures(c, tex.data(), static_cast<uint32_t>(tex.size()));
    if (rc) return rc;
    return crt_set_camera_from(c, s);
}

int crt_set_camera_from(crt_ctx* c, const crt_scene* s)
{
    if (!c) return CRT_EINVAL;
    if (!s) return fail(c, CRT_EINVAL, "scene is NULL");
    return crt_set_camera(c, s->scene.getCamera().getPosition().data(), s->scene.getCamera().getRotationMatrix().data());
}

} // extern "C"
