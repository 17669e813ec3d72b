// HIP kernel for gfx950 (MI355X): mode 200, path tracing (BASELINE.json configs[4]) -- no counterpart in the reference, whose
// closestHit shader only colours the primary hit (R/HLSL/ray_tracing_shaders.hlsl:78-169); specification: oracle trace_path.
#include "shading.hip.h"

namespace crt {
namespace {

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
constexpr uint32_t kPathCounterStride = 16; // dwords between the work counters of two ranges (one 64-byte line each)

struct PathScratch {
    float4* shade;   // kShadePlanes x B: {o, rng} {d, id | bounce << 16} {t, u, v, tri}
    float4* trace;   // kTracePlanes x B: {o, rng} {d, id | bounce << 16}
    float4* done;    // B, by path id (sample-in-pass * tile pixels + pixel-in-tile): the path's radiance so far, final when it ends
    float4* thr;     // B, by path id: its throughput (only the shade stage changes it; the queues carry the ray, not this)
    float4* accum;   // 256: running sum over the samples of earlier passes
    uint32_t B;      // plane stride of the queues
    uint32_t* take;  // GLOBAL queues only: cursor of the queue a stream consumes ...
    uint32_t* put;   // ... and length of the queue it fills
    uint32_t chunk;  // ... entries reserved per atomic
};

// A path's id shares a word with its bounce count.  Wavefront-private queues: 16 + 16 bits (<= 1024 paths per workgroup);
// global queues (the stages as separate launches): 25 + 7 bits (<= 2^25 paths per pass, max_bounces <= 64).
template <bool GLOBAL> struct PathId {
    static constexpr uint32_t kShift = GLOBAL ? 25u : 16u;
    static constexpr uint32_t kMask = (1u << kShift) - 1u;
    static constexpr uint32_t kOne = 1u << kShift;
};

// one atomic per wavefront: lane 0 reserves n slots for all lanes (call in wave-uniform control flow)
__device__ __forceinline__ uint32_t waveReserve(uint32_t* counter, uint32_t n)
{
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(counter, n);
    return __builtin_amdgcn_readfirstlane(base);
}

__device__ __forceinline__ uint32_t lanePrefix(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
}

// A wavefront's window on two GLOBAL queues (the stages as separate launches): entries are reserved a chunk at a time -- one atomic
// per `chunk` entries on the consumed queue's cursor and on the filled queue's length (an atomic on one address takes about 6 ns of
// the whole device's time: one per refill, 3.4 M per frame, made the pipeline five times slower than the traversal it feeds).  The
// first chunk consumed is the wavefront's own (chunk number = blockIdx.x, no atomic); later ones come from the cursor, which starts
// behind the grid's own chunks.  What a wavefront has reserved of the filled queue and not used when it ends is written as NULL
// entries (id word 0xFFFFFFFF), which a consumer skips.  Everything here is wave-uniform.
constexpr uint32_t kNullPath = 0xFFFFFFFFu;
struct GlobalTap {
    uint32_t inNext, inEnd, outNext, outEnd, n, chunk;
    bool dry; // the consumed queue's cursor has passed its end
    __device__ __forceinline__ void begin(uint32_t length, uint32_t chunkEntries)
    {
        n = length;
        chunk = chunkEntries;
        inNext = min(blockIdx.x * chunk, n);
        inEnd = min(inNext + chunk, n);
        outNext = outEnd = 0u;
        dry = false;
    }
    __device__ __forceinline__ bool more() const { return (inNext < inEnd) | !dry; }
    // slots of the consumed queue for the lanes of `mask` (call in wave-uniform control flow); a lane's slot is valid if < its end
    __device__ __forceinline__ uint32_t take(uint32_t* cursor, unsigned long long mask, bool& valid)
    {
        const uint32_t want = static_cast<uint32_t>(__popcll(mask)), avail = inEnd - inNext;
        uint32_t nb = 0u, nbEnd = 0u;
        if ((want > avail) & !dry) {
            nb = gridDim.x * chunk + waveReserve(cursor, chunk);
            if (nb >= n) { dry = true; nb = 0u; }
            else nbEnd = min(nb + chunk, n);
        }
        const uint32_t pre = lanePrefix(mask);
        const uint32_t idx = pre < avail ? inNext + pre : nb + (pre - avail);
        valid = (pre < avail) | (idx < nbEnd);
        if (want > avail) { inNext = min(nb + (want - avail), nbEnd); inEnd = nbEnd; }
        else inNext += want;
        return idx;
    }
    // slots of the filled queue for the lanes of `mask` (at most 64)
    __device__ __forceinline__ uint32_t put(uint32_t* length, unsigned long long mask)
    {
        const uint32_t need = static_cast<uint32_t>(__popcll(mask)), room = outEnd - outNext;
        uint32_t nb = 0u;
        if (need > room) nb = waveReserve(length, chunk);
        const uint32_t pre = lanePrefix(mask);
        const uint32_t idx = pre < room ? outNext + pre : nb + (pre - room);
        if (need > room) { outNext = nb + (need - room); outEnd = nb + chunk; }
        else outNext += need;
        return idx;
    }
    // the reserved and unused rest of the filled queue becomes null entries (plane1 = the plane that holds the id word)
    __device__ __forceinline__ void flush(float4* plane1)
    {
        for (uint32_t i = outNext + (threadIdx.x & 63u); i < outEnd; i += 64u) plane1[i] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kNullPath));
        outNext = outEnd;
    }
};

// Stage C as a stream: the bounce rays of the trace queue are incoherent and their traversals differ a lot in length, so
// a chunk-at-a-time loop leaves most lanes idle while the longest ray of each chunk finishes.  Here a lane that has finished
// retires its ray (miss -> radiance written, hit -> appended to the shade queue) and takes the next entry of the queue,
// as soon as at least `refillMin` lanes are idle: the wavefront stays full until the queue runs dry.  Every ray is still
// traced by one lane in its own fixed order, so results and fetch counts are those of the chunked loop.
#ifndef CRT_REFILL_MIN
#define CRT_REFILL_MIN 16
#endif
// GLOBAL: the queues are shared by every wavefront of the launch (pathTraceKernel): entries are taken with one atomic on the
// queue's cursor per refill and appended with one atomic on the other queue's length per retirement.
template <bool COUNT, class L, bool GLOBAL>
__device__ __forceinline__ void streamClosest(const float4* __restrict__ nodes, const float4* __restrict__ tris, uint32_t n_nodes,
                                              const PathScratch& q, uint32_t nTrace, uint32_t& nShade, F3 miss, Stack& stack, int innerMin,
                                              uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntClosest)
{
    Ray r = makeRay(f3(0.0f, 0.0f, 0.0f), f3(0.0f, 0.0f, 1.0f));
    Hit h;
    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.gid = 0;
    float tcull = kTMax * kCullPad;
    int cur = L::kDone;
    bool have = false;   // this lane holds a ray (being traced, or finished and not yet retired)
    uint32_t my = 0;     // its index in the trace queue
    uint32_t next = 0;   // wave-uniform: first queue entry not yet handed to a lane
    GlobalTap tap;
    tap.begin(nTrace, q.chunk);
    const unsigned long long all = __ballot(true);
    for (;;) {
        const bool idle = cur == L::kDone;
        const unsigned long long idleMask = __ballot(idle);
        const bool more = GLOBAL ? tap.more() : next < nTrace;
        if (idleMask == all || (more && static_cast<uint32_t>(__popcll(idleMask)) >= static_cast<uint32_t>(CRT_REFILL_MIN))) {
            // retire the finished rays ...
            const bool retire = idle & have, isHit = retire & (h.t < kTMax);
            const unsigned long long mh = __ballot(isHit);
            uint32_t slot = nShade + lanePrefix(mh);
            if (GLOBAL) slot = tap.put(q.put, mh);
            if (retire) {
                const float4 a0 = q.trace[my], a1 = q.trace[q.B + my];
                if (isHit) {
                    const uint32_t k = slot;
                    q.shade[k] = a0;
                    q.shade[q.B + k] = a1;
                    q.shade[2u * q.B + k] = make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
                } else {
                    const uint32_t id = __float_as_uint(a1.w) & PathId<GLOBAL>::kMask;
                    const float4 a2 = q.thr[id], a3 = q.done[id];
                    q.done[id] = make_float4(fmaf(a2.x, miss.x, a3.x), fmaf(a2.y, miss.y, a3.y), fmaf(a2.z, miss.z, a3.z), 0.0f);
                }
            }
            nShade += static_cast<uint32_t>(__popcll(mh));
            // ... and hand the next queue entries to the idle lanes
            bool valid = true;
            uint32_t idx = next + lanePrefix(idleMask);
            if (GLOBAL) idx = tap.take(q.take, idleMask, valid);
            else valid = idx < nTrace;
            if (idle) {
                have = valid;
                if (have) {
                    my = idx;
                    if (GLOBAL && __float_as_uint(q.trace[q.B + idx].w) == kNullPath) have = false; // reserved by a producer and never filled
                }
                if (have) {
                    const float4 a0 = q.trace[idx], a1 = q.trace[q.B + idx];
                    r = makeRay(f3(a0.x, a0.y, a0.z), f3(a1.x, a1.y, a1.z));
                    h.t = kTMax; h.u = 0.0f; h.v = 0.0f; h.tri = 0; h.gid = 0;
                    tcull = kTMax * kCullPad;
                    stack.sp = 0;
                    cur = n_nodes ? L::kRoot : L::kDone;
                    if (COUNT) cntClosest++;
                }
            }
            next += static_cast<uint32_t>(__popcll(idleMask));
            if (__ballot(have) == 0ull && !(GLOBAL && tap.more())) break; // queue empty and every ray retired
        }
        closestIteration<COUNT, L, 8>(nodes, tris, r, 0.0f, tcull, stack, innerMin, h, cur, iters, cntNodes, cntTris);
    }
    if (GLOBAL) tap.flush(q.shade + q.B);
}

// Stage B as a stream (same idea as streamClosest): the entries of the shade queue are shaded by whichever lane is free.
// A lane's life with one entry: fetch (surface, material; mirror / glass / constant finish at once) -> for each light with a
// positive cosine, in light order: one any-hit shadow ray, its contribution added when unoccluded (oracle: direct_light) ->
// retire (radiance update, next direction drawn, appended to the trace queue or written out as finished).  Shadow rays end at
// their first hit, so their traversals differ even more in length than the bounce rays': refilling keeps the wavefront full.
template <bool COUNT, class L, bool GLOBAL>
__device__ __forceinline__ void streamShade(const RenderParams& p, const float4* __restrict__ nodes, const float4* __restrict__ tris,
                                            const PathScratch& q, uint32_t nShade, uint32_t& nTrace, Stack& stack, int innerMin,
                                            uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow)
{
    const LightRec* lights = reinterpret_cast<const LightRec*>(p.lights);
    Ray sr = makeRay(f3(0.0f, 0.0f, 0.0f), f3(0.0f, 0.0f, 1.0f)); // the shadow ray in flight
    float dist = 0.0f, tcull = 0.0f, kcur = 0.0f;                  // its length, cull bound, and the light's weight if it arrives
    bool occluded = false;
    int cur = L::kDone;
    bool have = false, diffuse = false, tracing = false, alive = false;
    float thrMul = 0.0f;  // 1: throughput *= albedo when the path goes on; -1: CONSTANT (radiance += throughput * albedo)
    uint32_t my = 0, li = 0;
    F3 Po = f3(0.f, 0.f, 0.f), N = f3(0.f, 0.f, 1.f), albedo = f3(0.f, 0.f, 0.f);
    F3 aux = f3(0.f, 0.f, 0.f); // DIFFUSE: direct light gathered so far; REFLECTIVE / REFRACTIVE: the next direction
    uint32_t next = 0;
    GlobalTap tap;
    tap.begin(nShade, q.chunk);
    const unsigned long long all = __ballot(true);
    for (;;) {
        const bool idle = cur == L::kDone;
        const unsigned long long idleMask = __ballot(idle);
        const bool more = GLOBAL ? tap.more() : next < nShade;
        if (idleMask == all || (more && static_cast<uint32_t>(__popcll(idleMask)) >= static_cast<uint32_t>(CRT_REFILL_MIN))) {
            // 1. a shadow ray has come back: its light counts unless something is in the way
            if (idle & tracing) {
                if (!occluded) aux = f3(fmaf(albedo.x, kcur, aux.x), fmaf(albedo.y, kcur, aux.y), fmaf(albedo.z, kcur, aux.z));
                tracing = false;
                li++;
            }
            // 2. retire: the entries whose lights are all done (or that never had any to ask)
            const bool retire = idle & have & !(diffuse & (li < p.n_lights));
            F3 thr = f3(0.f, 0.f, 0.f), Lr = f3(0.f, 0.f, 0.f), nd = f3(0.f, 0.f, 0.f);
            uint32_t rng = 0, idb = 0;
            bool goesOn = false;
            if (retire) {
                rng = __float_as_uint(q.shade[my].w);
                idb = __float_as_uint(q.shade[q.B + my].w);
                const float4 a2 = q.thr[idb & PathId<GLOBAL>::kMask], a3 = q.done[idb & PathId<GLOBAL>::kMask];
                thr = f3(a2.x, a2.y, a2.z);
                Lr = f3(a3.x, a3.y, a3.z);
                goesOn = alive;
                nd = aux;
                if (thrMul < 0.0f) Lr = f3(fmaf(thr.x, albedo.x, Lr.x), fmaf(thr.y, albedo.y, Lr.y), fmaf(thr.z, albedo.z, Lr.z));
                if (diffuse) {
                    Lr = f3(fmaf(thr.x, aux.x, Lr.x), fmaf(thr.y, aux.y, Lr.y), fmaf(thr.z, aux.z, Lr.z));
                    if ((idb >> PathId<GLOBAL>::kShift) != p.max_bounces) {
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
                    if (goesOn) q.thr[idb & PathId<GLOBAL>::kMask] = make_float4(thr.x, thr.y, thr.z, 0.0f);
                }
                q.done[idb & PathId<GLOBAL>::kMask] = make_float4(Lr.x, Lr.y, Lr.z, 0.0f); // final if the path ends here, else the sum so far
                have = false;
            }
            const unsigned long long mOn = __ballot(goesOn);
            uint32_t slot = nTrace + lanePrefix(mOn);
            if (GLOBAL) slot = tap.put(q.put, mOn);
            if (goesOn) {
                const uint32_t k = slot;
                q.trace[k] = make_float4(Po.x, Po.y, Po.z, __uint_as_float(rng));
                q.trace[q.B + k] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(idb + PathId<GLOBAL>::kOne)); // next bounce
            }
            nTrace += static_cast<uint32_t>(__popcll(mOn));
            // 3. fetch: the lanes without an entry take the next ones of the queue
            const bool wantNew = idle & !have;
            const unsigned long long mNew = __ballot(wantNew);
            bool valid = true;
            uint32_t idx = next + lanePrefix(mNew);
            if (GLOBAL) idx = tap.take(q.take, mNew, valid);
            else valid = idx < nShade;
            if (GLOBAL && wantNew && valid && __float_as_uint(q.shade[q.B + idx].w) == kNullPath) valid = false; // reserved by a producer and never filled
            if (wantNew && valid) {
                have = true;
                my = idx;
                const float4 a0 = q.shade[idx], a1 = q.shade[q.B + idx];
                const Ray r = makeRay(f3(a0.x, a0.y, a0.z), f3(a1.x, a1.y, a1.z));
                const uint32_t bounce = __float_as_uint(a1.w) >> PathId<GLOBAL>::kShift;
                Hit h;
                const float4 a4 = q.shade[2u * q.B + idx];
                h.t = a4.x; h.u = a4.y; h.v = a4.z; h.tri = __float_as_uint(a4.w); h.gid = 0;
                const Surface sf = surfaceAt<L>(p, tris, r, h);
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
            if ((cur == L::kDone) & have & diffuse & !tracing) {
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
                        cur = p.n_nodes ? L::kRoot : L::kDone;
                        if (COUNT) cntShadow++;
                        break;
                    }
                    li++;
                }
            }
            if (__ballot(have) == 0ull && !(GLOBAL && tap.more())) break;
        }
        anyIteration<COUNT, L, 8>(nodes, tris, sr, 0.0f, dist, tcull, stack, innerMin, occluded, cur, iters, cntNodes, cntTris);
    }
    if (GLOBAL) tap.flush(q.trace + q.B);
}

#ifndef CRT_PATH_WAVES_PER_EU
#define CRT_PATH_WAVES_PER_EU 5
#endif
// Persistent wavefronts: the grid is what the chip can hold at once (pathGridSize), and every workgroup takes pixel tiles from
// a shared counter until none is left.  Scratch (queues, stack spill arena) therefore belongs to the RESIDENT workgroup, not to
// the tile: 5120 regions instead of 130 560 at 3840x2160 (3.9 GB -> 152 MB at 4 spp), and a workgroup that finishes a cheap
// tile goes on with the next one instead of leaving its wave slot to a fresh launch.  Every wavefront reaches the exit: the
// counter only grows, and a value >= n_work ends the loop.
template <bool COUNT, class L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_PATH_WAVES_PER_EU, 8))) void pathKernel(const RenderParams p)
{
    extern __shared__ int s_stack[];
    const bool big = p.path_tile == 16u;                  // work item = whole macro tile (four 8x8 packets per sample) or one 8x8 packet
    const uint32_t subCount = big ? 4u : 1u;
    const uint32_t tilePixels = subCount * 64u;
    const uint32_t nWork = p.path_work_items;
    uint32_t cntNodes = 0, cntTris = 0, cntShadow = 0, cntClosest = 0, iters = 0;
    const uint32_t lane = threadIdx.x & 63u;
    const float4* nodes = reinterpret_cast<const float4*>(p.nodes);
    const float4* tris = reinterpret_cast<const float4*>(p.tris);
    Stack stack;
    stack.lds = s_stack + lane;
    stack.spill = p.spill + (static_cast<size_t>(blockIdx.x) * 64u + lane) * p.spill_stride;
    stack.cap = static_cast<int>(p.stack_entries);
    stack.sp = 0;
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
        q.take = q.put = nullptr;
        q.chunk = 0u;
    }
    const F3 miss = f3(p.miss[0], p.miss[1], p.miss[2]);
#if CRT_PROF // diagnostic build: cycles and lane use of the three stages (0 = A camera rays, 1 = B shade + shadow rays, 2 = C bounce rays)
    unsigned long long pT[3] = { 0, 0, 0 }, pTN[3] = { 0, 0, 0 }, pTL[3] = { 0, 0, 0 };
    uint32_t pIN[3] = { 0, 0, 0 }, pIL[3] = { 0, 0, 0 }, pLN[3] = { 0, 0, 0 }, pLL[3] = { 0, 0, 0 };
    unsigned long long pT0 = 0;
    const unsigned long long pK0 = __builtin_amdgcn_s_memtime(), pR0 = __builtin_amdgcn_s_memrealtime();
#define CRT_PATH_PROF_BEGIN() { stack.tNode = stack.tLeaf = 0; stack.itNode = stack.itLeaf = stack.lanesNode = stack.lanesLeaf = 0; pT0 = __builtin_amdgcn_s_memtime(); }
#define CRT_PATH_PROF_END(S) { pT[S] += __builtin_amdgcn_s_memtime() - pT0; pTN[S] += stack.tNode; pTL[S] += stack.tLeaf; pIN[S] += stack.itNode; pIL[S] += stack.itLeaf; pLN[S] += stack.lanesNode; pLL[S] += stack.lanesLeaf; }
#else
#define CRT_PATH_PROF_BEGIN()
#define CRT_PATH_PROF_END(S)
#endif
    // XCD affinity: the work items (which walk the frame in 4x4 blocks of tiles) are cut into p.path_ranges contiguous ranges, one
    // counter each.  A wavefront works through the range of the XCD it runs on (HW_REG_XCC_ID), so the XCD's own L2 keeps
    // serving one part of the frame and of the scene; when that range is used up it goes on with the next XCD's (stealing).
    const uint32_t nRanges = p.path_ranges;
    const uint32_t rangeLen = (nWork + nRanges - 1u) / nRanges;
    uint32_t range = nRanges > 1u ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu) % nRanges : 0u; // HW_REG_XCC_ID[3:0]
    uint32_t tried = 0;
  for (;;) {
    uint32_t item = 0;
    if (lane == 0) item = atomicAdd(p.path_counter + range * kPathCounterStride, 1u);
    item = __builtin_amdgcn_readfirstlane(item) + range * rangeLen;
    if (item >= min(nWork, (range + 1u) * rangeLen)) { // this range is used up (every wavefront gets here: the counters only grow)
        if (++tried >= nRanges) break;
        range = range + 1u == nRanges ? 0u : range + 1u;
        continue;
    }
    iters = 0; // the raised issue priority of a long tile (kBoostAfter) ends with it
    __builtin_amdgcn_s_setprio(0);
    const uint32_t frame = p.n_batch > 1u ? item % p.n_batch : 0u;
    const uint32_t wg = p.n_batch > 1u ? item / p.n_batch : item;
    const uint32_t j = big ? wg : wg >> 2;                // position of the macro tile in this rank's list
    const uint32_t subFirst = big ? 0u : (wg & 3u);
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
    if (!valid) continue;

    for (uint32_t s0 = 0; s0 < p.spp; s0 += p.path_samples) {
        const uint32_t nS = min(p.path_samples, p.spp - s0);
        uint32_t nShade = 0; // wave-uniform queue lengths
        CRT_PATH_PROF_BEGIN()
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
                    traceClosest<COUNT, L>(nodes, tris, p.n_nodes, r, kTMin, kTMax, stack, innerMin, h, iters, cntNodes, cntTris);
                    isHit = h.t < kTMax;
                    // radiance so far: a miss ends the path with throughput (1) x miss colour; a hit starts from nothing, throughput 1
                    q.done[id] = isHit ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fmaf(1.0f, miss.x, 0.0f), fmaf(1.0f, miss.y, 0.0f), fmaf(1.0f, miss.z, 0.0f), 0.0f);
                    if (isHit) q.thr[id] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                    if (s0 + sl == 0u && frame == 0u) { // the hit outputs report sample 0's camera ray
                        const size_t pix = static_cast<size_t>(py) * p.width + px;
                        uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
                        if (isHit) {
                            const float4* T = L::triPtr(tris, h.tri);
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
        CRT_PATH_PROF_END(0)
        while (nShade != 0u) {
            uint32_t nTrace = 0;
            CRT_PATH_PROF_BEGIN()
            streamShade<COUNT, L, false>(p, nodes, tris, q, nShade, nTrace, stack, static_cast<int>(p.tune_inner_min_any), iters, cntNodes, cntTris, cntShadow); // stage B
            CRT_PATH_PROF_END(1)
            nShade = 0;
            CRT_PATH_PROF_BEGIN()
            streamClosest<COUNT, L, false>(nodes, tris, p.n_nodes, q, nTrace, nShade, miss, stack, innerMin, iters, cntNodes, cntTris, cntClosest); // stage C
            CRT_PATH_PROF_END(2)
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
  } // next work item
#if CRT_PROF
    if (COUNT && lane == 0) { // counters[4] = wave lifetime, then 7 words per stage: cycles, in node steps, in leaf steps, node / leaf phases, lanes in them
        atomicAdd(&p.counters[4], __builtin_amdgcn_s_memtime() - pK0);
        atomicAdd(&p.counters[26], __builtin_amdgcn_s_memrealtime() - pR0); // the same on the constant 100 MHz clock: sum of the wavefronts' lives
        atomicAdd(&p.counters[27], 1ull);
        for (int st = 0; st < 3; st++) {
            unsigned long long* c = p.counters + 5 + 7 * st;
            atomicAdd(&c[0], pT[st]); atomicAdd(&c[1], pTN[st]); atomicAdd(&c[2], pTL[st]);
            atomicAdd(&c[3], static_cast<unsigned long long>(pIN[st])); atomicAdd(&c[4], static_cast<unsigned long long>(pIL[st]));
            atomicAdd(&c[5], static_cast<unsigned long long>(pLN[st])); atomicAdd(&c[6], static_cast<unsigned long long>(pLL[st]));
        }
    }
#endif
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

// ---- mode 200, the stages as SEPARATE LAUNCHES over global queues ("wavefront" pipeline, round 3) -----------------------------
// The persistent kernel above keeps a tile's whole pipeline in one wavefront: 96 registers (the union of what the three stages
// hold) with 17 spilled, 5 wavefronts per SIMD, and each stage of each tile drains on its own.  The incoherent stages wait for
// memory (a node step of a bounce ray is a dependent fetch that mostly misses the L2: tools/prof_run.py --path), so what they need
// is more rays in flight.  Compiled on their own the stages need 70 (bounce rays) and 74 (shade + shadow rays) registers without a
// spill: 7 and 6 wavefronts per SIMD.  So: one launch per stage, persistent wavefronts, queues shared by the whole launch --
//   camera   a pixel tile's camera rays, 64 at a time (coherent packets); hits are appended to the shade queue
//   shade    every entry of the shade queue (surface, material, shadow rays, next direction); paths that go on -> trace queue
//   trace    every entry of the trace queue (closest hit of the bounce ray); hits -> shade queue            [x max_bounces]
//   resolve  per pixel: the samples' radiances summed in sample order, averaged, quantised
// A queue's entries are taken 64-lane-refill-wise with one atomic per refill, so a wavefront only drains when the QUEUE is dry
// (once per launch, not once per tile and stage).  Per path nothing changes: same arithmetic, same RNG stream, same order of
// radiance updates; which wavefront carries a path, and the order of the queues' entries, are the only things left to chance,
// and no result depends on them.  Path id = (work item - first item of the pass) * 64 * path_samples + sample * 64 + pixel lane.
constexpr uint32_t kWfRangeDwords = 8u * kPathCounterStride; // camera launch: the work counters of the eight ranges
__device__ __forceinline__ uint32_t* wfQueueLength(const RenderParams& p, uint32_t k) { return p.wf_counts + kWfRangeDwords + 2u * k; }
__device__ __forceinline__ uint32_t* wfQueueCursor(const RenderParams& p, uint32_t k) { return p.wf_counts + kWfRangeDwords + 2u * k + 1u; }

struct WfItem {
    uint32_t tile_x, tile_y, frame;
    bool valid;
};
// work item (index within the rank's list of 8x8 packets x frames of a batch) -> tile; same walk as the persistent kernel's
__device__ __forceinline__ WfItem wfDecode(const RenderParams& p, uint32_t item)
{
    WfItem w;
    w.frame = p.n_batch > 1u ? item % p.n_batch : 0u;
    const uint32_t wg = p.n_batch > 1u ? item / p.n_batch : item;
    const uint32_t j = wg >> 2, sub = wg & 3u; // position of the 16x16 macro tile in this rank's list, 8x8 packet inside it
    uint32_t mx, my;
    if (p.n_ranks == 1) {
        const uint32_t blocks_x = (p.tiles_x + 3u) >> 2;
        const uint32_t blk = j >> 4, within = j & 15u;
        mx = (blk % blocks_x) * 4u + (within & 3u);
        my = (blk / blocks_x) * 4u + (within >> 2);
        w.valid = (mx < p.tiles_x) & (my < p.tiles_y);
    } else {
        const uint32_t k = j * p.n_ranks + p.rank;
        w.valid = k < p.tiles_x * p.tiles_y;
        mx = k % p.tiles_x;
        my = k / p.tiles_x;
    }
    w.tile_x = mx * 2u + (sub & 1u); // in 8-pixel units
    w.tile_y = my * 2u + (sub >> 1);
    return w;
}

__device__ __forceinline__ PathScratch wfScratch(const RenderParams& p)
{
    PathScratch q;
    q.B = p.wf_stride;
    q.chunk = p.wf_chunk;
    q.shade = static_cast<float4*>(p.wf_shade_q);
    q.trace = static_cast<float4*>(p.wf_trace_q);
    q.done = static_cast<float4*>(p.wf_done);
    q.thr = static_cast<float4*>(p.wf_thr);
    q.accum = static_cast<float4*>(p.wf_accum);
    q.take = nullptr;
    q.put = nullptr;
    return q;
}

__device__ __forceinline__ Stack wfStack(const RenderParams& p, int* s_stack)
{
    const uint32_t lane = threadIdx.x & 63u;
    Stack stack;
    stack.lds = s_stack + lane;
    stack.spill = p.spill + (static_cast<size_t>(blockIdx.x) * 64u + lane) * p.spill_stride;
    stack.cap = static_cast<int>(p.stack_entries);
    stack.sp = 0;
    return stack;
}

template <bool COUNT>
__device__ __forceinline__ void wfCount(const RenderParams& p, uint32_t cntNodes, uint32_t cntTris, uint32_t cntShadow, uint32_t cntClosest)
{
    if (COUNT) {
        const uint32_t a = waveSum(cntNodes), c = waveSum(cntTris), sh = waveSum(cntShadow), cl = waveSum(cntClosest);
        if ((threadIdx.x & 63u) == 0u) {
            if (a) atomicAdd(&p.counters[0], static_cast<unsigned long long>(a));
            if (c) atomicAdd(&p.counters[1], static_cast<unsigned long long>(c));
            if (sh) atomicAdd(&p.counters[2], static_cast<unsigned long long>(sh));
            if (cl) atomicAdd(&p.counters[3], static_cast<unsigned long long>(cl));
        }
    }
}

#ifndef CRT_WF_CAMERA_WAVES
#define CRT_WF_CAMERA_WAVES 5
#endif
#ifndef CRT_WF_SHADE_WAVES
#define CRT_WF_SHADE_WAVES 6
#endif
#ifndef CRT_WF_TRACE_WAVES
#define CRT_WF_TRACE_WAVES 7
#endif

template <bool COUNT, class L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_WF_CAMERA_WAVES, 8))) void pathCameraKernel(const RenderParams p)
{
    extern __shared__ int s_stack[];
    const uint32_t lane = threadIdx.x & 63u;
    const float4* nodes = reinterpret_cast<const float4*>(p.nodes);
    const float4* tris = reinterpret_cast<const float4*>(p.tris);
    Stack stack = wfStack(p, s_stack);
    const PathScratch q = wfScratch(p);
    const int innerMin = static_cast<int>(p.tune_inner_min);
    const F3 miss = f3(p.miss[0], p.miss[1], p.miss[2]);
    uint32_t cntNodes = 0, cntTris = 0, cntClosest = 0, iters = 0;
    const uint32_t nWork = p.wf_items, nS = min(p.path_samples, p.spp - p.wf_s0);
    // XCD affinity, as in the persistent kernel: eight contiguous ranges of the pass's work items, own range first
    const uint32_t nRanges = p.path_ranges;
    const uint32_t rangeLen = (nWork + nRanges - 1u) / nRanges;
    uint32_t range = nRanges > 1u ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu) % nRanges : 0u; // HW_REG_XCC_ID[3:0]
    uint32_t tried = 0;
    GlobalTap tap;
    tap.begin(0u, p.wf_chunk);
    for (;;) {
        uint32_t local = 0;
        if (lane == 0) local = atomicAdd(p.wf_counts + range * kPathCounterStride, 1u);
        local = __builtin_amdgcn_readfirstlane(local) + range * rangeLen;
        if (local >= min(nWork, (range + 1u) * rangeLen)) {
            if (++tried >= nRanges) break;
            range = range + 1u == nRanges ? 0u : range + 1u;
            continue;
        }
        iters = 0;
        __builtin_amdgcn_s_setprio(0);
        const WfItem w = wfDecode(p, p.wf_item0 + local);
        if (!w.valid) continue;
        const float* camPos = w.frame ? p.batch_pos[w.frame - 1u] : p.pos;
        const float* camRot = w.frame ? p.batch_rot[w.frame - 1u] : p.rot;
        const uint32_t px = w.tile_x * 8u + (lane & 7u), py = w.tile_y * 8u + (lane >> 3);
        const bool active = (px < p.width) & (py < p.height);
        for (uint32_t sl = 0; sl < nS; sl++) {
            const uint32_t id = local * (64u * p.path_samples) + sl * 64u + lane;
            bool isHit = false;
            Ray r;
            Hit h;
            uint32_t rng = 0;
            if (active) {
                const uint32_t pixId = py * p.width + px;
                rng = pcgHash(pixId ^ pcgHash((p.wf_s0 + sl) + pcgHash(p.seed)));
                const float jx = rngNext(rng), jy = rngNext(rng);
                r = makeRay(f3(camPos[0], camPos[1], camPos[2]), rayDirJ(camRot, px, py, jx, jy, static_cast<float>(p.width), static_cast<float>(p.height)));
                if (COUNT) cntClosest++;
                traceClosest<COUNT, L>(nodes, tris, p.n_nodes, r, kTMin, kTMax, stack, innerMin, h, iters, cntNodes, cntTris);
                isHit = h.t < kTMax;
                // radiance so far: a miss ends the path with throughput (1) x miss colour; a hit starts from nothing, throughput 1
                q.done[id] = isHit ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : make_float4(fmaf(1.0f, miss.x, 0.0f), fmaf(1.0f, miss.y, 0.0f), fmaf(1.0f, miss.z, 0.0f), 0.0f);
                if (isHit) q.thr[id] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                if (p.wf_s0 + sl == 0u && w.frame == 0u) { // the hit outputs report sample 0's camera ray
                    const size_t pix = static_cast<size_t>(py) * p.width + px;
                    uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
                    if (isHit) {
                        const float4* T = L::triPtr(tris, h.tri);
                        inst = __float_as_uint(T[0].w);
                        prim = __float_as_uint(T[1].w);
                    }
                    if (p.hit_inst) p.hit_inst[pix] = inst;
                    if (p.hit_prim) p.hit_prim[pix] = prim;
                    if (p.hit_t) p.hit_t[pix] = isHit ? h.t : kTMax;
                }
            }
            const unsigned long long m = __ballot(isHit);
            const uint32_t slot = tap.put(wfQueueLength(p, 0u), m);
            if (isHit) {
                const uint32_t i = slot;
                q.shade[i] = make_float4(r.o.x, r.o.y, r.o.z, __uint_as_float(rng));
                q.shade[q.B + i] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(id)); // bounce 0 in the upper bits
                q.shade[2u * q.B + i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
            }
        }
    }
    tap.flush(q.shade + q.B);
    wfCount<COUNT>(p, cntNodes, cntTris, 0u, cntClosest);
}

template <bool COUNT, class L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_WF_SHADE_WAVES, 8))) void pathShadeKernel(const RenderParams p)
{
    extern __shared__ int s_stack[];
    const uint32_t nShade = *wfQueueLength(p, p.wf_queue);
    if (nShade == 0u) return;
    Stack stack = wfStack(p, s_stack);
    PathScratch q = wfScratch(p);
    q.take = wfQueueCursor(p, p.wf_queue);
    q.put = wfQueueLength(p, p.wf_queue + 1u);
    uint32_t cntNodes = 0, cntTris = 0, cntShadow = 0, iters = 0, nTrace = 0;
    streamShade<COUNT, L, true>(p, reinterpret_cast<const float4*>(p.nodes), reinterpret_cast<const float4*>(p.tris), q, nShade, nTrace, stack,
                                static_cast<int>(p.tune_inner_min_any), iters, cntNodes, cntTris, cntShadow);
    wfCount<COUNT>(p, cntNodes, cntTris, cntShadow, 0u);
}

template <bool COUNT, class L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRT_WF_TRACE_WAVES, 8))) void pathTraceKernel(const RenderParams p)
{
    extern __shared__ int s_stack[];
    const uint32_t nTrace = *wfQueueLength(p, p.wf_queue);
    if (nTrace == 0u) return;
    Stack stack = wfStack(p, s_stack);
    PathScratch q = wfScratch(p);
    q.take = wfQueueCursor(p, p.wf_queue);
    q.put = wfQueueLength(p, p.wf_queue + 1u);
    uint32_t cntNodes = 0, cntTris = 0, cntClosest = 0, iters = 0, nShade = 0;
    streamClosest<COUNT, L, true>(reinterpret_cast<const float4*>(p.nodes), reinterpret_cast<const float4*>(p.tris), p.n_nodes, q, nTrace, nShade,
                                  f3(p.miss[0], p.miss[1], p.miss[2]), stack, static_cast<int>(p.tune_inner_min), iters, cntNodes, cntTris, cntClosest);
    wfCount<COUNT>(p, cntNodes, cntTris, 0u, cntClosest);
}

// one wavefront per work item of the pass: this pass's samples join the running sums in sample order; after the last pass the
// average is quantised and stored
__global__ __launch_bounds__(64) void pathResolveKernel(const RenderParams p)
{
    const uint32_t lane = threadIdx.x & 63u, local = blockIdx.x;
    const WfItem w = wfDecode(p, p.wf_item0 + local);
    if (!w.valid) return;
    const uint32_t px = w.tile_x * 8u + (lane & 7u), py = w.tile_y * 8u + (lane >> 3);
    if ((px >= p.width) | (py >= p.height)) return;
    const uint32_t nS = min(p.path_samples, p.spp - p.wf_s0);
    const bool last = p.wf_s0 + nS >= p.spp;
    F3 acc = f3(0.0f, 0.0f, 0.0f);
    if (p.wf_s0 != 0u) {
        const float4 a = static_cast<const float4*>(p.wf_accum)[local * 64u + lane];
        acc = f3(a.x, a.y, a.z);
    }
    for (uint32_t sl = 0; sl < nS; sl++) {
        const float4 Ls = static_cast<const float4*>(p.wf_done)[local * (64u * p.path_samples) + sl * 64u + lane];
        acc = f3(acc.x + Ls.x, acc.y + Ls.y, acc.z + Ls.z);
    }
    if (!last) {
        static_cast<float4*>(p.wf_accum)[local * 64u + lane] = make_float4(acc.x, acc.y, acc.z, 0.0f);
        return;
    }
    const float inv = 1.0f / static_cast<float>(p.spp);
    const F3 col = f3(acc.x * inv, acc.y * inv, acc.z * inv);
    const uint32_t packed = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
    uint32_t* outRgba8 = w.frame ? p.batch_rgba8[w.frame - 1u] : p.rgba8;
    const size_t pix = static_cast<size_t>(py) * p.width + px;
    if (p.staging) {
        const uint32_t mx = w.tile_x >> 1, my = w.tile_y >> 1; // 16x16 tile of the frame, pixel inside it
        const uint32_t lx = (w.tile_x & 1u) * 8u + (lane & 7u), ly = (w.tile_y & 1u) * 8u + (lane >> 3);
        outRgba8[static_cast<size_t>((my * p.tiles_x + mx) / p.n_ranks) * (kTile * kTile) + ly * kTile + lx] = packed;
    } else {
        outRgba8[pix] = packed;
    }
    if (p.rgb_f32 && w.frame == 0u) {
        p.rgb_f32[3 * pix + 0] = col.x;
        p.rgb_f32[3 * pix + 1] = col.y;
        p.rgb_f32[3 * pix + 2] = col.z;
    }
}

} // namespace

namespace { uint32_t wfResident(int which, uint32_t stackEntries); } // resident workgroups of the wavefront pipeline's launches (0 camera, 1 shade, 2 trace)

// scratch the path-tracing pipeline needs per resident workgroup: the two queues, the finished-path radiances and the cross-pass
// sums; how many work items (pixel tiles x frames of a batch) a launch has; and how many workgroups it starts
size_t pathRegionBytes(uint32_t tile, uint32_t samples_per_pass)
{
    const size_t pixels = static_cast<size_t>(tile) * tile, B = pixels * samples_per_pass;
    return (kShadePlanes + kTracePlanes + 2u) * B * sizeof(float4) + pixels * sizeof(float4); // queues + radiance + throughput, + cross-pass sums
}
uint32_t pathWorkgroupCount(const RenderParams& p) { return renderUnitCount(p) / (p.path_tile == 16u ? 4u : 1u) * (p.n_batch ? p.n_batch : 1u); }

// resident workgroups of the persistent kernel: what the occupancy calculator allows per CU (registers, LDS) x CUs.  An estimate
// that is one too high per CU only means a few workgroups start late and find the counter exhausted; nothing waits on them.
uint32_t pathGridSize(const RenderParams& p)
{
    static int perCu = 0, cus = 0;
    if (perCu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        const size_t lds = static_cast<size_t>(kStackEntries) * 64u * sizeof(int); // worst case LDS: the grid must fit any stack_entries
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(&pathKernel<false, LayLegacy>), 64, lds) != hipSuccess || perCu <= 0) {
            perCu = 16;
            cus = 256;
        } else {
            cus = prop.multiProcessorCount;
        }
        perCu = perCu > 32 ? 32 : perCu;
    }
    const uint32_t cap = static_cast<uint32_t>(perCu) * static_cast<uint32_t>(cus);
    const uint32_t work = pathWorkgroupCount(p);
    if (p.path_wavefront) { // the largest grid of the pipeline's launches (the stack spill arena is sized by it)
        uint32_t most = 0;
        for (int k = 0; k < 3; k++) most = wfResident(k, p.stack_entries) > most ? wfResident(k, p.stack_entries) : most;
        return most;
    }
    return work < cap ? work : cap;
}

// ---- wavefront pipeline, host side
uint32_t pathWavefrontPassItems(const RenderParams& p, uint32_t max_paths)
{
    const uint32_t perItem = 64u * p.path_samples, total = pathWorkgroupCount(p);
    uint32_t items = max_paths / perItem;
    items = items < 64u ? 64u : items & ~63u; // whole 4x4 blocks of tiles
    return total < items ? total : items;
}

// queue geometry of a pass of `items` work items: entries a wavefront reserves per atomic, and the queues' capacity = plane stride
// (every path once, plus what the wavefronts of ONE launch can leave reserved and unused: less than a chunk each)
void pathWavefrontLayout(const RenderParams& p, uint32_t items, uint32_t& chunk, uint32_t& stride)
{
    const uint32_t paths = items * 64u * p.path_samples;
    uint32_t most = 0;
    for (int k = 0; k < 3; k++) most = wfResident(k, p.stack_entries) > most ? wfResident(k, p.stack_entries) : most;
    const uint32_t waves = paths / 64u < most ? paths / 64u : most; // no launch of the pass starts more wavefronts than this
    chunk = ((paths / (4u * (waves ? waves : 1u))) + 63u) & ~63u;   // about four reservations per wavefront and queue ...
    chunk = chunk < 64u ? 64u : (chunk > 256u ? 256u : chunk);      // ... of 64 to 256 entries
    stride = paths + waves * chunk;
}

size_t pathWavefrontBytes(const RenderParams& p, uint32_t items)
{
    uint32_t chunk, stride;
    pathWavefrontLayout(p, items, chunk, stride);
    const size_t paths = static_cast<size_t>(items) * 64u * p.path_samples;
    size_t bytes = kWfHeadBytes + (static_cast<size_t>(stride) * (kShadePlanes + kTracePlanes) + paths * 2u) * sizeof(float4);
    if (p.spp > p.path_samples) bytes += static_cast<size_t>(items) * 64u * sizeof(float4);
    return bytes;
}

namespace {
// resident workgroups of a persistent launch: what the occupancy calculator allows per CU x CUs
uint32_t residentGroups(const void* kernel, int& cache, uint32_t stackEntries)
{
    static int cus = 0;
    if (cache == 0) {
        int dev = 0, perCu = 0;
        hipDeviceProp_t prop;
        const size_t lds = static_cast<size_t>(stackEntries) * 64u * sizeof(int);
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, 64, lds) != hipSuccess || perCu <= 0) {
            perCu = 16;
            cus = 256;
        } else {
            cus = prop.multiProcessorCount;
        }
        cache = (perCu > 32 ? 32 : perCu) * cus;
    }
    return static_cast<uint32_t>(cache);
}
uint32_t wfResident(int which, uint32_t stackEntries) // 0 camera, 1 shade, 2 trace; the LDS part of the stacks decides with the registers
{
    static int cache[3][kStackEntries + 1] = {};
    const void* k[3] = { reinterpret_cast<const void*>(&pathCameraKernel<false, LayLegacy>), reinterpret_cast<const void*>(&pathShadeKernel<false, LayLegacy>),
                         reinterpret_cast<const void*>(&pathTraceKernel<false, LayLegacy>) };
    const uint32_t e = stackEntries > kStackEntries ? kStackEntries : stackEntries;
    return residentGroups(k[which], cache[which][e], e);
}

int launchPathWavefront(const RenderParams& p0, bool counting, ihipStream_t* stream)
{
    if (!p0.wf_counts || !p0.wf_shade_q || p0.wf_paths == 0u || p0.path_tile != 8u || p0.layout != 0u) return static_cast<int>(hipErrorInvalidValue);
    const uint32_t total = pathWorkgroupCount(p0), perItem = 64u * p0.path_samples;
    const uint32_t passItems = p0.wf_paths / perItem;
    if (passItems == 0u || p0.max_bounces > 64u || p0.wf_paths > (1u << 25)) return static_cast<int>(hipErrorInvalidValue);
    const size_t lds = static_cast<size_t>(p0.stack_entries) * 64u * sizeof(int);
    const dim3 block(64);
    RenderParams p = p0;
    for (uint32_t item0 = 0; item0 < total; item0 += passItems) {
        p.wf_item0 = item0;
        p.wf_items = total - item0 < passItems ? total - item0 : passItems;
        for (uint32_t s0 = 0; s0 < p.spp; s0 += p.path_samples) {
            p.wf_s0 = s0;
            hipError_t e = hipMemsetAsync(p.wf_counts, 0, kWfHeadBytes, stream);
            if (e != hipSuccess) return static_cast<int>(e);
            const uint32_t maxWaves = p.wf_items * perItem / 64u; // no stage has more entries than the pass has paths
            const uint32_t camGrid = p.wf_items < wfResident(0, p.stack_entries) ? p.wf_items : wfResident(0, p.stack_entries);
            if (counting) hipLaunchKernelGGL((pathCameraKernel<true, LayLegacy>), dim3(camGrid), block, lds, stream, p);
            else hipLaunchKernelGGL((pathCameraKernel<false, LayLegacy>), dim3(camGrid), block, lds, stream, p);
            for (uint32_t b = 0; b <= p.max_bounces; b++) {
                p.wf_queue = 2u * b;
                const uint32_t sg = maxWaves < wfResident(1, p.stack_entries) ? maxWaves : wfResident(1, p.stack_entries);
                if (counting) hipLaunchKernelGGL((pathShadeKernel<true, LayLegacy>), dim3(sg), block, lds, stream, p);
                else hipLaunchKernelGGL((pathShadeKernel<false, LayLegacy>), dim3(sg), block, lds, stream, p);
                if (b == p.max_bounces) break;
                p.wf_queue = 2u * b + 1u;
                const uint32_t tg = maxWaves < wfResident(2, p.stack_entries) ? maxWaves : wfResident(2, p.stack_entries);
                if (counting) hipLaunchKernelGGL((pathTraceKernel<true, LayLegacy>), dim3(tg), block, lds, stream, p);
                else hipLaunchKernelGGL((pathTraceKernel<false, LayLegacy>), dim3(tg), block, lds, stream, p);
            }
            hipLaunchKernelGGL(pathResolveKernel, dim3(p.wf_items), block, 0, stream, p);
            e = hipGetLastError();
            if (e != hipSuccess) return static_cast<int>(e);
        }
    }
    return static_cast<int>(hipSuccess);
}
} // namespace

int launchPath(const RenderParams& p, bool counting, ihipStream_t* stream)
{
    if (p.path_wavefront) return launchPathWavefront(p, counting, stream);
    // one wavefront per resident slot carries the paths of one pixel tile after the other through the pipeline
    if (!p.path_counter || p.path_work_items == 0) return static_cast<int>(hipErrorInvalidValue);
    const dim3 grid(pathGridSize(p)), block(64);
    const size_t lds = static_cast<size_t>(p.stack_entries) * 64u * sizeof(int);
#define CRT_LAUNCH(LAY)                                                                                \
    if (counting) hipLaunchKernelGGL((pathKernel<true, LAY>), grid, block, lds, stream, p);            \
    else hipLaunchKernelGGL((pathKernel<false, LAY>), grid, block, lds, stream, p);
#if CRT_PACKED_LAYOUTS
    if (p.layout == 8u) { CRT_LAUNCH(LayPacked<8>) }
    else if (p.layout == 4u) { CRT_LAUNCH(LayPacked<4>) }
    else
#endif
    { CRT_LAUNCH(LayLegacy) }
#undef CRT_LAUNCH
    return static_cast<int>(hipGetLastError());
}

} // namespace crt
