// Split packets (option "split_units", renderKernel<..., SPLIT = true>): the most expensive 8x8 packets of a frame are rendered
// by 64 / R wavefronts, one per block of R pixels (option "split_rays": R = 16, 8 or 4; blocks 4x4, 4x2, 2x2), and inside such
// a wavefront the 64 lanes share the work of the block's R rays.
//
// Why: a launch lasts as long as its slowest wavefront, a wavefront as long as its slowest ray, and a grazing ray walks hundreds
// of dependent steps -- on one GPU the launch order hides that (the slow packets start first, the chip stays full), on an N-rank
// tile share it is the whole launch time (tools/timeline_share.py: at 8 ranks the chip is a third full after a third of the
// launch and ten wavefronts run on alone to 175 us).
// How: every ray is cut into K pieces (option "split_segments": 4, 8 or 16) of its way through the scene's box, (t_k, t_k+1), neighbours overlapping by a
// rounding so that the union is all of (tmin, tmax).  The (ray, segment) pairs are work items handed to whichever lane is free,
// early segments first.  A triangle's t does not depend on the interval it was found in, so the ray's closest hit is the hit of
// the LOWEST segment that has one -- same triangle, t, u, v bit for bit (all candidates of an equal-t tie lie in the same
// segments) -- and later segments of a ray that has a hit in an earlier one are dropped or abandoned.  A shadow ray is occluded
// if any segment is.  (A first version gave every ray four fixed lanes and four fixed segments: the pieces of a grazing ray are
// nothing like equally long -- the quarters of the worst packet ran 145..182 us against 175 us unsplit.)
// Results never change; the fetch counters of a split packet do (every segment descends from the root).
#pragma once

#include "shading.hip.h"

namespace crt {
namespace {

constexpr uint32_t kSplitRefillMin = 8;                // idle lanes that trigger a hand-out of new items

__device__ __forceinline__ uint32_t laneIndex() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ float readLaneF(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint32_t readLaneU(uint32_t v, int l) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), l)); }
__device__ __forceinline__ float shflF(float v, uint32_t l) { return __shfl(v, static_cast<int>(l), 64); }
__device__ __forceinline__ uint32_t shflU(uint32_t v, uint32_t l) { return static_cast<uint32_t>(__shfl(static_cast<int>(v), static_cast<int>(l), 64)); }

// segment k of nSeg of [t0, t1].  Neighbours OVERLAP by a thousandth of the lower one's end: a segment's lower end is a tmin
// in the middle of the geometry, and the slab test culls a box whose far distance comes out below tmin -- a far distance that
// rounds differently from the Moeller-Trumbore t of a triangle lying in that box's far face (the same mismatch the cull pad
// covers on the near side, kCullPad).  With the overlap such a triangle is found by the segment before; found twice is harmless
// (closest: same t, u, v; any-hit: an OR).
__device__ __forceinline__ void splitRange(float t0, float t1, uint32_t seg, uint32_t nSeg, float& lo, float& hi)
{
    const float q = (t1 - t0) / static_cast<float>(nSeg);
    const float a = fmaf(q, static_cast<float>(seg), t0), b = fmaf(q, static_cast<float>(seg + 1u), t0);
    lo = a - fabsf(a) * 0x1p-10f;
    hi = b;
}

// the part of (tmin, tmax) a ray spends inside the scene's box; a ray that misses it gets an empty interval at tmin
__device__ __forceinline__ void sceneInterval(const Ray& r, const float* lo, const float* hi, float tmin, float tmax, float& t0, float& t1)
{
    const float x0 = fmaf(lo[0], r.idir.x, r.noid.x), x1 = fmaf(hi[0], r.idir.x, r.noid.x);
    const float y0 = fmaf(lo[1], r.idir.y, r.noid.y), y1 = fmaf(hi[1], r.idir.y, r.noid.y);
    const float z0 = fmaf(lo[2], r.idir.z, r.noid.z), z1 = fmaf(hi[2], r.idir.z, r.noid.z);
    t0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
    t1 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    if (!(t0 <= t1)) t0 = t1 = tmin;
}

// One stream over the (ray, segment) items of the block (raysLog2 = log2 R).  Ray q's data lives in lane q (q < R): origin + direction in
// (ox..dz), its interval [t0, t1], `want` = the ray is to be traced at all.  CLOSEST: on return lane q holds the ray's closest
// hit in best (best.t = tmaxAll on a miss); else `occluded` of lane q says whether anything lies in (t0, t1).
// The first segment starts at tminAll (wave-uniform) and, for closest hits, the last one ends at tmaxAll (wave-uniform), so that
// the union of the segments is the whole open interval whatever the box test's rounding did; an any-hit ray ends at its own t1.
template <bool COUNT, class L, bool CLOSEST>
__device__ __forceinline__ void splitStream(const float4* __restrict__ nodes, const float4* __restrict__ tris, uint32_t n_nodes, F3 o, F3 d, float t0, float t1,
                                            bool want, float tminAll, float tmaxAll, uint32_t raysLog2, uint32_t segsLog2, Stack& stack, int innerMin, Hit& best, bool& occluded,
                                            uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris)
{
    const uint32_t lane = laneIndex();
    const uint32_t kSplitSegments = 1u << segsLog2;
    const uint32_t kSplitItems = kSplitSegments << raysLog2; // rays of the block x segments
    uint32_t bestSeg = kSplitSegments; // lane q: lowest segment of ray q with a hit so far (any-hit: 0 once occluded)
    best.t = tmaxAll; best.u = 0.0f; best.v = 0.0f; best.tri = 0u; best.gid = 0u;
    occluded = false;
    Ray r = makeRay(f3(0.0f, 0.0f, 0.0f), f3(0.0f, 0.0f, 1.0f));
    Hit h;
    h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.tri = 0u; h.gid = 0u;
    float tlo = 0.0f, thi = 0.0f, tcull = 0.0f;
    bool occ = false;
    int cur = L::kDone;
    bool have = false;
    uint32_t myRay = 0u, mySeg = 0u;
    uint32_t next = 0u; // wave-uniform: first item not yet handed out
    const unsigned long long all = __ballot(true);
    for (;;) {
        const bool idle = cur == L::kDone;
        const unsigned long long idleMask = __ballot(idle);
        if (idleMask == all || (next < kSplitItems && static_cast<uint32_t>(__popcll(idleMask)) >= kSplitRefillMin)) {
            // 1. retire: a finished item that found something reports to the lane that owns its ray
            const bool found = idle & have & (CLOSEST ? (h.t < thi) : occ);
            unsigned long long m = __ballot(found);
            while (m) {
                const int l = __ffsll(static_cast<long long>(m)) - 1;
                m &= m - 1ull;
                const uint32_t rayL = readLaneU(myRay, l), segL = readLaneU(mySeg, l);
                if (CLOSEST) {
                    const float t = readLaneF(h.t, l), u = readLaneF(h.u, l), v = readLaneF(h.v, l);
                    const uint32_t tri = readLaneU(h.tri, l), gid = readLaneU(h.gid, l);
                    if (lane == rayL && segL < bestSeg) {
                        bestSeg = segL;
                        best.t = t; best.u = u; best.v = v; best.tri = tri; best.gid = gid;
                    }
                } else {
                    if (lane == rayL) {
                        bestSeg = 0u;
                        occluded = true;
                    }
                }
            }
            if (idle) have = false;
            // 2. a ray that already has a hit in an earlier segment needs none of its later ones: abandon those in flight
            {
                const uint32_t done = shflU(bestSeg, myRay);
                if (have && done <= mySeg && !(CLOSEST && done == mySeg)) {
                    cur = L::kDone;
                    have = false;
                }
            }
            // 3. hand out the next items to the free lanes (segment-major: the early segments of all rays first)
            const bool freeLane = !have;
            const unsigned long long mf = __ballot(freeLane);
            const uint32_t idx = next + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mf >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mf), 0u));
            const bool take = freeLane && idx < kSplitItems;
            const uint32_t candRay = idx & ((1u << raysLog2) - 1u), candSeg = (idx >> raysLog2) & (kSplitSegments - 1u);
            // the item's ray, from the lane that owns it (every lane takes part in the shuffles)
            const float dx = shflF(d.x, candRay), dy = shflF(d.y, candRay), dz = shflF(d.z, candRay);
            const float ox = shflF(o.x, candRay), oy = shflF(o.y, candRay), oz = shflF(o.z, candRay);
            const float c0 = shflF(t0, candRay), c1 = shflF(t1, candRay);
            const uint32_t candDone = shflU(bestSeg, candRay);
            const bool candWant = __shfl(static_cast<int>(want), static_cast<int>(candRay), 64) != 0;
            if (take && candWant && candDone > candSeg) {
                have = true;
                myRay = candRay;
                mySeg = candSeg;
                r = makeRay(f3(ox, oy, oz), f3(dx, dy, dz));
                splitRange(c0, c1, candSeg, kSplitSegments, tlo, thi);
                if (candSeg == 0u) tlo = tminAll;
                if (candSeg == kSplitSegments - 1u) thi = CLOSEST ? tmaxAll : c1; // (any-hit: the ITEM's ray ends at its own t1, not at this lane's)
                h.t = thi; h.u = 0.0f; h.v = 0.0f; h.tri = 0u; h.gid = 0u;
                tcull = thi * kCullPad;
                occ = false;
                stack.sp = 0;
                cur = n_nodes ? L::kRoot : L::kDone;
            }
            next += static_cast<uint32_t>(__popcll(mf));
            if (next >= kSplitItems && __ballot(have) == 0ull) break;
            if (__ballot(cur != L::kDone) == 0ull) continue; // everything handed out was dead: hand out again
        }
        if (CLOSEST) closestIteration<COUNT, L, 8>(nodes, tris, r, tlo, tcull, stack, innerMin, h, cur, iters, cntNodes, cntTris);
        else anyIteration<COUNT, L, 8>(nodes, tris, r, tlo, thi, tcull, stack, innerMin, occ, cur, iters, cntNodes, cntTris);
    }
}

// One block of R pixels of a split packet (part `part` of 64 / R): rayGen, closest hit, shading (incl. the shadow rays of mode
// 100) and the stores of its pixels.  Lane q (q < R) owns pixel q of the block; the other lanes only lend their hands to the
// streams.  Pixel i = part * R + q of the packet sits at the Z-order position of i: x from its even bits, y from its odd bits.
__device__ __forceinline__ uint32_t evenBits3(uint32_t i) { return (i & 1u) | ((i >> 1) & 2u) | ((i >> 2) & 4u); }

template <bool COUNT, bool PHONG, class L>
__device__ __forceinline__ void renderSplitQuarter(const RenderParams& p, const float* camPos, const float* camRot, uint32_t* outRgba8, uint32_t frame,
                                                   uint32_t tile_x, uint32_t tile_y, uint32_t wave, uint32_t part, uint32_t raysLog2, uint32_t segsLog2, Stack& stack, uint32_t& iters,
                                                   uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow, uint32_t& cntClosest)
{
    const float4* nodes = reinterpret_cast<const float4*>(p.nodes);
    const float4* tris = reinterpret_cast<const float4*>(p.tris);
    const int innerMin = static_cast<int>(p.tune_inner_min);
    const uint32_t lane = laneIndex(), q = lane & ((1u << raysLog2) - 1u);
    const uint32_t pixelInPacket = (part << raysLog2) | q;
    const uint32_t lx = (wave & 1u) * 8u + evenBits3(pixelInPacket), ly = (wave >> 1) * 8u + evenBits3(pixelInPacket >> 1);
    const uint32_t px = tile_x * kTile + lx, py = tile_y * kTile + ly;
    const bool owner = lane < (1u << raysLog2) && px < p.width && py < p.height;
    const F3 o = f3(camPos[0], camPos[1], camPos[2]);
    const Ray r = makeRay(o, rayDir(camRot, px, py, static_cast<float>(p.width), static_cast<float>(p.height)));
    float t0, t1;
    sceneInterval(r, p.scene_lo, p.scene_hi, kTMin, kTMax, t0, t1);
    if (COUNT && owner) cntClosest++;
    Hit h;
    bool unusedOcc;
    splitStream<COUNT, L, true>(nodes, tris, p.n_nodes, r.o, r.d, t0, t1, owner, kTMin, kTMax, raysLog2, segsLog2, stack, innerMin, h, unusedOcc, iters, cntNodes, cntTris);
    const bool hit = owner && h.t < kTMax;
    F3 col = f3(p.miss[0], p.miss[1], p.miss[2]); // miss shader (hlsl:72-76)
    uint32_t inst = 0xFFFFFFFFu, prim = 0xFFFFFFFFu;
    if (hit) {
        const float4* T = L::triPtr(tris, h.tri);
        inst = __float_as_uint(T[0].w);
        prim = __float_as_uint(T[1].w);
    }
    if (p.mode >= 100u) {
        // shadeLambert / directLight (shading.hip.h), with every shadow ray traced as a stream of segments by all 64 lanes
        Surface sf;
        sf.P = f3(0.f, 0.f, 0.f); sf.N = f3(0.f, 0.f, 1.f); sf.albedo = f3(0.f, 0.f, 0.f); sf.mtype = 1u; sf.entering = true; sf.ior = 1.0f;
        if (hit) sf = surfaceAt<L>(p, tris, r, h);
        const F3 Po = biasPoint(sf.P, sf.N, kShadowBias), view = f3(-r.d.x, -r.d.y, -r.d.z);
        F3 rgb = f3(0.0f, 0.0f, 0.0f);
        const LightRec* lights = reinterpret_cast<const LightRec*>(p.lights);
        for (uint32_t li = 0; li < p.n_lights; li++) {
            const LightRec Lt = lights[li];
            const F3 Lv = sub3(f3(Lt.x, Lt.y, Lt.z), Po);
            const float r2 = dot3(Lv, Lv);
            const float dist = sqrtf(r2);
            const float invr = 1.0f / dist;
            const F3 Ld = f3(Lv.x * invr, Lv.y * invr, Lv.z * invr);
            const float cosv = fmaxf(0.0f, dot3(sf.N, Ld));
            const bool need = hit && cosv > 0.0f;
            if (COUNT && need) cntShadow++;
            Hit unusedHit;
            bool occluded;
            splitStream<COUNT, L, false>(nodes, tris, p.n_nodes, Po, Ld, 0.0f, dist, need, 0.0f, dist, raysLog2, segsLog2, stack, static_cast<int>(p.tune_inner_min_any), unusedHit, occluded, iters, cntNodes, cntTris);
            if (need && !occluded) {
                const float k = (Lt.intensity / (kFourPi * r2)) * cosv;
                rgb.x = fmaf(sf.albedo.x, k, rgb.x);
                rgb.y = fmaf(sf.albedo.y, k, rgb.y);
                rgb.z = fmaf(sf.albedo.z, k, rgb.z);
                if (PHONG && p.phong_ks > 0.0f) {
                    const float nl2 = 2.0f * dot3(sf.N, Ld);
                    const F3 R = f3(fmaf(nl2, sf.N.x, -Ld.x), fmaf(nl2, sf.N.y, -Ld.y), fmaf(nl2, sf.N.z, -Ld.z));
                    const float rv = fmaxf(0.0f, dot3(R, view));
                    const float sp = (p.phong_ks * (Lt.intensity / (kFourPi * r2))) * powUint(rv, p.phong_exp);
                    rgb.x += sp; rgb.y += sp; rgb.z += sp;
                }
            }
        }
        if (hit) col = rgb;
    } else if (hit) {
        col = shadeDebug(p.mode, inst, prim, h.t, h.u, h.v, r.o, r.d);
    }
    if (owner) {
        const uint32_t packed = unorm8(col.x) | (unorm8(col.y) << 8) | (unorm8(col.z) << 16) | 0xFF000000u;
        const size_t pix = static_cast<size_t>(py) * p.width + px;
        if (p.staging) outRgba8[static_cast<size_t>((tile_y * p.tiles_x + tile_x) / p.n_ranks) * (kTile * kTile) + ly * kTile + lx] = packed;
        else outRgba8[pix] = packed;
        if (frame == 0u) {
            if (p.hit_inst) p.hit_inst[pix] = inst;
            if (p.hit_prim) p.hit_prim[pix] = prim;
            if (p.hit_t) p.hit_t[pix] = hit ? h.t : kTMax;
            if (p.rgb_f32) {
                p.rgb_f32[3 * pix + 0] = col.x;
                p.rgb_f32[3 * pix + 1] = col.y;
                p.rgb_f32[3 * pix + 2] = col.z;
            }
        }
    }
}

} // namespace
} // namespace crt
