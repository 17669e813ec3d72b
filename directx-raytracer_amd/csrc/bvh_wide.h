// The two rules that turn a binary BVH into what the kernels traverse, written once for the host builder (g++) and the
// GPU builder (hipcc): which binary nodes a 4-wide node absorbs (DESIGN.md "BVH4 collapse") and how a wide node is
// quantised to 64 bytes (DESIGN.md "Quantised nodes").  oracle/crt_oracle.c restates both independently.
#pragma once

#include "../../include/crt_hip.h"

#include <cmath>
#include <cstdint>
#include <limits>

#ifdef __HIPCC__
#define CRT_HD __host__ __device__
#else
#define CRT_HD
#endif

namespace crt {

struct WideSlot {
    int32_t ref;      // binary node index (>= 0) or leaf reference (< 0)
    float mn[3], mx[3];
    uint32_t depth;   // depth of the referenced node in the binary tree (root = 0); meaningful for ref >= 0
};

CRT_HD inline float slotHalfArea(const WideSlot& s)
{
    const float dx = s.mx[0] - s.mn[0], dy = s.mx[1] - s.mn[1], dz = s.mx[2] - s.mn[2];
    return (dx * dy + dy * dz) + dz * dx;
}

CRT_HD inline void binaryChildren(const crt_bvh_node& N, uint32_t depthOfN, WideSlot& l, WideSlot& r)
{
    l.ref = N.left;
    r.ref = N.right;
    l.mn[0] = N.lx0; l.mx[0] = N.lx1; l.mn[1] = N.ly0; l.mx[1] = N.ly1; l.mn[2] = N.lz0; l.mx[2] = N.lz1;
    r.mn[0] = N.rx0; r.mx[0] = N.rx1; r.mn[1] = N.ry0; r.mx[1] = N.ry1; r.mn[2] = N.rz0; r.mx[2] = N.rz1;
    l.depth = r.depth = depthOfN + 1u;
}

// The slots of the W-wide node rooted at binary node `b`: its two children, then, while fewer than W, the inner slot of
// largest half-area (first on ties) replaced in place by its two children (left keeps the position, right goes right after
// it).  Returns the slot count; *deepest = the largest depth + 1 over b and the binary nodes absorbed.
template <int W>
CRT_HD inline int wideSlotsT(const crt_bvh_node* nodes, int32_t b, uint32_t depthOfB, WideSlot* sl /* [W] */, uint32_t* deepest)
{
    int n = 2;
    binaryChildren(nodes[b], depthOfB, sl[0], sl[1]);
    uint32_t deep = depthOfB + 1u;
    while (n < W) {
        int best = -1;
        float bestArea = -1.0f;
        for (int i = 0; i < n; i++) {
            if (sl[i].ref < 0) continue;
            const float a = slotHalfArea(sl[i]);
            if (a > bestArea) { bestArea = a; best = i; }
        }
        if (best < 0) break;
        if (sl[best].depth + 1u > deep) deep = sl[best].depth + 1u;
        WideSlot l, r;
        binaryChildren(nodes[sl[best].ref], sl[best].depth, l, r);
        for (int i = n; i > best + 1; i--) sl[i] = sl[i - 1];
        sl[best] = l;
        sl[best + 1] = r;
        n++;
    }
    if (deepest) *deepest = deep;
    return n;
}
CRT_HD inline int wideSlots(const crt_bvh_node* nodes, int32_t b, uint32_t depthOfB, WideSlot sl[4], uint32_t* deepest)
{
    return wideSlotsT<4>(nodes, b, depthOfB, sl, deepest);
}

// the wide node of n slots, inner refs still binary indices; unused slots: an inverted box and CRT_BVH_EMPTY (quantised: a point)
CRT_HD inline void fillWide(const WideSlot sl[4], int n, crt_bvh_node4& W)
{
    for (int i = 0; i < 4; i++) {
        if (i < n) {
            W.minx[i] = sl[i].mn[0]; W.maxx[i] = sl[i].mx[0];
            W.miny[i] = sl[i].mn[1]; W.maxy[i] = sl[i].mx[1];
            W.minz[i] = sl[i].mn[2]; W.maxz[i] = sl[i].mx[2];
            W.ref[i] = sl[i].ref;
        } else {
            const float inf = std::numeric_limits<float>::infinity();
            W.minx[i] = W.miny[i] = W.minz[i] = inf;
            W.maxx[i] = W.maxy[i] = W.maxz[i] = -inf;
            W.ref[i] = CRT_BVH_EMPTY;
        }
        W.pad[i] = 0;
    }
}

CRT_HD inline float decodePlane(uint32_t q, float s, float lo) { return fmaf(static_cast<float>(q), s, lo); }

// Quantised nodes.  Per axis: lo / hi = the node's own extent over the children whose box is finite and ordered on that
// axis; quantum s = (hi - lo) / 255 nudged up so that fma(255, s, lo) >= hi; a child's planes are the largest q with
// fma(q, s, lo) <= min and the smallest q with fma(q, s, lo) >= max (the decode expression itself is what is checked, so
// the decoded box contains the full-precision one whatever the rounding).  A child that is not finite on an axis spans
// the whole node there (q = 0..255).  Extents beyond 3e38 are clamped (coordinates that large are not supported).
CRT_HD inline void quantizeNode4(const crt_bvh_node4& W, crt_bvh_node4q& Q)
{
    const float* mins[3] = { W.minx, W.miny, W.minz };
    const float* maxs[3] = { W.maxx, W.maxy, W.maxz };
    uint32_t qlo[3] = { 0, 0, 0 }, qhi[3] = { 0, 0, 0 };
    const float inf = std::numeric_limits<float>::infinity();
    const float tiny = 1.17549435e-38f; // FLT_MIN
    for (int a = 0; a < 3; a++) {
        float lo = inf, hi = -inf;
        bool valid[4];
        for (int k = 0; k < 4; k++) {
            const float mn = mins[a][k], mx = maxs[a][k];
            valid[k] = W.ref[k] != CRT_BVH_EMPTY && mn - mn == 0.0f && mx - mx == 0.0f && mn <= mx; // x - x == 0: finite
            if (valid[k]) {
                lo = mn < lo ? mn : lo;
                hi = mx > hi ? mx : hi;
            }
        }
        if (!(lo <= hi)) lo = hi = 0.0f; // no finite child on this axis
        float ext = hi - lo;
        if (!(ext < 3.0e38f)) ext = 3.0e38f;
        float s = (ext * (1.0f / 255.0f)) * 1.000001f;
        if (!(s >= tiny)) s = tiny;
        Q.lo[a] = lo;
        Q.s[a] = s;
        for (int k = 0; k < 4; k++) {
            uint32_t l = 0, h = 255;
            if (W.ref[k] == CRT_BVH_EMPTY) { // a point at the node's minimum corner (crt_hip.h)
                l = 0;
                h = 0;
            } else if (valid[k]) {
                const float fl = (mins[a][k] - lo) / s, fh = (maxs[a][k] - lo) / s;
                l = fl >= 255.0f ? 255u : (fl > 0.0f ? static_cast<uint32_t>(fl) : 0u);
                while (l > 0 && decodePlane(l, s, lo) > mins[a][k]) l--;
                h = fh >= 255.0f ? 255u : (fh > 0.0f ? static_cast<uint32_t>(fh) : 0u);
                while (h < 255 && decodePlane(h, s, lo) < maxs[a][k]) h++;
            }
            qlo[a] |= l << (8 * k);
            qhi[a] |= h << (8 * k);
        }
    }
    Q.qlo_x = qlo[0]; Q.qhi_x = qhi[0];
    Q.qlo_y = qlo[1]; Q.qhi_y = qhi[1];
    Q.qlo_z = qlo[2]; Q.qhi_z = qhi[2];
    for (int k = 0; k < 4; k++) Q.ref[k] = W.ref[k];
}

} // namespace crt
