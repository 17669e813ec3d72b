// The packed wide tree: what sits in HBM and what the kernels traverse (DESIGN.md "Packed wide tree").  One buffer holds the
// node records AND the triangle records; the children of a node -- inner nodes' records and leaves' triangle runs, in slot
// order -- lie back to back in one "child block", so a node names all of them with ONE base address plus a small per-child
// offset.  That is what shrinks a 4-wide node from 64 to 48 bytes (three 16-byte fetches per lane instead of four) and lets
// an 8-wide node fit 80 bytes (five).  Written once for the host builder (g++) and the GPU builder (hipcc);
// oracle/crt_oracle.c restates it independently and the tests compare the bytes.
//
// Addresses count granules from the start of the buffer: 48 bytes for W = 4 (node = triangle = one granule), 16 bytes for
// W = 8 (node 5, triangle 3).  A reference is (address << 3) | kind: kind 7 = inner node, 0..6 = leaf of that many
// triangles stored back to back from `address` (0 = the empty leaf of an unused slot); kRefDone ends a traversal.
//
// Node record, dwords:              W = 4 (12 dwords)                      W = 8 (20 dwords)
//   0..2   lo.xyz: minimum corner of the node's own box
//   3      base8 = (address of the child block) << 3
//   4      scales: bits 0..9 s.x, 10..19 s.y, 21..30 s.z; each = the top 10 bits of a positive float (8-bit exponent,
//          2-bit mantissa): s = as_float(field << 21), rounded UP from the quantum (extent / 255) it has to cover
//   5..    child designators m_k = (offset_k << 3) | kind_k, offset in granules from the child block's start:
//                                   dword 5, byte k                        dwords 5..7, 10 bits each, three per dword
//   then   planes, byte k of a word = child k (W = 8: two words per plane set, children 0..3 and 4..7):
//                                   6..11: qlo_x qhi_x qlo_y qhi_y qlo_z qhi_z   8..19: qlo_x[2] qhi_x[2] qlo_y[2] ...
// A child's box on axis a is [fma(qlo, s_a, lo_a), fma(qhi, s_a, lo_a)], rounded outwards with the decode expression itself.
#pragma once

#include "bvh_wide.h"


namespace crt {

constexpr uint32_t kRefInner = 7u;
constexpr uint32_t kRefDone = 0xFFFFFFFEu;
constexpr uint32_t kPackLeafMax = 6u; // triangles a leaf reference can name (kind 0..6)

template <int W> struct PackFmt;
template <> struct PackFmt<4> {
    static constexpr uint32_t kGranuleBytes = 48, kNodeGranules = 1, kTriGranules = 1, kNodeDwords = 12, kPayloadBits = 8, kPlaneDword = 6;
};
template <> struct PackFmt<8> {
    static constexpr uint32_t kGranuleBytes = 16, kNodeGranules = 5, kTriGranules = 3, kNodeDwords = 20, kPayloadBits = 10, kPlaneDword = 8;
};

// full-precision W-wide node, the builders' intermediate form (DFS pre-order; ref >= 0: wide node index, < 0: leaf reference
// of the binary tree ~((first << 3) | count) into the leaf-ordered triangle array, CRT_BVH_EMPTY: unused slot)
template <int W> struct WideNodeT {
    float mn[3][W], mx[3][W];
    int32_t ref[W];
};

template <int W>
CRT_HD inline void fillWideT(const WideSlot* sl, int n, WideNodeT<W>& N)
{
    const float inf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < W; k++) {
        for (int a = 0; a < 3; a++) {
            N.mn[a][k] = k < n ? sl[k].mn[a] : inf;
            N.mx[a][k] = k < n ? sl[k].mx[a] : -inf;
        }
        N.ref[k] = k < n ? sl[k].ref : CRT_BVH_EMPTY;
    }
}

CRT_HD inline uint32_t packFloatBits(float f) { return __builtin_bit_cast(uint32_t, f); }
CRT_HD inline float packBitsFloat(uint32_t u) { return __builtin_bit_cast(float, u); }

// the 10-bit scale that covers quantum `need` (> 0, finite): smallest representable value >= need, at least FLT_MIN
CRT_HD inline uint32_t encodeScale(float need)
{
    uint32_t f = (packFloatBits(need) + 0x1FFFFFu) >> 21;
    if (f < 4u) f = 4u;          // 2^-126
    if (f > 0x3FBu) f = 0x3FBu;  // 1.75 * 2^127: stays finite
    return f;
}
CRT_HD inline float decodeScale(uint32_t field) { return packBitsFloat(field << 21); }

// granules a child occupies in its parent's block, and its kind
template <int W>
CRT_HD inline uint32_t childGranules(int32_t ref)
{
    if (ref >= 0) return PackFmt<W>::kNodeGranules;
    return (static_cast<uint32_t>(~ref) & 7u) * PackFmt<W>::kTriGranules;
}
template <int W>
CRT_HD inline uint32_t blockGranules(const WideNodeT<W>& N)
{
    uint32_t g = 0;
    for (int k = 0; k < W; k++) g += childGranules<W>(N.ref[k]);
    return g;
}

// Quantise node N, whose child block starts at granule `base`, into its record `out` (PackFmt<W>::kNodeDwords dwords).
// Planes: per axis lo / hi = the node's own extent over the children whose box is finite and ordered there; quantum =
// (hi - lo) / 255 nudged up, then rounded up to the 10-bit scale; a child's planes are the largest q with
// fma(q, s, lo) <= min and the smallest q with fma(q, s, lo) >= max.  A child that is not finite on an axis spans the
// whole node there; an unused slot is the point at the minimum corner (q = 0, 0), kind 0 at offset 0.
template <int W>
CRT_HD inline void packNode(const WideNodeT<W>& N, uint32_t base, uint32_t* out)
{
    typedef PackFmt<W> F;
    const float inf = std::numeric_limits<float>::infinity();
    for (uint32_t i = 0; i < F::kNodeDwords; i++) out[i] = 0u;
    uint32_t scales = 0;
    for (int a = 0; a < 3; a++) {
        float lo = inf, hi = -inf;
        bool valid[W];
        for (int k = 0; k < W; k++) {
            const float mn = N.mn[a][k], mx = N.mx[a][k];
            valid[k] = N.ref[k] != CRT_BVH_EMPTY && mn - mn == 0.0f && mx - mx == 0.0f && mn <= mx; // x - x == 0: finite
            if (valid[k]) {
                lo = mn < lo ? mn : lo;
                hi = mx > hi ? mx : hi;
            }
        }
        if (!(lo <= hi)) lo = hi = 0.0f; // no finite child on this axis
        float ext = hi - lo;
        if (!(ext < 3.0e38f)) ext = 3.0e38f;
        const uint32_t field = encodeScale((ext * (1.0f / 255.0f)) * 1.000001f);
        const float s = decodeScale(field);
        out[a] = packFloatBits(lo);
        scales |= field << (a == 2 ? 21 : 10 * a);
        for (int k = 0; k < W; k++) {
            uint32_t l = 0, h = 255;
            if (N.ref[k] == CRT_BVH_EMPTY) {
                l = 0;
                h = 0;
            } else if (valid[k]) {
                const float fl = (N.mn[a][k] - lo) / s, fh = (N.mx[a][k] - lo) / s;
                l = fl >= 255.0f ? 255u : (fl > 0.0f ? static_cast<uint32_t>(fl) : 0u);
                while (l > 0 && decodePlane(l, s, lo) > N.mn[a][k]) l--;
                h = fh >= 255.0f ? 255u : (fh > 0.0f ? static_cast<uint32_t>(fh) : 0u);
                while (h < 255 && decodePlane(h, s, lo) < N.mx[a][k]) h++;
            }
            const uint32_t words = W / 4; // words per plane set
            const uint32_t dl = F::kPlaneDword + (2u * a) * words + static_cast<uint32_t>(k) / 4u;
            const uint32_t dh = F::kPlaneDword + (2u * a + 1u) * words + static_cast<uint32_t>(k) / 4u;
            out[dl] |= l << (8 * (k & 3));
            out[dh] |= h << (8 * (k & 3));
        }
    }
    out[3] = base << 3;
    out[4] = scales;
    uint32_t off = 0;
    for (int k = 0; k < W; k++) {
        const int32_t ref = N.ref[k];
        const uint32_t kind = ref >= 0 ? kRefInner : (static_cast<uint32_t>(~ref) & 7u);
        const uint32_t m = ref == CRT_BVH_EMPTY ? 0u : ((off << 3) | kind);
        if (W == 4) out[5] |= m << (8 * k);
        else out[5 + k / 3] |= m << (10 * (k % 3));
        off += childGranules<W>(ref);
    }
}

} // namespace crt
