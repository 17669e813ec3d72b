// Device-side ray generation and shading shared by the render kernel and the path-tracing pipeline: rayGen, the reference's
// closest-hit colour functions (modes 0..6), textures, the surface at a hit, direct light (Lambert + optional Phong, shadow
// rays), the counter-based RNG.  Internal to the including translation unit.
#pragma once

#include "traversal.hip.h"

namespace crt {
namespace {

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

template <class L>
__device__ __forceinline__ Surface surfaceAt(const RenderParams& p, const float4* tris, const Ray& r, const Hit& h)
{
    Surface sf;
    const float4* T = L::triPtr(tris, h.tri);
    const float4 tb = T[1], tc = T[2];
    const uint32_t rec = L::shadeIndex(h.tri, __float_as_uint(tc.w)); // shading / uv record: leaf order, or input order (packed tree)
    const float* S = reinterpret_cast<const float*>(p.shade) + 12 * static_cast<size_t>(rec);
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
                    const float* U = reinterpret_cast<const float*>(p.uvs) + 6 * static_cast<size_t>(rec);
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
// direct light at Po: one any-hit shadow ray per light with a positive cosine (oracle: direct_light).  PHONG (mode 100
// only): plus the specular term ks * I / (4 pi r^2) * max(0, R . view)^n, R = the light direction mirrored about N.
template <bool COUNT, class L, bool PHONG>
__device__ __forceinline__ F3 directLight(const RenderParams& p, const float4* nodes, const float4* tris, F3 Po, F3 N, F3 albedo, F3 view,
                                          Stack& stack, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow)
{
    F3 rgb = f3(0.0f, 0.0f, 0.0f);
    const LightRec* lights = reinterpret_cast<const LightRec*>(p.lights);
    for (uint32_t li = 0; li < p.n_lights; li++) {
        const LightRec Lt = lights[li];
        const F3 Lv = sub3(f3(Lt.x, Lt.y, Lt.z), Po);
        const float r2 = dot3(Lv, Lv);
        const float dist = sqrtf(r2);
        const float invr = 1.0f / dist;
        const F3 Ld = f3(Lv.x * invr, Lv.y * invr, Lv.z * invr);
        const float cosv = fmaxf(0.0f, dot3(N, Ld));
        if (cosv > 0.0f) {
            const Ray sr = makeRay(Po, Ld);
            if (COUNT) cntShadow++;
            const bool occluded = traceAny<COUNT, L>(nodes, tris, p.n_nodes, sr, 0.0f, dist, stack, static_cast<int>(p.tune_inner_min_any), iters, cntNodes, cntTris);
            if (!occluded) {
                const float k = (Lt.intensity / (kFourPi * r2)) * cosv;
                rgb.x = fmaf(albedo.x, k, rgb.x);
                rgb.y = fmaf(albedo.y, k, rgb.y);
                rgb.z = fmaf(albedo.z, k, rgb.z);
                if (PHONG && p.phong_ks > 0.0f) {
                    const float nl2 = 2.0f * dot3(N, Ld);
                    const F3 R = f3(fmaf(nl2, N.x, -Ld.x), fmaf(nl2, N.y, -Ld.y), fmaf(nl2, N.z, -Ld.z));
                    const float rv = fmaxf(0.0f, dot3(R, view));
                    const float sp = (p.phong_ks * (Lt.intensity / (kFourPi * r2))) * powUint(rv, p.phong_exp);
                    rgb.x += sp; rgb.y += sp; rgb.z += sp;
                }
            }
        }
    }
    return rgb;
}

// mode 100: Lambert (+ optional Phong highlight) + one shadow ray per light, every material treated as diffuse (oracle: shade_lambert)
template <bool COUNT, class L, bool PHONG>
__device__ __forceinline__ F3 shadeLambert(const RenderParams& p, const float4* nodes, const float4* tris, const Ray& r,
                                           const Hit& h, Stack& stack, uint32_t& iters, uint32_t& cntNodes, uint32_t& cntTris, uint32_t& cntShadow)
{
    const Surface sf = surfaceAt<L>(p, tris, r, h);
    return directLight<COUNT, L, PHONG>(p, nodes, tris, biasPoint(sf.P, sf.N, kShadowBias), sf.N, sf.albedo, f3(-r.d.x, -r.d.y, -r.d.z), stack, iters, cntNodes, cntTris, cntShadow);
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

} // namespace
} // namespace crt
