// Deterministic binned-SAH BVH2 builder (host).  See bvh_build.h and DESIGN.md "BVH build" for the spec;
// tests/test_bvh.py checks the result against the oracle's independent restatement byte for byte.
//
// Structure: triangle bounds/centroids in SoA arrays; one recursive routine per index range that bins the
// three axes in a single sweep; large ranges fork their left half as an OpenMP task (ranges are disjoint, so
// the result does not depend on scheduling); nodes are allocated from a shared pool in arbitrary order and
// renumbered to DFS pre-order at the end.
#include "mem_util.h"
#include "bvh_build.h"
#include "bvh_wide.h"
#include "bvh_pack.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <omp.h>
#include <sched.h>
#include <algorithm>
#include <limits>
#include <stdexcept>

namespace crt {
namespace {

inline float fmin_sel(float a, float b) { return a < b ? a : b; }
inline float fmax_sel(float a, float b) { return a > b ? a : b; }

struct Box {
    float mn[3], mx[3];
    void clear()
    {
        for (int a = 0; a < 3; a++) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const Box& o)
    {
        for (int a = 0; a < 3; a++) { mn[a] = fmin_sel(mn[a], o.mn[a]); mx[a] = fmax_sel(mx[a], o.mx[a]); }
    }
    float halfArea() const
    {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return (dx * dy + dy * dz) + dz * dx;
    }
};

// ranges at least this long run their sweeps in parallel chunks (Builder::build)
constexpr uint32_t kParallelSweep = 131072;

struct TempNode {
    Box lb, rb;
    int32_t left, right; // >= 0: pool index, < 0: final leaf reference
};

inline int32_t leafRef(uint32_t first, uint32_t count) { return ~static_cast<int32_t>((first << 3) | count); }

struct Builder {
    const Box* primBox;
    const float* cent; // 3 per triangle
    uint32_t* order;
    uint32_t* scratch;
    TempNode* pool;
    std::atomic<uint32_t> poolUsed{ 0 };
    std::atomic<uint32_t> deepest{ 0 };

    inline int binOf(uint32_t prim, int axis, float lo, float scale) const
    {
        const float f = (cent[3 * prim + axis] - lo) * scale; // NaN (triangle with a NaN vertex) -> bin 0, never out of range
        return f >= 0.0f ? (f < static_cast<float>(kBins) ? static_cast<int>(f) : kBins - 1) : 0;
    }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth, Box& bounds)
    {
        // Large ranges (the top levels of the tree, which are otherwise one thread's work while the others wait) run their
        // three sweeps chunk-parallel on the task team.  Everything merged across chunks is a min, a max, a count or a
        // stable concatenation, so the result is the sequential one bit for bit.
        const uint32_t nChunks = count >= kParallelSweep ? std::min<uint32_t>(64u, count / (kParallelSweep / 4u)) : 1u;
        const uint32_t chunkLen = (count + nChunks - 1u) / nChunks;
        Box cb;
        bounds.clear();
        cb.clear();
        auto sweepBounds = [&](uint32_t b, uint32_t e, Box& bb, Box& cc) {
            for (uint32_t i = b; i < e; i++) {
                const uint32_t p = order[i];
                bb.grow(primBox[p]);
                for (int a = 0; a < 3; a++) {
                    const float c = cent[3 * p + a];
                    cc.mn[a] = fmin_sel(cc.mn[a], c);
                    cc.mx[a] = fmax_sel(cc.mx[a], c);
                }
            }
        };
        if (nChunks == 1u) {
            sweepBounds(first, first + count, bounds, cb);
        } else {
            std::vector<Box> pb(nChunks), pc(nChunks);
#pragma omp taskloop grainsize(1) shared(pb, pc)
            for (uint32_t c = 0; c < nChunks; c++) {
                pb[c].clear();
                pc[c].clear();
                const uint32_t b = first + c * chunkLen, e = std::min(first + count, b + chunkLen);
                sweepBounds(b, e, pb[c], pc[c]);
            }
            for (uint32_t c = 0; c < nChunks; c++) {
                bounds.grow(pb[c]);
                cb.grow(pc[c]);
            }
        }
        uint32_t seen = deepest.load(std::memory_order_relaxed);
        while (depth > seen && !deepest.compare_exchange_weak(seen, depth, std::memory_order_relaxed)) {}
        if (count <= 1 || depth >= static_cast<uint32_t>(kMaxDepth)) return leafRef(first, count);

        // ---- bin all three axes in one sweep
        Box binBox[3][kBins];
        uint32_t binCnt[3][kBins];
        float scale[3];
        bool live[3];
        for (int a = 0; a < 3; a++) {
            const float ext = cb.mx[a] - cb.mn[a];
            live[a] = ext > 0.0f;
            scale[a] = live[a] ? static_cast<float>(kBins) / ext : 0.0f;
            for (int b = 0; b < kBins; b++) { binBox[a][b].clear(); binCnt[a][b] = 0; }
        }
        struct Bins { Box box[3][kBins]; uint32_t cnt[3][kBins]; };
        auto sweepBins = [&](uint32_t b, uint32_t e, Box (*bx)[kBins], uint32_t (*bc)[kBins]) {
            for (uint32_t i = b; i < e; i++) {
                const uint32_t p = order[i];
                for (int a = 0; a < 3; a++) {
                    if (!live[a]) continue;
                    const int bin = binOf(p, a, cb.mn[a], scale[a]);
                    bc[a][bin]++;
                    bx[a][bin].grow(primBox[p]);
                }
            }
        };
        if (nChunks == 1u) {
            sweepBins(first, first + count, binBox, binCnt);
        } else {
            std::vector<Bins> part(nChunks);
#pragma omp taskloop grainsize(1) shared(part)
            for (uint32_t c = 0; c < nChunks; c++) {
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < kBins; b++) { part[c].box[a][b].clear(); part[c].cnt[a][b] = 0; }
                const uint32_t b = first + c * chunkLen, e = std::min(first + count, b + chunkLen);
                sweepBins(b, e, part[c].box, part[c].cnt);
            }
            for (uint32_t c = 0; c < nChunks; c++)
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < kBins; b++) {
                        binBox[a][b].grow(part[c].box[a][b]);
                        binCnt[a][b] += part[c].cnt[a][b];
                    }
        }
        // ---- evaluate the 3 x 15 planes: axis ascending, plane ascending, first strict minimum wins
        float bestCost = std::numeric_limits<float>::infinity();
        int bestAxis = -1, bestPlane = 0;
        for (int a = 0; a < 3; a++) {
            if (!live[a]) continue;
            Box suffix[kBins];
            uint32_t suffixCnt[kBins];
            Box acc;
            acc.clear();
            uint32_t n = 0;
            for (int b = kBins - 1; b >= 1; b--) {
                acc.grow(binBox[a][b]);
                n += binCnt[a][b];
                suffix[b] = acc;
                suffixCnt[b] = n;
            }
            acc.clear();
            n = 0;
            for (int s = 1; s < kBins; s++) {
                acc.grow(binBox[a][s - 1]);
                n += binCnt[a][s - 1];
                if (n == 0 || suffixCnt[s] == 0) continue;
                const float cost = acc.halfArea() * static_cast<float>(n) + suffix[s].halfArea() * static_cast<float>(suffixCnt[s]);
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestPlane = s; }
            }
        }

        const float area = bounds.halfArea();
        if (count <= static_cast<uint32_t>(kLeafMax)) {
            if (bestAxis < 0 || !(kTravCost * area + bestCost < static_cast<float>(count) * area)) return leafRef(first, count);
        }

        uint32_t nLeft = 0;
        if (bestAxis >= 0) {
            uint32_t nl = 0, nr = 0;
            const float lo = cb.mn[bestAxis], sc = scale[bestAxis];
            if (nChunks == 1u) {
                for (uint32_t i = first; i < first + count; i++) { // stable partition
                    const uint32_t p = order[i];
                    if (binOf(p, bestAxis, lo, sc) < bestPlane) order[first + nl++] = p;
                    else scratch[first + nr++] = p;
                }
                crt::copyBytes(order + first + nl, scratch + first, sizeof(uint32_t) * nr);
            } else {
                // stable partition in three chunk-parallel steps: count the lefts of every chunk, place every element at its
                // final position in scratch (lefts of chunk c after the lefts of chunks < c, rights likewise), copy back
                std::vector<uint32_t> lefts(nChunks);
#pragma omp taskloop grainsize(1) shared(lefts)
                for (uint32_t c = 0; c < nChunks; c++) {
                    const uint32_t b = first + c * chunkLen, e = std::min(first + count, b + chunkLen);
                    uint32_t k = 0;
                    for (uint32_t i = b; i < e; i++) k += binOf(order[i], bestAxis, lo, sc) < bestPlane ? 1u : 0u;
                    lefts[c] = k;
                }
                std::vector<uint32_t> lOff(nChunks), rOff(nChunks);
                for (uint32_t c = 0; c < nChunks; c++) {
                    const uint32_t b = c * chunkLen, e = std::min(count, b + chunkLen);
                    lOff[c] = nl;
                    rOff[c] = nr;
                    nl += lefts[c];
                    nr += (e - b) - lefts[c];
                }
                const uint32_t totalLeft = nl;
#pragma omp taskloop grainsize(1) shared(lOff, rOff)
                for (uint32_t c = 0; c < nChunks; c++) {
                    const uint32_t b = first + c * chunkLen, e = std::min(first + count, b + chunkLen);
                    uint32_t l = first + lOff[c], r = first + totalLeft + rOff[c];
                    for (uint32_t i = b; i < e; i++) {
                        const uint32_t p = order[i];
                        if (binOf(p, bestAxis, lo, sc) < bestPlane) scratch[l++] = p;
                        else scratch[r++] = p;
                    }
                }
#pragma omp taskloop grainsize(1)
                for (uint32_t c = 0; c < nChunks; c++) {
                    const uint32_t b = first + c * chunkLen, e = std::min(first + count, b + chunkLen);
                    crt::copyBytes(order + b, scratch + b, sizeof(uint32_t) * (e - b));
                }
            }
            nLeft = nl;
            const uint64_t cap = static_cast<uint64_t>(kLeafMax) << (kMaxDepth - depth - 1);
            if (static_cast<uint64_t>(nl > nr ? nl : nr) > cap) nLeft = 0;
        }
        if (nLeft == 0) nLeft = count / 2;

        const uint32_t me = poolUsed.fetch_add(1, std::memory_order_relaxed);
        TempNode& N = pool[me];
        const uint32_t nRight = count - nLeft;
        if (count >= 32768) {
            int32_t l = 0;
#pragma omp task shared(l, N) firstprivate(first, nLeft, depth)
            l = build(first, nLeft, depth + 1, N.lb);
            const int32_t r = build(first + nLeft, nRight, depth + 1, N.rb);
#pragma omp taskwait
            N.left = l;
            N.right = r;
        } else {
            N.left = build(first, nLeft, depth + 1, N.lb);
            N.right = build(first + nLeft, nRight, depth + 1, N.rb);
        }
        return static_cast<int32_t>(me);
    }
};

void writeNode(crt_bvh_node& d, const TempNode& s)
{
    d.lx0 = s.lb.mn[0]; d.lx1 = s.lb.mx[0]; d.ly0 = s.lb.mn[1]; d.ly1 = s.lb.mx[1]; d.lz0 = s.lb.mn[2]; d.lz1 = s.lb.mx[2];
    d.rx0 = s.rb.mn[0]; d.rx1 = s.rb.mx[0]; d.ry0 = s.rb.mn[1]; d.ry1 = s.rb.mx[1]; d.rz0 = s.rb.mn[2]; d.rz1 = s.rb.mx[2];
    d.pad0 = 0;
    d.pad1 = 0;
}

} // namespace

// Flatten the meshes in InstanceID order: triangle records {v0, e1, e2, ids}, shading records, and per triangle its
// bounds (6 floats: min xyz, max xyz) and box centroid (3 floats) -- 9 floats per triangle in boxCent.
// Threads this process can really use: OpenMP's default is the machine's CPU count, but a container often grants a share
// (cgroup v2 cpu.max) -- 256 threads on a 16-CPU share made the 1M-triangle build 20-40 % slower and erratic.
int usableThreads()
{
    static const int cached = [] {
        int n = omp_get_max_threads();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, CPU_COUNT(&set));
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = { 0 };
            long long period = 0;
            if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
                n = std::min(n, std::max(1, static_cast<int>((std::atof(quota) / static_cast<double>(period)) + 0.5)));
            std::fclose(f);
        }
        return std::max(1, n);
    }();
    return cached;
}

void flattenMeshes(const crt_mesh_view* meshes, uint32_t n_meshes, std::vector<crt_bvh_tri>& inTri, std::vector<crt_bvh_shade>& inShade,
                   std::vector<float>& boxCent)
{
    uint64_t total = 0;
    for (uint32_t m = 0; m < n_meshes; m++) {
        if (meshes[m].n_triangles && (!meshes[m].xyz || !meshes[m].idx)) throw std::runtime_error("mesh with triangles but null vertex/index pointer");
        total += meshes[m].n_triangles;
    }
    if (total >= (1ull << 28)) throw std::runtime_error("too many triangles (limit 2^28 - 1)");
    const uint32_t n = static_cast<uint32_t>(total);
    inTri.resize(n);
    inShade.resize(n);
    boxCent.resize(9 * static_cast<size_t>(n));
    uint32_t g0 = 0;
    bool bad = false; // an exception must not leave the parallel region: remember, finish, throw afterwards
    for (uint32_t m = 0; m < n_meshes; m++) {
        const crt_mesh_view& M = meshes[m];
        const long long nt = static_cast<long long>(M.n_triangles);
#pragma omp parallel for schedule(static) if (nt > 65536) reduction(|| : bad) num_threads(usableThreads())
        for (long long tt = 0; tt < nt; tt++) {
            const uint32_t t = static_cast<uint32_t>(tt), g = g0 + t;
            const uint32_t i0 = M.idx[3 * t], i1 = M.idx[3 * t + 1], i2 = M.idx[3 * t + 2];
            if (i0 >= M.n_vertices || i1 >= M.n_vertices || i2 >= M.n_vertices) {
                bad = true;
                continue;
            }
            const float* A = M.xyz + 3 * static_cast<size_t>(i0);
            const float* B = M.xyz + 3 * static_cast<size_t>(i1);
            const float* C = M.xyz + 3 * static_cast<size_t>(i2);
            crt_bvh_tri& T = inTri[g];
            float* bc = &boxCent[9 * static_cast<size_t>(g)];
            for (int k = 0; k < 3; k++) {
                T.v0[k] = A[k];
                T.e1[k] = B[k] - A[k];
                T.e2[k] = C[k] - A[k];
                bc[k] = fmin_sel(fmin_sel(A[k], B[k]), C[k]);
                bc[3 + k] = fmax_sel(fmax_sel(A[k], B[k]), C[k]);
                bc[6 + k] = (bc[k] + bc[3 + k]) * 0.5f;
            }
            T.inst = m;
            T.prim = t;
            T.gid = g;
            crt_bvh_shade& S = inShade[g];
            std::memset(&S, 0, sizeof(S));
            S.material = static_cast<uint32_t>(M.material_index);
            if (M.normals) {
                crt::copyBytes(S.n0, M.normals + 3 * static_cast<size_t>(i0), 12);
                crt::copyBytes(S.n1, M.normals + 3 * static_cast<size_t>(i1), 12);
                crt::copyBytes(S.n2, M.normals + 3 * static_cast<size_t>(i2), 12);
            }
        }
        if (bad) throw std::runtime_error("triangle index out of range");
        g0 += M.n_triangles;
    }
}

void flattenUvs(const crt_mesh_view* meshes, uint32_t n_meshes, std::vector<crt_bvh_uv>& inUv)
{
    inUv.clear();
    bool any = false;
    uint64_t total = 0;
    for (uint32_t m = 0; m < n_meshes; m++) {
        any |= meshes[m].uvs != nullptr && meshes[m].n_triangles > 0;
        total += meshes[m].n_triangles;
    }
    if (!any) return;
    inUv.assign(total, crt_bvh_uv{});
    uint32_t g = 0;
    for (uint32_t m = 0; m < n_meshes; m++) {
        const crt_mesh_view& M = meshes[m];
        for (uint32_t t = 0; t < M.n_triangles; t++, g++) {
            if (!M.uvs) continue;
            crt::copyBytes(inUv[g].uv0, M.uvs + 3 * static_cast<size_t>(M.idx[3 * t]), 8);
            crt::copyBytes(inUv[g].uv1, M.uvs + 3 * static_cast<size_t>(M.idx[3 * t + 1]), 8);
            crt::copyBytes(inUv[g].uv2, M.uvs + 3 * static_cast<size_t>(M.idx[3 * t + 2]), 8);
        }
    }
}

void reorderUvs(const std::vector<crt_bvh_uv>& inUv, Bvh& bvh)
{
    bvh.uvs.clear();
    if (inUv.empty()) return;
    bvh.uvs.resize(bvh.tris.size());
    for (size_t i = 0; i < bvh.tris.size(); i++) bvh.uvs[i] = inUv[bvh.tris[i].gid]; // gid = input ordinal
}

void buildBvh(const crt_mesh_view* meshes, uint32_t n_meshes, Bvh& out, int width)
{
    const bool timing = std::getenv("CRT_BUILD_TIMING") != nullptr; // phase breakdown on stderr
    auto tlast = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[sah build] %-24s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tlast).count());
        tlast = t;
    };
    std::vector<crt_bvh_tri> inTri;
    std::vector<crt_bvh_shade> inShade;
    std::vector<float> boxCent;
    flattenMeshes(meshes, n_meshes, inTri, inShade, boxCent);
    const uint32_t n = static_cast<uint32_t>(inTri.size());
    lap("flatten");

    out.nodes.clear();
    out.nodes4.clear();
    out.depth4 = 0;
    out.tris.clear();
    out.shade.clear();
    out.maxDepth = 0;
    out.uvs.clear();
    out.nTris = 0;
    out.devTris = out.devShade = out.devUvs = nullptr;
    out.devNodes = out.devNodes4 = out.devNodes4q = nullptr;
    out.nNodes = out.nNodes4 = 0;
    out.width = static_cast<uint32_t>(width);
    out.packed.clear();
    out.packedGranules = out.nWide = out.depthWide = 0;
    out.devPacked = nullptr;
    if (n == 0) return;

    std::vector<Box> primBox(n);
    std::vector<float> cent(3 * static_cast<size_t>(n));
    for (uint32_t i = 0; i < n; i++) {
        crt::copyBytes(primBox[i].mn, &boxCent[9 * static_cast<size_t>(i)], 12);
        crt::copyBytes(primBox[i].mx, &boxCent[9 * static_cast<size_t>(i) + 3], 12);
        crt::copyBytes(&cent[3 * static_cast<size_t>(i)], &boxCent[9 * static_cast<size_t>(i) + 6], 12);
    }

    std::vector<uint32_t> order(n), scratch(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::vector<TempNode> pool(n); // a binary tree over n leaves has < n inner nodes

    Builder B;
    B.primBox = primBox.data();
    B.cent = cent.data();
    B.order = order.data();
    B.scratch = scratch.data();
    B.pool = pool.data();
    Box rootBox;
    int32_t root = 0;
    lap("setup arrays");
#pragma omp parallel num_threads(usableThreads())
#pragma omp single
    root = B.build(0, n, 0, rootBox);
    out.maxDepth = B.deepest.load();
    lap("recursive build");

    if (root < 0) {
        // the whole scene fits one leaf: node 0 must exist, so wrap it; the right child is an empty leaf
        // with the same box (an empty box would be 'hit' by the slab test's inf arithmetic)
        TempNode t;
        t.lb = rootBox;
        t.rb = rootBox;
        out.nodes.resize(1);
        writeNode(out.nodes[0], t);
        out.nodes[0].left = root;
        out.nodes[0].right = leafRef(0, 0);
    } else {
        // ---- renumber to DFS pre-order (left subtree first)
        const uint32_t used = B.poolUsed.load();
        out.nodes.resize(used);
        std::vector<int32_t> newIndex(used, -1);
        std::vector<int32_t> stack;
        stack.push_back(root);
        uint32_t next = 0;
        while (!stack.empty()) {
            const int32_t t = stack.back();
            stack.pop_back();
            newIndex[t] = static_cast<int32_t>(next++);
            if (pool[t].right >= 0) stack.push_back(pool[t].right);
            if (pool[t].left >= 0) stack.push_back(pool[t].left);
        }
        for (uint32_t t = 0; t < used; t++) {
            crt_bvh_node& d = out.nodes[newIndex[t]];
            writeNode(d, pool[t]);
            d.left = pool[t].left >= 0 ? newIndex[pool[t].left] : pool[t].left;
            d.right = pool[t].right >= 0 ? newIndex[pool[t].right] : pool[t].right;
        }
    }

    lap("renumber");
    out.tris.resize(n);
    out.shade.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        out.tris[i] = inTri[order[i]];
        out.shade[i] = inShade[order[i]];
    }
    lap("reorder records");
    out.nTris = static_cast<uint32_t>(out.tris.size());
    std::vector<crt_bvh_uv> inUv;
    flattenUvs(meshes, n_meshes, inUv);
    if (width == 0) {
        collapseBvh4(out);
        lap("collapse");
        reorderUvs(inUv, out);
    } else {
        packBvh(out, width);
        lap("collapse + pack");
        out.nNodes = static_cast<uint32_t>(out.nodes.size());
        out.shade = inShade; // the packed tree's triangle records carry their gid: shading records and uvs stay in input order
        out.uvs = inUv;
    }
}

namespace {
// Binary -> W-wide collapse (the rule of bvh_wide.h wideSlotsT, DFS pre-order) followed by the packing of bvh_pack.h: child
// blocks in wide-node order behind the root's record; every block holds its node's children in slot order.
template <int W>
void packBvhT(Bvh& bvh)
{
    typedef PackFmt<W> F;
    std::vector<WideNodeT<W>> wide;
    wide.reserve(bvh.nodes.size() / (W / 2) + 1);
    bvh.depthWide = 0;
    struct Work { int32_t binary; int32_t parent; int slot; uint32_t depth; };
    std::vector<Work> work;
    work.push_back({ 0, -1, 0, 1 });
    while (!work.empty()) {
        const Work w = work.back();
        work.pop_back();
        WideSlot sl[W];
        const int n = wideSlotsT<W>(bvh.nodes.data(), w.binary, 0u, sl, nullptr);
        const int32_t me = static_cast<int32_t>(wide.size());
        wide.emplace_back();
        fillWideT<W>(sl, n, wide.back());
        if (w.parent >= 0) wide[w.parent].ref[w.slot] = me;
        if (w.depth > bvh.depthWide) bvh.depthWide = w.depth;
        for (int i = n - 1; i >= 0; i--)
            if (sl[i].ref >= 0) work.push_back({ sl[i].ref, me, i, w.depth + 1 });
    }
    const size_t n = wide.size();
    // addresses: the root's record at 0, then the child blocks in node order; a node's own record lies in its parent's block
    std::vector<uint32_t> addr(n), base(n);
    uint64_t next = F::kNodeGranules;
    for (size_t i = 0; i < n; i++) {
        base[i] = static_cast<uint32_t>(next);
        next += blockGranules<W>(wide[i]);
        if (next >= (1ull << 29)) throw std::runtime_error("scene too large for the packed tree's 29-bit addresses");
    }
    addr[0] = 0;
    for (size_t i = 0; i < n; i++) { // parents precede their children in pre-order
        uint32_t off = 0;
        for (int k = 0; k < W; k++) {
            const int32_t ref = wide[i].ref[k];
            if (ref >= 0) addr[ref] = base[i] + off;
            else if ((static_cast<uint32_t>(~ref) & 7u) > kPackLeafMax) throw std::runtime_error("leaf too large for the packed tree");
            off += childGranules<W>(ref);
        }
    }
    const size_t dwordsPerGranule = F::kGranuleBytes / 4;
    bvh.packed.assign(static_cast<size_t>(next) * dwordsPerGranule, 0u);
    bvh.packedGranules = static_cast<uint32_t>(next);
    bvh.nWide = static_cast<uint32_t>(n);
    const int64_t nn = static_cast<int64_t>(n);
#pragma omp parallel for schedule(static) if (nn > 65536)
    for (int64_t i = 0; i < nn; i++) {
        const WideNodeT<W>& N = wide[static_cast<size_t>(i)];
        packNode<W>(N, base[i], &bvh.packed[static_cast<size_t>(addr[i]) * dwordsPerGranule]);
        uint32_t off = 0;
        for (int k = 0; k < W; k++) {
            const int32_t ref = N.ref[k];
            if (ref < 0) {
                const uint32_t code = static_cast<uint32_t>(~ref), first = code >> 3, cnt = code & 7u;
                if (cnt) crt::copyBytes(&bvh.packed[static_cast<size_t>(base[i] + off) * dwordsPerGranule], &bvh.tris[first], sizeof(crt_bvh_tri) * cnt);
            }
            off += childGranules<W>(ref);
        }
    }
}
} // namespace

void packBvh(Bvh& bvh, int width)
{
    bvh.packed.clear();
    bvh.packedGranules = bvh.nWide = bvh.depthWide = 0;
    bvh.width = static_cast<uint32_t>(width);
    if (bvh.nodes.empty()) return;
    if (width == 4) packBvhT<4>(bvh);
    else if (width == 8) packBvhT<8>(bvh);
    else throw std::runtime_error("packed tree width must be 4 or 8");
}

// Binary -> 4-wide collapse.  A wide node takes the two children of a binary node and, while it has fewer than four
// slots, replaces the inner slot of largest half-area (first on ties) by that node's two children, in place (left keeps
// the position, right goes right after it).  Wide nodes are numbered in DFS pre-order, children in slot order.
// Iterative formulation (explicit work stack, pre-order ids patched into the parent when a child is emitted).
void collapseBvh4(Bvh& bvh)
{
    bvh.nodes4.clear();
    bvh.nodes4q.clear();
    bvh.depth4 = 0;
    if (bvh.nodes.empty()) return;
    struct Work { int32_t binary; int32_t parent; int slot; uint32_t depth; }; // parent wide node / slot to patch
    std::vector<Work> work;
    work.push_back({ 0, -1, 0, 1 });
    bvh.nodes4.reserve(bvh.nodes.size() / 2 + 1);
    while (!work.empty()) {
        const Work w = work.back();
        work.pop_back();
        WideSlot sl[4];
        const int n = wideSlots(bvh.nodes.data(), w.binary, 0u, sl, nullptr); // the rule itself: bvh_wide.h
        const int32_t me = static_cast<int32_t>(bvh.nodes4.size());
        bvh.nodes4.emplace_back();
        fillWide(sl, n, bvh.nodes4.back()); // inner refs are binary indices until the child is emitted
        if (w.parent >= 0) bvh.nodes4[w.parent].ref[w.slot] = me;
        if (w.depth > bvh.depth4) bvh.depth4 = w.depth;
        // children must come out in slot order right after this node (pre-order): push them in reverse
        for (int i = n - 1; i >= 0; i--)
            if (sl[i].ref >= 0) work.push_back({ sl[i].ref, me, i, w.depth + 1 });
    }
    bvh.nNodes = static_cast<uint32_t>(bvh.nodes.size());
    bvh.nNodes4 = static_cast<uint32_t>(bvh.nodes4.size());
    bvh.nodes4q.resize(bvh.nodes4.size());
    const int64_t nq = static_cast<int64_t>(bvh.nodes4.size());
#pragma omp parallel for schedule(static) if (nq > 65536)
    for (int64_t i = 0; i < nq; i++) quantizeNode4(bvh.nodes4[static_cast<size_t>(i)], bvh.nodes4q[static_cast<size_t>(i)]);
}

} // namespace crt
