// Host side of librt_amd.so: BVH construction in the reference's layout, flattening into the GPU
// layout, camera set-up.  Pure CPU code; runs once per scene, never per ray.
#include "rt_types.h"
#include "rt_tri.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <utility>

namespace rt {

static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// BVH build: median split on the longest axis, leaves of at most 4 triangles
// (reference src/bvh.cpp:48-95, 111-214, 381-512).
//
// The reference sorts 100-byte primitives with a comparator that recomputes both centroids; here
// each level sorts 8-byte {key, index} records instead.  std::sort's sequence of comparisons and
// moves depends only on the comparator's answers, which are the same floats, so the resulting
// permutation - including the order among equal keys, which std::sort leaves unspecified but
// deterministic - is the one the reference's build produces with the same standard library.
// ---------------------------------------------------------------------------------------------
namespace {

struct Builder {
    const RefPrimitive* in;
    std::vector<vec3> lo, hi, centroid; // per input primitive
    std::vector<uint32_t> order;        // working permutation of input indices
    std::vector<std::pair<float, uint32_t>> keys;
    rt_bvh* out;
    bool sahBinning = false; // features.extra.enableBvhSahBinning (src/common.h:63)
    bool undefined = false;  // a span's centroids coincide on its split axis: the reference's bin index is 0/0 there
    std::vector<float> leftArea; // surface area of the box around the first k primitives of the sorted span
    std::vector<uint8_t> filled; // device build: primitives whose lo / hi / centroid entries are in place

    static float surfaceArea(vec3 lower, vec3 upper) // calculateAABBSurfaceArea (src/extra.cpp:609-613)
    {
        const float x = fabsf(upper.x - lower.x), y = fabsf(upper.y - lower.y), z = fabsf(upper.z - lower.z);
        return 2.0f * (x * y + x * z + y * z);
    }

    // splitPrimitivesBySAHBin (src/extra.cpp:623-699) on a span already sorted along the axis (keys[] holds its
    // centroid coordinates in order).  The reference files the primitives into up to 50 bins of equal width over
    // the centroid interval and prices a split after every bin as 1.5 + (area x count, left and right) / area of
    // the span's box, recomputing both boxes each time; here the boxes grow by one primitive per step from either
    // end (min / max are exact, so the areas are the same floats).  Returns 0 for "no split: leaf".
    size_t sahSplit(const uint32_t* idx, size_t n, const RefNode& node)
    {
        const size_t nBins = n < 50 ? n : 50;
        const float first = keys[0].first, len = keys[n - 1].first - first;
        leftArea.resize(n + 1);
        vec3 lower = splat(FLT_MAX), upper = splat(-FLT_MAX);
        for (size_t k = 0; k < n; k++) {
            lower = gmin(lower, lo[idx[k]]);
            upper = gmax(upper, hi[idx[k]]);
            leftArea[k + 1] = surfaceArea(lower, upper);
        }
        const float outer = surfaceArea(load3(node.lower), load3(node.upper));
        // walk the candidate splits from the right so that the right-hand box grows with them, keep the cheapest;
        // among equal costs the reference keeps the leftmost split (strict < while scanning left to right)
        float minCost = FLT_MAX;
        size_t minIndex = 0;
        lower = splat(FLT_MAX), upper = splat(-FLT_MAX);
        size_t covered = n; // primitives [covered, n) are inside the right-hand box
        size_t k = n;       // primitives [k, n) are known to lie in bins above the current one
        for (size_t bin = nBins - 1; bin >= 1; bin--) {
            // leftSize of the split after bin-1 = number of primitives in bins < bin
            while (k > 0) {
                const float q = float(nBins) * (keys[k - 1].first - first) / len; // determineBucketIndex (src/extra.cpp:582-607)
                if (!(q >= 0.0f && q < 1.0e9f)) {
                    undefined = true;
                    return 0;
                }
                size_t b = size_t(q);
                if (b == nBins) b--;
                if (b < bin) break;
                k--;
            }
            const size_t leftSize = k;
            while (covered > leftSize) {
                covered--;
                lower = gmin(lower, lo[idx[covered]]);
                upper = gmax(upper, hi[idx[covered]]);
            }
            const float cost = 1.5f + (leftArea[leftSize] * float(leftSize) + surfaceArea(lower, upper) * float(n - leftSize)) / outer;
            if (cost <= minCost) { // <= while scanning right to left: the leftmost of equal costs wins; NaN never does
                minCost = cost;
                minIndex = leftSize;
            }
        }
        if (minCost >= float(n)) return 0;
        return minIndex;
    }

    void emitLeafCopy(const uint32_t* idx, size_t n)
    {
        for (size_t i = 0; i < n; i++) {
            out->prims.push_back(in[idx[i]]);
            out->primIds.push_back(idx[i]);
        }
    }

    void build(uint32_t* idx, size_t n, uint32_t nodeIndex)
    {
        // computeSpanAABB (src/bvh.cpp:125-137); an empty span keeps the default box [0,1]^3
        RefNode node;
        if (n < 1) {
            node.lower[0] = node.lower[1] = node.lower[2] = 0.0f;
            node.upper[0] = node.upper[1] = node.upper[2] = 1.0f;
        } else {
            vec3 lower = splat(FLT_MAX), upper = splat(-FLT_MAX);
            for (size_t i = 0; i < n; i++) {
                lower = gmin(lower, lo[idx[i]]);
                upper = gmax(upper, hi[idx[i]]);
            }
            node.lower[0] = lower.x, node.lower[1] = lower.y, node.lower[2] = lower.z;
            node.upper[0] = upper.x, node.upper[1] = upper.y, node.upper[2] = upper.z;
        }
        auto leaf = [&]() {
            // The reference appends the leaf's primitives, records the offset AFTER that append, then
            // appends them again (src/bvh.cpp:387-392, 458 / 492-493): primitives() holds every leaf twice and
            // the recorded offset addresses the second copy.  Kept so nodes()/primitives() read the same.
            emitLeafCopy(idx, n);
            node.data[0] = uint32_t(out->prims.size()) | kLeafBit;
            node.data[1] = uint32_t(n);
            out->nodes[nodeIndex] = node;
            emitLeafCopy(idx, n);
        };
        if (n <= 4) {
            leaf();
            return;
        }
        // computeAABBLongestAxis (src/bvh.cpp:160-169): ties go to the lower axis index
        const float ax = fabsf(node.upper[0] - node.lower[0]), ay = fabsf(node.upper[1] - node.lower[1]),
                    az = fabsf(node.upper[2] - node.lower[2]);
        const int axis = (ax >= ay && ax >= az) ? 0 : ((ay >= ax && ay >= az) ? 1 : 2);
        // splitPrimitivesByMedian (src/bvh.cpp:181-214); the SAH split sorts the same way first (src/extra.cpp:643)
        keys.resize(n);
        for (size_t i = 0; i < n; i++) {
            const vec3 c = centroid[idx[i]];
            keys[i] = { axis == 0 ? c.x : (axis == 1 ? c.y : c.z), idx[i] };
        }
        std::sort(keys.begin(), keys.end(),
                  [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first < b.first; });
        for (size_t i = 0; i < n; i++) idx[i] = keys[i].second;
        size_t split = (n + 1) / 2;
        if (sahBinning) { // src/bvh.cpp:472-495: a span the heuristic does not split becomes a leaf of any size
            split = sahSplit(idx, n, node);
            if (split == 0) {
                leaf();
                return;
            }
        }
        const uint32_t left = uint32_t(out->nodes.size());
        out->nodes.emplace_back();
        const uint32_t right = uint32_t(out->nodes.size());
        out->nodes.emplace_back();
        node.data[0] = left;
        node.data[1] = right;
        out->nodes[nodeIndex] = node;
        build(idx, split, left);
        build(idx + split, n - split, right);
    }
};

void countLevelsAndLeaves(rt_bvh& b) // buildNumLevels / buildNumLeaves (src/bvh.cpp:517-577)
{
    std::vector<uint32_t> level { 0u }, next;
    uint32_t levels = 0, leaves = 0;
    while (!level.empty()) {
        next.clear();
        for (uint32_t i : level) {
            const RefNode& n = b.nodes[i];
            if (n.data[0] & kLeafBit) leaves++;
            else {
                next.push_back(n.data[0]);
                next.push_back(n.data[1]);
            }
        }
        levels++;
        level.swap(next);
    }
    b.numLevels = levels;
    b.numLeaves = leaves;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// Flattening: reference nodes()/primitives() -> NodePair / TriPlane / TriEdges / TriAttr
// ---------------------------------------------------------------------------------------------
static void precomputeTriangle(const RefPrimitive& p, uint32_t primId, TriPlane& pl, TriEdges& ed, TriAttr& at) { precompute_triangle(p, primId, pl, ed, at); }

namespace {
struct Flattener {
    const RefNode* nodes;
    uint32_t nNodes;
    const RefPrimitive* prims;
    uint32_t nPrims;
    const uint32_t* primIds;
    uint32_t nMaterials;
    FlatScene& out;
    char* msg;
    size_t msgLen;
    int status = RT_OK;
    std::vector<uint8_t> seen;
    std::vector<std::pair<uint32_t, uint32_t>> arrayPos; // (position in primitives(), compact index)
    std::vector<uint8_t> leafOf;                         // one entry per compact triangle met so far

    uint32_t fail(int code)
    {
        status = code;
        return kLeafBit;
    }
    // Child reference for node i; lays out NodePairs in pre-order and triangles in left-to-right
    // leaf order.  Recursion is bounded by the depth limit checked on entry.
    uint32_t visit(uint32_t i, uint32_t depth, uint32_t route)
    {
        if (status != RT_OK) return kLeafBit;
        if (i >= nNodes) {
            snprintf(msg, msgLen, "flatten: child index %u out of range (%u nodes)", i, nNodes);
            return fail(RT_ERR_INVALID);
        }
        if (depth > uint32_t(kStackDepth)) {
            snprintf(msg, msgLen, "flatten: tree deeper than %d levels (traversal stack limit)", kStackDepth);
            return fail(RT_ERR_LIMIT);
        }
        if (seen[i]) {
            snprintf(msg, msgLen, "flatten: node %u is reachable twice (not a tree)", i);
            return fail(RT_ERR_INVALID);
        }
        seen[i] = 1;
        if (depth > out.maxDepth) out.maxDepth = depth;
        const RefNode& n = nodes[i];
        for (int k = 0; k < 3; k++) {
            const float c[2] = { n.lower[k], n.upper[k] };
            for (float x : c) {
                const float ax = fabsf(x);
                if (!(x == 0.0f || (ax >= 8.6736174e-19f && ax <= 1073741824.0f))) out.fastBounds = false; // outside {0} U [2^-60, 2^30]
            }
        }
        if (n.data[0] & kLeafBit) {
            const uint32_t off = n.data[0] & ~kLeafBit, cnt = n.data[1];
            if (uint64_t(off) + cnt > nPrims || (cnt && !prims)) {
                snprintf(msg, msgLen, "flatten: leaf %u range [%u,%u) exceeds %u primitives", i, off, off + cnt, nPrims);
                return fail(RT_ERR_INVALID);
            }
            const uint32_t first = uint32_t(leafOf.size());
            if (uint64_t(first) + cnt > kRefOffsetMask) {
                snprintf(msg, msgLen, "flatten: more than %u triangles", kRefOffsetMask);
                return fail(RT_ERR_LIMIT);
            }
            // (the walk does not touch the primitives themselves - 100 bytes apart, a cache miss each: their mesh indices are
            // checked with the per-triangle work that follows the walk, on several threads)
            const float box[6] = { n.lower[0], n.lower[1], n.lower[2], n.upper[0], n.upper[1], n.upper[2] };
            for (uint32_t k = 0; k < cnt; k++) {
                arrayPos.push_back({ off + k, uint32_t(leafOf.size()) });
                leafOf.push_back(0); // (one entry per compact triangle: the walk only counts them; their records are computed afterwards)
                out.leafBox.insert(out.leafBox.end(), box, box + 6);
                out.leafRoute.push_back(route);
            }
            if (cnt >= kMaxLeafCount) { // (the SAH-binned build leaves spans it cannot split cheaply whole: hundreds of triangles in one leaf)
                if (out.leafCounts.empty()) out.leafCounts.assign(nPrims, 0u);
                out.leafCounts[first] = cnt;
            }
            return kLeafBit | (std::min(cnt, kMaxLeafCount) << kRefCountShift) | first;
        }
        const uint32_t l = n.data[0], r = n.data[1];
        if (l >= nNodes || r >= nNodes) {
            snprintf(msg, msgLen, "flatten: node %u has a child index out of range", i);
            return fail(RT_ERR_INVALID);
        }
        for (uint32_t c : { l, r })
            for (int a = 0; a < 3; a++)
                if (!(nodes[c].lower[a] >= n.lower[a] && nodes[c].upper[a] <= n.upper[a])) out.nested = false;
        const uint32_t pair = uint32_t(out.pairs.size());
        out.pairs.emplace_back();
        NodePair np;
        memcpy(np.lo0, nodes[l].lower, 12);
        memcpy(np.hi0, nodes[l].upper, 12);
        memcpy(np.lo1, nodes[r].lower, 12);
        memcpy(np.hi1, nodes[r].upper, 12);
        np.pad0 = np.pad1 = 0;
        np.ref0 = visit(l, depth + 1, route * 2u); // depth <= 32: the marker bit stays inside 32 bits
        np.ref1 = visit(r, depth + 1, route * 2u + 1u);
        out.pairs[pair] = np;
        return pair;
    }
};
} // namespace

int flatten(const RefNode* nodes, uint32_t nNodes, const RefPrimitive* prims, uint32_t nPrims, const uint32_t* primIds,
            uint32_t nMaterials, FlatScene& out, const char** err, bool records)
{
    static thread_local char msg[256];
    msg[0] = 0;
    *err = msg;
    if (!nodes || nNodes == 0) {
        snprintf(msg, sizeof(msg), "flatten: empty node array");
        return RT_ERR_INVALID;
    }
    out = FlatScene {};
    Flattener f { nodes, nNodes, prims, nPrims, primIds, nMaterials, out, msg, sizeof(msg) };
    f.seen.assign(nNodes, 0);
    {
        const size_t guess = nPrims / 2 + 16; // (the reference stores every triangle twice; a caller's tree may not)
        out.leafRoute.reserve(guess);
        out.leafBox.reserve(6 * guess), out.pairs.reserve(nNodes / 2 + 1), f.arrayPos.reserve(guess);
    }
    memcpy(out.rootLo, nodes[0].lower, 12);
    memcpy(out.rootHi, nodes[0].upper, 12);
    const bool verbose = getenv("RT_AMD_VERBOSE") != nullptr;
    const auto tWalk = std::chrono::steady_clock::now();
    out.rootRef = f.visit(0, 1, 1u);
    if (f.status != RT_OK) return f.status;
    const auto tRecords = std::chrono::steady_clock::now();
    { // where compact triangle i lies in primitives(), and the index it reports
        const size_t n = f.arrayPos.size();
        out.srcPos.resize(n), out.srcPrimId.resize(n);
        for (size_t i = 0; i < n; i++) out.srcPos[i] = f.arrayPos[i].first, out.srcPrimId[i] = primIds ? primIds[f.arrayPos[i].first] : f.arrayPos[i].first;
    }
    { // the triangles' records (when asked for) and whether every vertex lies in the root box: independent of one another
        const size_t n = f.arrayPos.size();
        if (records) out.triPlane.resize(n), out.triEdges.resize(n), out.triAttr.resize(n);
        std::atomic<bool> inside { true }; // every vertex lies in the root box (part of `nested`)
        std::atomic<uint32_t> badMesh { 0xffffffffu }; // the first primitive (in array order) whose mesh has no material
        auto fill = [&](size_t lo, size_t hi) {
            bool in = true;
            for (size_t i = lo; i < hi; i++) {
                const uint32_t pos = f.arrayPos[i].first;
                const RefPrimitive& p = prims[pos];
                if (p.meshID >= nMaterials) {
                    uint32_t seen = badMesh.load();
                    while (pos < seen && !badMesh.compare_exchange_weak(seen, pos)) {}
                    continue;
                }
                for (const RefVertex* v : { &p.v0, &p.v1, &p.v2 })
                    for (int a = 0; a < 3; a++)
                        if (!(v->position[a] >= nodes[0].lower[a] && v->position[a] <= nodes[0].upper[a])) in = false;
                if (records) precomputeTriangle(p, out.srcPrimId[i], out.triPlane[i], out.triEdges[i], out.triAttr[i]);
            }
            if (!in) inside.store(false);
        };
        const unsigned hw = std::thread::hardware_concurrency();
        size_t workers = n < 16384 ? 1 : std::min<size_t>(8, hw ? hw : 1);
        if (const char* e = getenv("RT_AMD_FLATTEN_THREADS")) workers = std::max(1, atoi(e));
        if (workers <= 1) {
            fill(0, n);
        } else {
            std::vector<std::thread> pool;
            const size_t share = (n + workers - 1) / workers;
            for (size_t k = 1; k < workers; k++) pool.emplace_back(fill, std::min(n, k * share), std::min(n, (k + 1) * share));
            fill(0, std::min(n, share));
            for (std::thread& t : pool) t.join();
        }
        if (!inside.load()) out.nested = false;
        if (badMesh.load() != 0xffffffffu) {
            const uint32_t pos = badMesh.load();
            snprintf(msg, sizeof(msg), "flatten: primitive %u has meshID %u but only %u materials", pos, prims[pos].meshID, nMaterials);
            return RT_ERR_INVALID;
        }
    }
    const auto tOrder = std::chrono::steady_clock::now();
    // Brute-force order (enableAccelStructure off, src/bvh.cpp:351-363) = primitives() array order.
    // (every position occurs at most once - a leaf range met twice means a node met twice, which visit refuses - so the sort is a scatter)
    std::vector<uint32_t> atPos(nPrims, 0xffffffffu);
    bool unique = true;
    for (const auto& ap : f.arrayPos) {
        if (atPos[ap.first] != 0xffffffffu) unique = false;
        atPos[ap.first] = ap.second;
    }
    out.bruteOrder.clear();
    out.bruteOrder.reserve(f.arrayPos.size());
    if (unique) {
        for (uint32_t p = 0; p < nPrims; p++)
            if (atPos[p] != 0xffffffffu) out.bruteOrder.push_back(atPos[p]);
    } else { // overlapping leaf ranges: positions repeat, keep every occurrence in array order
        std::sort(f.arrayPos.begin(), f.arrayPos.end());
        for (const auto& ap : f.arrayPos) out.bruteOrder.push_back(ap.second);
    }
    if (verbose) {
        const auto tEnd = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "flatten: walk %.2f ms, triangle records %.2f ms, array order %.2f ms\n", ms(tWalk, tRecords), ms(tRecords, tOrder), ms(tOrder, tEnd));
    }
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// Query tree: binned surface-area heuristic over the compact triangles, leaves of at most 4.
// ---------------------------------------------------------------------------------------------
namespace {
struct QueryBuilder {
    static constexpr int kBins = 16;
    uint32_t kLeaf = 4;
    struct BNode { // binary tree of the build; collapsed into Node4 afterwards
        vec3 lo, hi;
        uint32_t child[2]; // index into `tree`, or a leaf reference (kLeafBit set)
    };
    std::vector<vec3> lo, hi, c;
    std::vector<uint32_t> idx;
    std::vector<BNode> tree;
    bool sah = true;

    static float area(const vec3 a, const vec3 b)
    {
        const vec3 e = b - a;
        return e.x * e.y + e.y * e.z + e.z * e.x;
    }
    static float comp(const vec3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

    // builds [b, e) of idx; returns a reference (tree index or leaf) and the subtree's box
    uint32_t build(uint32_t b, uint32_t e, vec3& boxLo, vec3& boxHi)
    {
        boxLo = splat(FLT_MAX), boxHi = splat(-FLT_MAX);
        vec3 cLo = splat(FLT_MAX), cHi = splat(-FLT_MAX);
        for (uint32_t i = b; i < e; i++) {
            boxLo = gmin(boxLo, lo[idx[i]]), boxHi = gmax(boxHi, hi[idx[i]]);
            cLo = gmin(cLo, c[idx[i]]), cHi = gmax(cHi, c[idx[i]]);
        }
        const uint32_t n = e - b;
        if (n <= kLeaf) return kLeafBit | (n << kRefCountShift) | b;
        uint32_t mid = b;
        if (sah) {
            float bestCost = FLT_MAX;
            int bestAxis = -1, bestBin = 0;
            for (int a = 0; a < 3; a++) {
                const float c0 = comp(cLo, a), c1 = comp(cHi, a);
                if (!(c1 > c0)) continue;
                const float scale = float(kBins) / (c1 - c0);
                vec3 bl[kBins], bh[kBins];
                uint32_t bn[kBins];
                for (int k = 0; k < kBins; k++) bl[k] = splat(FLT_MAX), bh[k] = splat(-FLT_MAX), bn[k] = 0;
                for (uint32_t i = b; i < e; i++) {
                    int k = int((comp(c[idx[i]], a) - c0) * scale);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    bl[k] = gmin(bl[k], lo[idx[i]]), bh[k] = gmax(bh[k], hi[idx[i]]), bn[k]++;
                }
                float rightArea[kBins];
                uint32_t rightN[kBins];
                vec3 rl = splat(FLT_MAX), rh = splat(-FLT_MAX);
                uint32_t rn = 0;
                for (int k = kBins - 1; k > 0; k--) {
                    rl = gmin(rl, bl[k]), rh = gmax(rh, bh[k]), rn += bn[k];
                    rightArea[k] = rn ? area(rl, rh) : 0.0f, rightN[k] = rn;
                }
                vec3 ll = splat(FLT_MAX), lh = splat(-FLT_MAX);
                uint32_t ln = 0;
                for (int k = 1; k < kBins; k++) { // split between bin k-1 and bin k
                    ll = gmin(ll, bl[k - 1]), lh = gmax(lh, bh[k - 1]), ln += bn[k - 1];
                    if (ln == 0 || rightN[k] == 0) continue;
                    const float cost = area(ll, lh) * float(ln) + rightArea[k] * float(rightN[k]);
                    if (cost < bestCost) bestCost = cost, bestAxis = a, bestBin = k;
                }
            }
            if (bestAxis >= 0) {
                const float c0 = comp(cLo, bestAxis), c1 = comp(cHi, bestAxis);
                const float scale = float(kBins) / (c1 - c0);
                const int axis = bestAxis, bin = bestBin;
                mid = uint32_t(std::partition(idx.begin() + b, idx.begin() + e,
                                              [&](uint32_t t) {
                                                  int k = int((comp(c[t], axis) - c0) * scale);
                                                  k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                                                  return k < bin;
                                              })
                               - idx.begin());
            }
        }
        if (mid == b || mid == e) { // no useful plane: halve along the widest centroid axis
            const vec3 ext = cHi - cLo;
            const int a = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
            mid = b + n / 2;
            std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e,
                             [&](uint32_t x, uint32_t y) { return comp(c[x], a) < comp(c[y], a); });
        }
        const uint32_t me = uint32_t(tree.size());
        tree.emplace_back();
        BNode nd;
        nd.lo = boxLo, nd.hi = boxHi;
        vec3 l, h;
        nd.child[0] = build(b, mid, l, h);
        nd.child[1] = build(mid, e, l, h);
        tree[me] = nd;
        return me;
    }

    // box of a reference of the binary tree
    void boxOf(uint32_t ref, vec3& l, vec3& h) const
    {
        if (ref & kLeafBit) {
            const uint32_t first = ref & kRefOffsetMask, cnt = (ref >> kRefCountShift) & kMaxLeafCount;
            l = splat(FLT_MAX), h = splat(-FLT_MAX);
            for (uint32_t i = first; i < first + cnt; i++) l = gmin(l, lo[idx[i]]), h = gmax(h, hi[idx[i]]);
        } else {
            l = tree[ref].lo, h = tree[ref].hi;
        }
    }

    // NodePair records in pre-order for binary node `b` (as the reference's tree is laid out); returns the pair index
    uint32_t emitPairs(uint32_t b, std::vector<NodePair>& out, uint32_t depth, uint32_t& maxDepth) const
    {
        if (depth > maxDepth) maxDepth = depth;
        const uint32_t me = uint32_t(out.size());
        out.emplace_back();
        NodePair np;
        vec3 l, h;
        boxOf(tree[b].child[0], l, h);
        np.lo0[0] = l.x, np.lo0[1] = l.y, np.lo0[2] = l.z, np.hi0[0] = h.x, np.hi0[1] = h.y, np.hi0[2] = h.z;
        boxOf(tree[b].child[1], l, h);
        np.lo1[0] = l.x, np.lo1[1] = l.y, np.lo1[2] = l.z, np.hi1[0] = h.x, np.hi1[1] = h.y, np.hi1[2] = h.z;
        np.pad0 = np.pad1 = 0;
        np.ref0 = (tree[b].child[0] & kLeafBit) ? tree[b].child[0] : emitPairs(tree[b].child[0], out, depth + 1, maxDepth);
        np.ref1 = (tree[b].child[1] & kLeafBit) ? tree[b].child[1] : emitPairs(tree[b].child[1], out, depth + 1, maxDepth);
        out[me] = np;
        return me;
    }

    // Node4 for binary node `b`: its two children, then the inner child with the largest box opened until four
    uint32_t collapse(uint32_t b, std::vector<Node4>& out, uint32_t depth, uint32_t& maxDepth)
    {
        if (depth > maxDepth) maxDepth = depth;
        uint32_t kids[4] = { tree[b].child[0], tree[b].child[1], kEmptyRef, kEmptyRef };
        int n = 2;
        while (n < 4) {
            int best = -1;
            float bestArea = -1.0f;
            for (int k = 0; k < n; k++) {
                if (kids[k] & kLeafBit) continue;
                const float a = area(tree[kids[k]].lo, tree[kids[k]].hi);
                if (a > bestArea) bestArea = a, best = k;
            }
            if (best < 0) break;
            const uint32_t open = kids[best];
            kids[best] = tree[open].child[0];
            kids[n++] = tree[open].child[1];
        }
        const uint32_t me = uint32_t(out.size());
        out.emplace_back();
        Node4 nd;
        for (int k = 0; k < 4; k++) {
            vec3 l = splat(0.0f), h = splat(0.0f);
            if (k < n) boxOf(kids[k], l, h);
            nd.lox[k] = l.x, nd.loy[k] = l.y, nd.loz[k] = l.z, nd.hix[k] = h.x, nd.hiy[k] = h.y, nd.hiz[k] = h.z;
            nd.ref[k] = kEmptyRef, nd.pad[k] = 0;
        }
        for (int k = 0; k < n; k++) nd.ref[k] = (kids[k] & kLeafBit) ? kids[k] : collapse(kids[k], out, depth + 1, maxDepth);
        out[me] = nd;
        return me;
    }
};
} // namespace

void build_query_tree(const FlatScene& flat, QueryTree& out, bool onDevice, bool hostLayouts, const RefPrimitive* prims, const float* dLo4,
                      const float* dHi4)
{
    out = QueryTree {};
    const uint32_t n = uint32_t(flat.srcPos.size());
    // the proof a query needs rests on nested reference boxes and on slab distances that are exact with reciprocals
    if (n <= 8 || !flat.nested || !flat.fastBounds || (flat.rootRef & kLeafBit)) return;
    QueryBuilder qb;
    if (const char* e = getenv("RT_AMD_QLEAF")) qb.kLeaf = uint32_t(atoi(e));
    if (qb.kLeaf < 1 || qb.kLeaf > 8) qb.kLeaf = 4;
    qb.idx.resize(n);
    // the triangles' boxes on the host: for the host builder, and for a device build whose caller has not put them on the device
    auto host_boxes = [&]() {
        if (!qb.lo.empty()) return true;
        if (flat.triEdges.size() != n && !prims) return false;
        qb.lo.resize(n), qb.hi.resize(n), qb.c.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            vec3 v0, v1, v2;
            if (flat.triEdges.size() == n) {
                const TriEdges& t = flat.triEdges[i];
                v0 = make3(t.v0[0], t.v0[1], t.v0[2]), v1 = make3(t.v1x, t.v1y, t.v1z), v2 = make3(t.v2[0], t.v2[1], t.v2[2]);
            } else {
                const RefPrimitive& p = prims[flat.srcPos[i]];
                v0 = load3(p.v0.position), v1 = load3(p.v1.position), v2 = load3(p.v2.position);
            }
            qb.lo[i] = gmin(gmin(v0, v1), v2), qb.hi[i] = gmax(gmax(v0, v1), v2);
            qb.c[i] = (qb.lo[i] + qb.hi[i]) * 0.5f;
        }
        return true;
    };
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t0) { return std::chrono::duration<float, std::milli>(clock::now() - t0).count(); };
    auto emit = [&](uint32_t root) {
        out.nodes.clear(), out.pairs.clear();
        out.nodes.reserve(qb.tree.size()), out.pairs.reserve(qb.tree.size()); // (one pair per binary node, at most as many 4-wide nodes)
        out.maxDepth = 0, out.pairDepth = 1; // pairDepth counts levels of nodes as the flattener does: the root is level 1
        out.rootRef = qb.collapse(root, out.nodes, 1, out.maxDepth);
        out.pairRootRef = qb.emitPairs(root, out.pairs, 2, out.pairDepth);
    };
    bool built = false;
    if (onDevice) { // the same split rule, one level per launch (query_build.hip)
        const auto t0 = clock::now();
        std::vector<float> lo4, hi4;
        const bool resident = dLo4 && dHi4; // the caller computed the boxes on the device
        if (!resident && host_boxes()) {
            lo4.resize(size_t(n) * 4), hi4.resize(size_t(n) * 4);
            for (uint32_t i = 0; i < n; i++) {
                lo4[4 * size_t(i)] = qb.lo[i].x, lo4[4 * size_t(i) + 1] = qb.lo[i].y, lo4[4 * size_t(i) + 2] = qb.lo[i].z, lo4[4 * size_t(i) + 3] = 0.0f;
                hi4[4 * size_t(i)] = qb.hi[i].x, hi4[4 * size_t(i) + 1] = qb.hi[i].y, hi4[4 * size_t(i) + 2] = qb.hi[i].z, hi4[4 * size_t(i) + 3] = 0.0f;
            }
        }
        std::vector<DeviceBuildNode> nodes;
        uint32_t levels = 0;
        if ((resident || !lo4.empty())
            && device_build_query_hierarchy(resident ? dLo4 : lo4.data(), resident ? dHi4 : hi4.data(), resident, n, qb.kLeaf, nodes, qb.idx, levels, &out.deviceKernelMs)
            && !nodes.empty()) {
            qb.tree.resize(nodes.size());
            for (size_t i = 0; i < nodes.size(); i++) {
                qb.tree[i].lo = make3(nodes[i].lo[0], nodes[i].lo[1], nodes[i].lo[2]);
                qb.tree[i].hi = make3(nodes[i].hi[0], nodes[i].hi[1], nodes[i].hi[2]);
                qb.tree[i].child[0] = nodes[i].child[0], qb.tree[i].child[1] = nodes[i].child[1];
            }
            out.hierarchyMs = ms_since(t0);
            const auto t1 = clock::now();
            if (!host_boxes()) return; // (the layouts hold the boxes of the leaves, which the hierarchy does not)
            emit(0);
            out.layoutMs = ms_since(t1);
            built = out.pairDepth <= 48;
            out.builtOnDevice = built;
        }
    }
    if (!built && !host_boxes()) return; // (no triangle data on the host: the scene does without a query tree)
    for (int attempt = 0; attempt < 2 && !built; attempt++) {
        const auto t0 = clock::now();
        qb.tree.clear();
        qb.sah = (attempt == 0);
        for (uint32_t i = 0; i < n; i++) qb.idx[i] = i;
        vec3 l, h;
        const uint32_t root = qb.build(0, n, l, h); // an inner node: n exceeds the largest leaf
        out.hierarchyMs = ms_since(t0);
        const auto t1 = clock::now();
        emit(root);
        out.layoutMs = ms_since(t1);
        if (out.pairDepth <= 48) break; // a degenerate heuristic tree: fall back to halving
    }
    const auto tLayout = clock::now();
    out.order = qb.idx;
    if (!hostLayouts) { // the caller gathers the triangle records by `order` itself (on the device)
        out.layoutMs += ms_since(tLayout);
        out.usable = true;
        return;
    }
    out.triPlane.resize(n), out.triEdges.resize(n), out.cert.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t t = qb.idx[i];
        out.triPlane[i] = flat.triPlane[t];
        out.triEdges[i] = flat.triEdges[t];
        TriCert& c = out.cert[i];
        memcpy(c.leafLo, &flat.leafBox[6 * size_t(t)], 12);
        memcpy(c.leafHi, &flat.leafBox[6 * size_t(t) + 3], 12);
        c.compact = t;
        c.route = flat.leafRoute[t];
    }
    // the 4-wide form once more, for four lanes per ray
    out.quadNodes.clear(), out.quadLeaves.clear();
    // (only the schedules that run four lanes per ray read it - RT_AMD_QUAD_TRACE, RT_AMD_PATH_QUADS - and it is as large as
    // everything else of the query tree together)
    if (qb.kLeaf <= 4 && (getenv("RT_AMD_QUAD_TRACE") || getenv("RT_AMD_PATH_QUADS"))) {
        out.quadNodes.resize(out.nodes.size());
        for (size_t i = 0; i < out.nodes.size(); i++) {
            const Node4& nd = out.nodes[i];
            QuadNode& qn = out.quadNodes[i];
            for (int k = 0; k < 4; k++) {
                QuadChild& c = qn.c[k];
                c.lo[0] = nd.lox[k], c.lo[1] = nd.loy[k], c.lo[2] = nd.loz[k];
                c.hi[0] = nd.hix[k], c.hi[1] = nd.hiy[k], c.hi[2] = nd.hiz[k];
                c.ref = nd.ref[k], c.pad = 0;
                if (nd.ref[k] == kEmptyRef || !(nd.ref[k] & kLeafBit)) continue;
                const uint32_t first = nd.ref[k] & kRefOffsetMask, cnt = (nd.ref[k] >> kRefCountShift) & kMaxLeafCount;
                QuadLeaf lf;
                memset(&lf, 0, sizeof(lf));
                for (uint32_t t = 0; t < cnt && t < 4; t++) {
                    memcpy(lf.plane[t], &out.triPlane[first + t], 16);
                    const float* e = reinterpret_cast<const float*>(&out.triEdges[first + t]);
                    for (int j = 0; j < 5; j++) memcpy(lf.e[j][t], e + 4 * j, 16);
                    const uint32_t id = first + t;
                    memcpy(&lf.e[4][t][2], &id, 4);
                }
                c.ref = kLeafBit | (cnt << kRefCountShift) | uint32_t(out.quadLeaves.size());
                out.quadLeaves.push_back(lf);
            }
        }
    }
    out.layoutMs += ms_since(tLayout);
    out.usable = true;
}

} // namespace rt

// =============================================================================================
// C ABI, host side
// =============================================================================================
using namespace rt;

extern "C" {

const char* rt_last_error(void) { return g_error; }

// Per-primitive boxes and centroids, the root and the reference's dummy second node: what Builder::build starts from.
static void builder_begin(Builder& bl, rt_bvh* b, const void* prims, uint32_t nPrims, bool sahBinning, bool perPrimitive)
{
    bl.in = static_cast<const RefPrimitive*>(prims);
    bl.out = b;
    bl.sahBinning = sahBinning;
    if (perPrimitive) {
        bl.lo.resize(nPrims);
        bl.hi.resize(nPrims);
        bl.centroid.resize(nPrims);
        for (uint32_t i = 0; i < nPrims; i++) {
            const RefPrimitive& p = bl.in[i];
            const vec3 a = load3(p.v0.position), c = load3(p.v1.position), d = load3(p.v2.position);
            bl.lo[i] = gmin(gmin(a, c), d); // computePrimitiveAABB operand order (src/bvh.cpp:111-118)
            bl.hi[i] = gmax(gmax(a, c), d);
            bl.centroid[i] = (a + c + d) / 3.0f; // computePrimitiveCentroid (src/bvh.cpp:144-152)
        }
    }
    b->prims.reserve(size_t(nPrims) * 2);
    b->primIds.reserve(size_t(nPrims) * 2);
    b->nodes.reserve(size_t(nPrims) + 2);
    b->nodes.emplace_back(); // root
    b->nodes.emplace_back(); // the reference's dummy second node (src/bvh.cpp:82): default box, zero data
    RefNode& dummy = b->nodes[1];
    dummy.lower[0] = dummy.lower[1] = dummy.lower[2] = 0.0f;
    dummy.upper[0] = dummy.upper[1] = dummy.upper[2] = 1.0f;
    dummy.data[0] = dummy.data[1] = 0;
}
static int builder_end(Builder& bl, rt_bvh* b, rt_bvh** out)
{
    if (bl.undefined) {
        delete b;
        set_error("rt_bvh_build_sah_binning: more than four triangles share their centroid coordinate on the split axis; the "
                  "reference's bin index (src/extra.cpp:590-598) is 0/0 converted to size_t for them, which is undefined");
        return RT_ERR_INVALID;
    }
    countLevelsAndLeaves(*b);
    *out = b;
    return RT_OK;
}

static int bvh_build(const void* prims, uint32_t nPrims, bool sahBinning, rt_bvh** out);
int rt_bvh_build(const void* prims, uint32_t nPrims, rt_bvh** out) { return bvh_build(prims, nPrims, false, out); }
int rt_bvh_build_sah_binning(const void* prims, uint32_t nPrims, rt_bvh** out) { return bvh_build(prims, nPrims, true, out); }
static int bvh_build(const void* prims, uint32_t nPrims, bool sahBinning, rt_bvh** out)
{
    if (!out || (nPrims && !prims)) {
        set_error("rt_bvh_build: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    rt_bvh* b = new (std::nothrow) rt_bvh();
    if (!b) {
        set_error("rt_bvh_build: out of memory");
        return RT_ERR_NOMEM;
    }
    Builder bl;
    builder_begin(bl, b, prims, nPrims, sahBinning, true);
    bl.order.resize(nPrims);
    for (uint32_t i = 0; i < nPrims; i++) bl.order[i] = i;
    bl.build(bl.order.data(), nPrims, 0);
    return builder_end(bl, b, out);
}

namespace {
// BVH::buildRecursive's numbering (src/bvh.cpp:429-512) over spans that are already settled: a node's two children get the
// next two node indices when the node is reached, then the left subtree is numbered, then the right; a leaf's primitives
// are appended twice and the leaf points at the second copy (see Builder::build, which this hands over to for a span the
// device left to the host).
struct SpanAssembler {
    Builder& bl;
    const std::vector<RefBuildNode>& tree;
    uint32_t* order;
    uint32_t nPrims;
    uint32_t hostPrims = 0;
    bool prepared = false;

    void place(uint32_t t, uint32_t nodeIndex)
    {
        const RefBuildNode& T = tree[t];
        rt_bvh* const out = bl.out;
        if (T.split == kRefBuildHost) {
            if (!prepared) { // the host builder's per-primitive tables, for the primitives of the spans it gets
                prepared = true;
                bl.lo.resize(nPrims), bl.hi.resize(nPrims), bl.centroid.resize(nPrims);
                bl.filled.assign(nPrims, 0);
            }
            for (uint32_t k = 0; k < T.count; k++) {
                const uint32_t i = order[T.begin + k];
                if (bl.filled[i]) continue;
                bl.filled[i] = 1;
                const RefPrimitive& p = bl.in[i];
                const vec3 a = load3(p.v0.position), c = load3(p.v1.position), d = load3(p.v2.position);
                bl.lo[i] = gmin(gmin(a, c), d);
                bl.hi[i] = gmax(gmax(a, c), d);
                bl.centroid[i] = (a + c + d) / 3.0f;
            }
            hostPrims += T.count;
            bl.build(order + T.begin, T.count, nodeIndex);
            return;
        }
        RefNode node;
        for (int k = 0; k < 3; k++) node.lower[k] = T.lo[k], node.upper[k] = T.hi[k];
        if (T.split == 0) {
            bl.emitLeafCopy(order + T.begin, T.count);
            node.data[0] = uint32_t(out->prims.size()) | kLeafBit;
            node.data[1] = T.count;
            out->nodes[nodeIndex] = node;
            bl.emitLeafCopy(order + T.begin, T.count);
            return;
        }
        const uint32_t left = uint32_t(out->nodes.size());
        out->nodes.emplace_back();
        out->nodes.emplace_back();
        node.data[0] = left;
        node.data[1] = left + 1;
        out->nodes[nodeIndex] = node;
        place(T.left, left);
        place(T.left + 1, left + 1);
    }
};
} // namespace

int rt_bvh_build_device(const void* prims, uint32_t nPrims, int sahBinning, rt_bvh** out, uint32_t info[4])
{
    if (info) info[0] = info[1] = info[2] = info[3] = 0u;
    if (nPrims <= 4u) return bvh_build(prims, nPrims, sahBinning != 0, out); // the root is a leaf: nothing to sort
    if (!out || !prims) {
        set_error("rt_bvh_build_device: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    std::vector<RefBuildNode> tree;
    std::vector<uint32_t> order;
    uint32_t stats[4] = { 0, 0, 0, 0 };
    const bool verbose = getenv("RT_AMD_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    float kernelMs = 0.0f;
    if (!device_build_reference_spans(static_cast<const RefPrimitive*>(prims), nPrims, sahBinning != 0, tree, order, stats, &kernelMs)) {
        set_error("rt_bvh_build_device: the device build did not complete (no usable GPU, out of device memory, or a tree deeper than "
                  "96 levels); rt_bvh_build builds on the host");
        return RT_ERR_HIP;
    }
    const auto t1 = std::chrono::steady_clock::now();
    rt_bvh* b = new (std::nothrow) rt_bvh();
    if (!b) {
        set_error("rt_bvh_build_device: out of memory");
        return RT_ERR_NOMEM;
    }
    Builder bl;
    builder_begin(bl, b, prims, nPrims, sahBinning != 0, false);
    SpanAssembler as { bl, tree, order.data(), nPrims };
    as.place(0u, 0u);
    if (info) info[0] = stats[0], info[1] = stats[1], info[2] = stats[2], info[3] = as.hostPrims;
    const int rc = builder_end(bl, b, out);
    if (verbose) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "rt_bvh_build_device: %u primitives, %u spans in %u levels on the device (%.2f ms, launches %.2f), %u spans with %u primitives "
                        "left to the host; numbering and copying %.2f ms\n",
                nPrims, stats[0], stats[2], std::chrono::duration<double, std::milli>(t1 - t0).count(), kernelMs, stats[1], as.hostPrims,
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    return rc;
}
void rt_bvh_destroy(rt_bvh* bvh) { delete bvh; }
uint32_t rt_bvh_num_nodes(const rt_bvh* b) { return b ? uint32_t(b->nodes.size()) : 0; }
uint32_t rt_bvh_num_primitives(const rt_bvh* b) { return b ? uint32_t(b->prims.size()) : 0; }
uint32_t rt_bvh_num_levels(const rt_bvh* b) { return b ? b->numLevels : 0; }
uint32_t rt_bvh_num_leaves(const rt_bvh* b) { return b ? b->numLeaves : 0; }
const void* rt_bvh_nodes(const rt_bvh* b) { return b ? b->nodes.data() : nullptr; }
const void* rt_bvh_primitives(const rt_bvh* b) { return b ? b->prims.data() : nullptr; }
const uint32_t* rt_bvh_primitive_ids(const rt_bvh* b) { return b ? b->primIds.data() : nullptr; }

int rt_selftest_query_tree(const void* nodes, uint32_t nNodes, const void* prims, uint32_t nPrims, uint32_t nMaterials, uint32_t info[8])
{
    return rt_selftest_query_tree_built(nodes, nNodes, prims, nPrims, nMaterials, 0, info, nullptr);
}

int rt_selftest_query_tree_built(const void* nodes, uint32_t nNodes, const void* prims, uint32_t nPrims, uint32_t nMaterials, int onDevice,
                                 uint32_t info[8], float times[4])
{
    if (!nodes || !info) {
        set_error("rt_selftest_query_tree: null argument");
        return RT_ERR_INVALID;
    }
    FlatScene flat;
    const char* err = "";
    const int rc = flatten(static_cast<const RefNode*>(nodes), nNodes, static_cast<const RefPrimitive*>(prims), nPrims, nullptr, nMaterials,
                           flat, &err);
    if (rc != RT_OK) {
        set_error("%s", err);
        return rc;
    }
    QueryTree qt;
    build_query_tree(flat, qt, onDevice != 0);
    if (times) times[0] = qt.hierarchyMs, times[1] = qt.deviceKernelMs, times[2] = qt.layoutMs, times[3] = qt.builtOnDevice ? 1.0f : 0.0f;
    const uint32_t n = uint32_t(flat.triPlane.size());
    info[0] = qt.usable ? 1u : 0u, info[1] = n, info[2] = uint32_t(qt.pairs.size()), info[3] = qt.pairDepth;
    info[4] = uint32_t(qt.nodes.size()), info[5] = qt.maxDepth, info[6] = uint32_t(flat.pairs.size()), info[7] = flat.maxDepth;
    if (!qt.usable) return RT_OK;
    auto fail = [](const char* what) {
        set_error("rt_selftest_query_tree: %s", what);
        return RT_ERR_INVALID;
    };
    // the triangle records: a permutation of the reference's, each with its reference leaf's box and route
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        const TriCert& c = qt.cert[i];
        if (c.compact >= n || seen[c.compact]) return fail("triangle records are not a permutation");
        seen[c.compact] = 1;
        if (memcmp(&qt.triPlane[i], &flat.triPlane[c.compact], sizeof(TriPlane)) || memcmp(&qt.triEdges[i], &flat.triEdges[c.compact], sizeof(TriEdges)))
            return fail("triangle data differs from the reference's");
        // walk the route in the reference's tree: it must end in the leaf that holds the triangle, with this box
        uint32_t ref = flat.rootRef;
        const float *lo = flat.rootLo, *hi = flat.rootHi;
        int top = 31;
        while (top > 0 && !((c.route >> top) & 1u)) top--;
        for (int k = top - 1; k >= 0; k--) {
            if (ref & kLeafBit) return fail("route runs past a leaf");
            const NodePair& p = flat.pairs[ref];
            const bool second = (c.route >> k) & 1u;
            lo = second ? p.lo1 : p.lo0, hi = second ? p.hi1 : p.hi0, ref = second ? p.ref1 : p.ref0;
        }
        if (!(ref & kLeafBit)) return fail("route does not end in a leaf");
        const uint32_t first = ref & kRefOffsetMask;
        uint32_t cnt = (ref >> kRefCountShift) & kMaxLeafCount;
        if (cnt == kMaxLeafCount && !flat.leafCounts.empty()) cnt = flat.leafCounts[first];
        if (c.compact < first || c.compact >= first + cnt) return fail("route ends in another leaf");
        if (memcmp(lo, c.leafLo, 12) || memcmp(hi, c.leafHi, 12)) return fail("leaf box differs");
    }
    auto triInside = [&](uint32_t t, const float* lo, const float* hi) {
        const TriEdges& e = qt.triEdges[t];
        const float v[3][3] = { { e.v0[0], e.v0[1], e.v0[2] }, { e.v1x, e.v1y, e.v1z }, { e.v2[0], e.v2[1], e.v2[2] } };
        for (int a = 0; a < 3; a++)
            for (int k = 0; k < 3; k++)
                if (!(v[a][k] >= lo[k] && v[a][k] <= hi[k])) return false;
        return true;
    };
    // binary form: leaves cover [0, n) once, child boxes nested and around their triangles
    std::vector<uint8_t> covered(n, 0);
    std::vector<uint64_t> leavesA, leavesB;
    struct Item { uint32_t ref; float lo[3], hi[3]; };
    std::vector<Item> stack;
    auto pushChild = [&](uint32_t ref, const float* lo, const float* hi, const float* plo, const float* phi, bool checkParent) {
        if (checkParent)
            for (int k = 0; k < 3; k++)
                if (!(lo[k] >= plo[k] && hi[k] <= phi[k])) return false;
        Item it;
        it.ref = ref;
        memcpy(it.lo, lo, 12), memcpy(it.hi, hi, 12);
        stack.push_back(it);
        return true;
    };
    for (int form = 0; form < 2; form++) {
        std::vector<uint64_t>& leaves = form == 0 ? leavesA : leavesB;
        stack.clear();
        Item root;
        root.ref = form == 0 ? qt.pairRootRef : qt.rootRef;
        memcpy(root.lo, flat.rootLo, 12), memcpy(root.hi, flat.rootHi, 12);
        stack.push_back(root);
        bool isRoot = true;
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.ref & kLeafBit) {
                const uint32_t first = it.ref & kRefOffsetMask, cnt = (it.ref >> kRefCountShift) & kMaxLeafCount;
                if (first + cnt > n || cnt == 0) return fail("leaf range out of bounds");
                for (uint32_t t = first; t < first + cnt; t++) {
                    if (!triInside(t, it.lo, it.hi)) return fail("triangle outside its leaf box");
                    if (form == 0) {
                        if (covered[t]) return fail("triangle in two leaves");
                        covered[t] = 1;
                    }
                }
                leaves.push_back((uint64_t(first) << 32) | cnt);
                continue;
            }
            if (form == 0) {
                if (it.ref >= qt.pairs.size()) return fail("pair index out of range");
                const NodePair& p = qt.pairs[it.ref];
                if (!pushChild(p.ref0, p.lo0, p.hi0, it.lo, it.hi, !isRoot) || !pushChild(p.ref1, p.lo1, p.hi1, it.lo, it.hi, !isRoot))
                    return fail("child box outside its parent's");
            } else {
                if (it.ref >= qt.nodes.size()) return fail("node index out of range");
                const Node4& nd = qt.nodes[it.ref];
                for (int k = 0; k < 4; k++) {
                    if (nd.ref[k] == kEmptyRef) continue;
                    const float lo[3] = { nd.lox[k], nd.loy[k], nd.loz[k] }, hi[3] = { nd.hix[k], nd.hiy[k], nd.hiz[k] };
                    if (!pushChild(nd.ref[k], lo, hi, it.lo, it.hi, !isRoot)) return fail("child box outside its parent's");
                }
            }
            isRoot = false;
        }
    }
    for (uint32_t t = 0; t < n; t++)
        if (!covered[t]) return fail("triangle in no leaf");
    std::sort(leavesA.begin(), leavesA.end());
    std::sort(leavesB.begin(), leavesB.end());
    if (leavesA != leavesB) return fail("the two forms of the tree hold different leaves");
    return RT_OK;
}

int rt_camera_from_trackball(float fovDegrees, float distance, const float lookAt[3], const float rotationDegrees[3],
                             float aspect, rt_camera* out)
{
    if (!lookAt || !rotationDegrees || !out) {
        set_error("rt_camera_from_trackball: null argument");
        return RT_ERR_INVALID;
    }
    const float kRad = static_cast<float>(0.01745329251994329576923690768489); // glm::radians
    const float ex = rotationDegrees[0] * kRad, ey = rotationDegrees[1] * kRad, ez = rotationDegrees[2] * kRad;
    const float fovy = fovDegrees * kRad;
    // glm::quat(eulerAngles), type_quat.inl:208-217
    const float cx = cosf(ex * 0.5f), cy = cosf(ey * 0.5f), cz = cosf(ez * 0.5f);
    const float sx = sinf(ex * 0.5f), sy = sinf(ey * 0.5f), sz = sinf(ez * 0.5f);
    out->quat[0] = cx * cy * cz + sx * sy * sz;
    out->quat[1] = sx * cy * cz - cx * sy * sz;
    out->quat[2] = cx * sy * cz + sx * cy * sz;
    out->quat[3] = cx * cy * sz - sx * sy * cz;
    const vec3 pos = load3(lookAt) + quat_rotate(out->quat, make3(0.0f, 0.0f, -distance));
    out->position[0] = pos.x, out->position[1] = pos.y, out->position[2] = pos.z;
    out->halfHeight = tanf(fovy / 2.0f);
    out->halfWidth = aspect * out->halfHeight;
    return RT_OK;
}

} // extern "C"
