// Device build of the reference's own tree: the spans, boxes and primitive order of BVH::buildRecursive
// (src/bvh.cpp:429-512; median split src/bvh.cpp:181-214; SAH-binned split src/extra.cpp:582-699), level by level.
//
// What the reference computes per span is a function of the span's SET of primitives except in two places:
//   * std::sort by the centroid coordinate on the span's longest axis.  Distinct keys have one sorted order; equal keys
//     (float coincidences are the rule among tens of thousands of centroids) are left in whatever order the host's introsort
//     produces from the order it was given.  That order only shows in the result when the equal keys reach a leaf side by
//     side, or sit either side of the median; otherwise the next sort on a span that holds them, on an axis where they
//     differ, puts them in their one order again, and a split that parts them ends the matter.
//   * the span's box is folded with glm::min / glm::max in span order, which keeps the FIRST of +0 and -0 it meets.
// So every span carries one bit - "the order I was given is the reference's" - and the record of the nearest span above it
// (or itself) for which that holds.  A span sorted with equal keys passes the bit on to a child only if no two equal keys
// lie side by side inside the child.  When the unknown order would show - a leaf of at most four primitives given in unknown
// order, a span the heuristic does not split (a leaf in sorted order) with equal keys, equal keys either side of the median, a
// zero box coordinate with both signs present, a key that is NaN, a bin index the reference's conversion leaves undefined -
// that nearest span is marked, and the host builder rebuilds it and everything below it from the order the span was given,
// which the device keeps for every level.
// Everything else - the box (min / max are exact), the longest axis, the keys, the bin of a key, the areas and the costs of
// the 49 candidate splits - is evaluated here with the host builder's operations in the host builder's order (this file is
// compiled without FMA contraction, with correctly rounded division).
//
// One level = one pass over all n positions of the primitive order:
//   bounds   every open span's box, by atomics on order-preserving integers (a wave that lies inside one span reduces first)
//   keys     64-bit sort key per position: span begin << 32 | centroid coordinate; positions outside open spans: position << 32
//   sort     one radix sort of the whole array into the next level's copy: every span is sorted in place, the rest stays put
//   ties     adjacent equal keys (and NaN keys) are noted per span
//   bins     (SAH) per span: count and box of each of up to 50 bins of equal width over the span's key interval
//   decide   one thread per span: writes the span's record (box, begin, count, split, children), opens the children or marks
//   apply    every position learns the span it now belongs to; equal neighbours inside one child clear the child's bit
// The node numbering and the primitive array of BVHInterface::nodes()/primitives() follow from the spans by a depth-first
// walk on the host (rt_bvh_build_device).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstdint>
#include <vector>

#include "rt_types.h"

namespace rt {
namespace {

constexpr int kBlockR = 256;
constexpr uint32_t kClosed = 0xffffffffu;
constexpr uint32_t kDecideLeaf = 0xfffffffeu;
constexpr int kSahBins = 50;
constexpr int kBinWords = 9;     // count, lower[3], upper[3]; then, written by the span's deciding thread: primitives and box area left of the bin
constexpr int kLevelCounters = 8; // counters[kLevelCounters + level] = spans opened for that level
constexpr int kMaxRefLevels = 96;

struct RefBuildParams {
    const uint8_t* prims; // RefPrimitive records as handed in
    float4 *lo, *hi, *cent; // per primitive
    uint32_t *idxIn, *idxOut;
    unsigned long long *keyIn, *keyOut;
    uint32_t* segBegin;                                // per position: begin of its open span, or kClosed
    uint32_t *segCount, *segNode, *segFlag, *zeroMask; // per span begin (segFlag: 1 = equal keys, 2 = a NaN key)
    uint32_t *segAmbig, *segAnc;                       // per span begin: the span's order is not known to be the reference's; the
                                                       // record to rebuild from if that shows
    uint32_t* decision;                                // per span begin: left size, kDecideLeaf or kRefBuildHost
    uint32_t* box;                                     // per span begin: lower[3], upper[3] as ordered integers
    uint32_t* bins;                                    // per span begin / 5: kSahBins x kBinWords
    RefBuildNode* tree;
    uint32_t* counters; // [0] records, [kLevelCounters + level]
    uint32_t* hostMark; // per record: the host rebuilds this span
    uint32_t n, maxNodes, sah, level;
};

__device__ __forceinline__ uint32_t ordered(const float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(const uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
// calculateAABBSurfaceArea (src/extra.cpp:609-613)
__device__ __forceinline__ float surface_area(const float* lower, const float* upper)
{
    const float x = fabsf(upper[0] - lower[0]), y = fabsf(upper[1] - lower[1]), z = fabsf(upper[2] - lower[2]);
    return 2.0f * (x * y + x * z + y * z);
}

__global__ __launch_bounds__(kBlockR) void ref_prim_kernel(const RefBuildParams P)
{
    const uint32_t i = blockIdx.x * kBlockR + threadIdx.x;
    if (i >= P.n) return;
    const float* p = reinterpret_cast<const float*>(P.prims + size_t(i) * sizeof(RefPrimitive) + 4);
    const vec3 a = load3(p), c = load3(p + 8), d = load3(p + 16);
    const vec3 lo = gmin(gmin(a, c), d), hi = gmax(gmax(a, c), d); // computePrimitiveAABB (src/bvh.cpp:111-118)
    const vec3 ce = (a + c + d) / 3.0f;                            // computePrimitiveCentroid (src/bvh.cpp:144-152)
    P.lo[i] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    P.hi[i] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    P.cent[i] = make_float4(ce.x, ce.y, ce.z, 0.0f);
    P.idxIn[i] = i;
    P.segBegin[i] = 0u;
}

__device__ __forceinline__ void reset_span(const RefBuildParams& P, const uint32_t b, const uint32_t count, const uint32_t node, const uint32_t anc)
{
    P.segCount[b] = count, P.segNode[b] = node, P.segFlag[b] = 0u, P.zeroMask[b] = 0u;
    P.segAmbig[b] = 0u, P.segAnc[b] = anc;
    const uint32_t lo = ordered(FLT_MAX), hi = ordered(-FLT_MAX); // computeSpanAABB starts from these (src/bvh.cpp:125-137)
    for (int k = 0; k < 3; k++) P.box[size_t(b) * 6 + k] = lo, P.box[size_t(b) * 6 + 3 + k] = hi;
}

__global__ void ref_root_kernel(const RefBuildParams P)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) reset_span(P, 0u, P.n, 0u, 0u);
}

__global__ __launch_bounds__(kBlockR) void ref_bounds_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    const uint32_t b = pos < P.n ? P.segBegin[pos] : kClosed;
    const bool open = b != kClosed;
    uint32_t v[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u }; // neutral for min / max; a NaN coordinate stays neutral
    uint32_t zeros = 0u;
    if (open) {
        const uint32_t t = P.idxIn[pos];
        const float4 l = P.lo[t], h = P.hi[t];
        const float f[6] = { l.x, l.y, l.z, h.x, h.y, h.z };
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (f[k] == f[k]) v[k] = ordered(f[k]);
            if (f[k] == 0.0f) zeros |= 1u << (2 * k + (__float_as_uint(f[k]) >> 31));
        }
    }
    const uint32_t b0 = __shfl(b, 0);
    if (__ballot(b != b0) == 0ull) { // the whole wave lies in one span (or outside all)
        if (b0 == kClosed) return;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) v[k] = min(v[k], (uint32_t)__shfl_xor(v[k], off)), v[3 + k] = max(v[3 + k], (uint32_t)__shfl_xor(v[3 + k], off));
            zeros |= __shfl_xor(zeros, off);
        }
        if ((threadIdx.x & 63) != 0) return;
    } else if (!open) {
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (v[k] != 0xffffffffu) atomicMin(P.box + size_t(b) * 6 + k, v[k]);
        if (v[3 + k] != 0u) atomicMax(P.box + size_t(b) * 6 + 3 + k, v[3 + k]);
    }
    if (zeros) atomicOr(P.zeroMask + b, zeros);
}

__device__ __forceinline__ void load_box(const RefBuildParams& P, const uint32_t b, float* lower, float* upper)
{
    for (int k = 0; k < 3; k++) lower[k] = unordered(P.box[size_t(b) * 6 + k]), upper[k] = unordered(P.box[size_t(b) * 6 + 3 + k]);
}

__global__ __launch_bounds__(kBlockR) void ref_keys_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    if (pos >= P.n) return;
    const uint32_t b = P.segBegin[pos];
    uint2 key = make_uint2(0u, pos); // low word, high word
    if (b != kClosed && P.segCount[b] > 4u) {
        const uint32_t* const bx = P.box + size_t(b) * 6;
        // computeAABBLongestAxis (src/bvh.cpp:160-169): ties go to the lower axis index
        const float ax = fabsf(unordered(bx[3]) - unordered(bx[0])), ay = fabsf(unordered(bx[4]) - unordered(bx[1])),
                    az = fabsf(unordered(bx[5]) - unordered(bx[2]));
        const float4 c = P.cent[P.idxIn[pos]];
        float k = c.z;
        if (ay >= ax && ay >= az) k = c.y;
        if (ax >= ay && ax >= az) k = c.x;
        key = make_uint2(ordered(k), b);
    }
    reinterpret_cast<uint2*>(P.keyIn)[pos] = key;
}

__global__ __launch_bounds__(kBlockR) void ref_ties_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    if (pos >= P.n) return;
    const uint32_t b = P.segBegin[pos];
    if (b == kClosed) return;
    const uint32_t cnt = P.segCount[b];
    if (cnt <= 4u) return;
    const float k = unordered((uint32_t)P.keyOut[pos]);
    uint32_t flag = k == k ? 0u : 2u;
    if (pos + 1 < b + cnt && k == unordered((uint32_t)P.keyOut[pos + 1])) flag |= 1u;
    if (flag) atomicOr(P.segFlag + b, flag);
}

__global__ __launch_bounds__(kBlockR) void ref_bins_clear_kernel(const RefBuildParams P)
{
    const size_t i = size_t(blockIdx.x) * kBlockR + threadIdx.x;
    const size_t total = size_t(P.n / 5u + 1u) * kSahBins * kBinWords;
    if (i >= total) return;
    const uint32_t w = uint32_t(i % kBinWords);
    P.bins[i] = (w >= 1u && w <= 3u) ? 0xffffffffu : 0u;
}

// determineBucketIndex (src/extra.cpp:582-607) for every primitive of every span the heuristic will price
__global__ __launch_bounds__(kBlockR) void ref_bins_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    if (pos >= P.n) return;
    const uint32_t b = P.segBegin[pos];
    if (b == kClosed) return;
    const uint32_t cnt = P.segCount[b];
    if (cnt <= 4u || (P.segFlag[b] & 2u)) return;
    const uint32_t nBins = cnt < (uint32_t)kSahBins ? cnt : (uint32_t)kSahBins;
    const float first = unordered((uint32_t)P.keyOut[b]), len = unordered((uint32_t)P.keyOut[b + cnt - 1]) - first;
    const float q = float(nBins) * (unordered((uint32_t)P.keyOut[pos]) - first) / len;
    if (!(q >= 0.0f && q < 1.0e9f)) { // the host decides what the reference's conversion of such a value means
        P.decision[b] = kRefBuildHost;
        return;
    }
    uint32_t bin = (uint32_t)q;
    if (bin >= nBins) bin = nBins - 1u;
    uint32_t* w = P.bins + (size_t(b / 5u) * kSahBins + bin) * kBinWords;
    atomicAdd(w, 1u);
    const uint32_t t = P.idxOut[pos];
    const float4 l = P.lo[t], h = P.hi[t];
    const float f[6] = { l.x, l.y, l.z, h.x, h.y, h.z };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (f[k] == f[k]) atomicMin(w + 1 + k, ordered(f[k]));
        if (f[3 + k] == f[3 + k]) atomicMax(w + 4 + k, ordered(f[3 + k]));
    }
}

// splitPrimitivesBySAHBin (src/extra.cpp:623-699) from the bins of a sorted span: the left size of the cheapest split, 0 for
// "no split".  The boxes either side of a split are unions of whole bins; min / max are exact, so the areas are the floats
// the reference gets from the primitives one by one.
__device__ uint32_t sah_split(const RefBuildParams& P, const uint32_t b, const uint32_t n, const float* lower, const float* upper)
{
    const uint32_t nBins = n < (uint32_t)kSahBins ? n : (uint32_t)kSahBins;
    uint32_t* const bins = P.bins + size_t(b / 5u) * kSahBins * kBinWords;
    {
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        uint32_t count = 0u;
        for (uint32_t bin = 0; bin < nBins; bin++) {
            uint32_t* const w = bins + bin * kBinWords;
            w[7] = count;
            w[8] = __float_as_uint(count ? surface_area(lo, hi) : 0.0f);
            if (w[0]) {
                for (int k = 0; k < 3; k++) lo[k] = gmin(lo[k], unordered(w[1 + k])), hi[k] = gmax(hi[k], unordered(w[4 + k]));
                count += w[0];
            }
        }
    }
    const float outer = surface_area(lower, upper);
    float minCost = FLT_MAX;
    uint32_t minIndex = 0u;
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t bin = nBins - 1u; bin >= 1u; bin--) { // from the right, <=: the leftmost of equal costs wins, a NaN never does
        const uint32_t* w = bins + bin * kBinWords;
        if (w[0])
            for (int k = 0; k < 3; k++) lo[k] = gmin(lo[k], unordered(w[1 + k])), hi[k] = gmax(hi[k], unordered(w[4 + k]));
        const uint32_t leftSize = w[7];
        const float cost = 1.5f + (__uint_as_float(w[8]) * float(leftSize) + surface_area(lo, hi) * float(n - leftSize)) / outer;
        if (cost <= minCost) minCost = cost, minIndex = leftSize;
    }
    if (minCost >= float(n)) return 0u;
    return minIndex;
}

__global__ __launch_bounds__(kBlockR) void ref_decide_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    if (pos >= P.n || P.segBegin[pos] != pos) return; // one thread per open span: the one at its first position
    const uint32_t b = pos, n = P.segCount[b], node = P.segNode[b];
    float lower[3], upper[3];
    load_box(P, b, lower, upper);
    const bool ambiguous = P.segAmbig[b] != 0u;     // the order this span was given may not be the reference's
    const uint32_t anc = ambiguous ? P.segAnc[b] : node; // ... and this record's span is the nearest whose order was
    const uint32_t flag = n > 4u ? P.segFlag[b] : 0u;
    // what would show an unknown order (see the head of this file)
    bool harm = (flag & 2u) != 0u || (P.sah && n > 4u && P.decision[b] == kRefBuildHost) || (n <= 4u && ambiguous);
    const uint32_t zm = P.zeroMask[b];
    for (int k = 0; k < 3; k++) {
        if (lower[k] == 0.0f && ((zm >> (2 * k)) & 3u) == 3u) harm = true;
        if (upper[k] == 0.0f && ((zm >> (2 * (3 + k))) & 3u) == 3u) harm = true;
    }
    uint32_t split = 0u;
    if (!harm && n > 4u) {
        split = P.sah ? sah_split(P, b, n, lower, upper) : (n + 1u) / 2u;
        if (flag & 1u) {
            if (split == 0u) harm = true;
            else if (unordered((uint32_t)P.keyOut[b + split - 1u]) == unordered((uint32_t)P.keyOut[b + split])) harm = true;
        }
    }
    RefBuildNode out;
    for (int k = 0; k < 3; k++) out.lo[k] = lower[k], out.hi[k] = upper[k];
    out.begin = b, out.count = n, out.split = 0u, out.left = 0u, out.level = P.level, out.pad = 0u;
    uint32_t decision = kDecideLeaf;
    if (harm) {
        if (anc < P.maxNodes) P.hostMark[anc] = 1u;
    } else if (split != 0u) {
        const uint32_t left = atomicAdd(P.counters + 0, 2u);
        if (left + 1u < P.maxNodes) {
            out.split = split, out.left = left;
            decision = split;
            reset_span(P, b, split, left, anc);
            reset_span(P, b + split, n - split, left + 1u, anc);
            atomicAdd(P.counters + kLevelCounters + P.level + 1u, 2u);
        } else { // cannot happen for spans of at least one primitive each; the host then builds this span
            P.hostMark[node] = 1u;
        }
    }
    P.decision[b] = decision;
    if (node < P.maxNodes) P.tree[node] = out;
}

__global__ __launch_bounds__(kBlockR) void ref_apply_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    if (pos >= P.n) return;
    const uint32_t b = P.segBegin[pos];
    if (b == kClosed) return;
    const uint32_t d = P.decision[b];
    if (d == kDecideLeaf) {
        P.segBegin[pos] = kClosed;
        return;
    }
    const uint32_t child = pos < b + d ? b : b + d;
    P.segBegin[pos] = child;
    if (pos + 1u < P.n && pos + 1u != b + d) { // two equal keys side by side in one child: the order the child is given is std::sort's
        const uint2 k0 = reinterpret_cast<const uint2*>(P.keyOut)[pos], k1 = reinterpret_cast<const uint2*>(P.keyOut)[pos + 1u];
        if (k0.y == b && k1.y == b && unordered(k0.x) == unordered(k1.x)) P.segAmbig[child] = 1u;
    }
}

__global__ __launch_bounds__(kBlockR) void ref_decision_clear_kernel(const RefBuildParams P)
{
    const uint32_t pos = blockIdx.x * kBlockR + threadIdx.x;
    if (pos < P.n) P.decision[pos] = 0u;
}

#define RT_REF_HIP(call)              \
    do {                              \
        const hipError_t e_ = (call); \
        if (e_ != hipSuccess) {       \
            ok = false;               \
            goto done;                \
        }                             \
    } while (0)

} // namespace

// Runs the level passes for `n` primitives on the current device.  On success `tree` holds one record per span (record 0 = the
// root span; see RefBuildNode; split == kRefBuildHost on the spans the host has to rebuild) and `order` the primitive order the
// records refer to: for a span of the device's the order of its leaves, for a span left to the host the order that span was
// given.  False when the device cannot be used or the tree did not close within the level limit: the caller builds on the host.
bool device_build_reference_spans(const RefPrimitive* prims, uint32_t n, bool sah, std::vector<RefBuildNode>& tree, std::vector<uint32_t>& order,
                                  uint32_t info[4], float* kernelMs)
{
    bool ok = true;
    int devCount = 0;
    if (n <= 4u || n >= 0x7fffffffu || hipGetDeviceCount(&devCount) != hipSuccess || devCount == 0) return false;
    const uint32_t maxNodes = 2u * n + 2u;
    const size_t binWords = sah ? size_t(n / 5u + 1u) * kSahBins * kBinWords : 0;
    uint32_t posBits = 1;
    while ((1ull << posBits) < n) posBits++;
    auto pad = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    size_t sortBytes = 0;
    uint8_t* blob = nullptr;
    std::vector<uint32_t*> levelIdx; // the primitive order each level was given (levelIdx[L + 1] = level L's sorted output)
    std::vector<void*> levelBlocks;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<uint32_t> counters(kLevelCounters + kMaxRefLevels + 2, 0u);
    std::vector<uint32_t> marks;
    RefBuildParams P {};
    uint32_t levels = 0;
    {
        unsigned long long* nk = nullptr;
        uint32_t* nv = nullptr;
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, nk, nk, nv, nv, (int)n, 0, int(32 + posBits), nullptr) != hipSuccess) return false;
    }
    {
        const size_t sizes[] = { pad(size_t(n) * sizeof(RefPrimitive)), pad(size_t(n) * 16), pad(size_t(n) * 16), pad(size_t(n) * 16), // prims lo hi cent
                                 pad(size_t(n) * 8), pad(size_t(n) * 8),                                                               // key x2
                                 pad(size_t(n) * 4), pad(size_t(n) * 4), pad(size_t(n) * 4), pad(size_t(n) * 4), pad(size_t(n) * 4),   // segBegin count node flag zero
                                 pad(size_t(n) * 4), pad(size_t(n) * 4),                                                               // ambig anc
                                 pad(size_t(n) * 4), pad(size_t(n) * 24), pad(binWords * 4), pad(size_t(maxNodes) * sizeof(RefBuildNode)),
                                 pad(counters.size() * 4), pad(size_t(maxNodes) * 4), pad(sortBytes) };
        size_t total = 0;
        for (size_t s : sizes) total += s;
        RT_REF_HIP(hipMalloc(reinterpret_cast<void**>(&blob), total));
        uint8_t* p = blob;
        int k = 0;
        auto take = [&]() { uint8_t* r = p; p += sizes[k++]; return r; };
        P.prims = take();
        P.lo = reinterpret_cast<float4*>(take()), P.hi = reinterpret_cast<float4*>(take()), P.cent = reinterpret_cast<float4*>(take());
        P.keyIn = reinterpret_cast<unsigned long long*>(take()), P.keyOut = reinterpret_cast<unsigned long long*>(take());
        P.segBegin = reinterpret_cast<uint32_t*>(take()), P.segCount = reinterpret_cast<uint32_t*>(take());
        P.segNode = reinterpret_cast<uint32_t*>(take()), P.segFlag = reinterpret_cast<uint32_t*>(take()), P.zeroMask = reinterpret_cast<uint32_t*>(take());
        P.segAmbig = reinterpret_cast<uint32_t*>(take()), P.segAnc = reinterpret_cast<uint32_t*>(take());
        P.decision = reinterpret_cast<uint32_t*>(take()), P.box = reinterpret_cast<uint32_t*>(take()), P.bins = reinterpret_cast<uint32_t*>(take());
        P.tree = reinterpret_cast<RefBuildNode*>(take());
        P.counters = reinterpret_cast<uint32_t*>(take()), P.hostMark = reinterpret_cast<uint32_t*>(take());
        void* sortTemp = take();
        P.n = n, P.maxNodes = maxNodes, P.sah = sah ? 1u : 0u, P.level = 0u;
        const dim3 grid((n + kBlockR - 1) / kBlockR), block(kBlockR);
        // levels launched before the first look at the open-span counters: the median split's depth follows from n alone
        uint32_t firstChunk = 8u;
        if (!sah) {
            firstChunk = 1u;
            for (uint32_t s = n; s > 4u; s = (s + 1u) / 2u) firstChunk++;
        }
        auto more_levels = [&](uint32_t count) { // `count` more copies of the primitive order
            void* block = nullptr;
            if (hipMalloc(&block, pad(size_t(n) * 4) * count) != hipSuccess) return false;
            levelBlocks.push_back(block);
            for (uint32_t c = 0; c < count; c++) levelIdx.push_back(reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(block) + pad(size_t(n) * 4) * c));
            return true;
        };
        if (!more_levels(firstChunk + 1u)) {
            ok = false;
            goto done;
        }
        RT_REF_HIP(hipMemcpy(const_cast<uint8_t*>(P.prims), prims, size_t(n) * sizeof(RefPrimitive), hipMemcpyHostToDevice));
        counters[0] = 1u, counters[kLevelCounters] = 1u; // the root span: record 0
        RT_REF_HIP(hipMemcpy(P.counters, counters.data(), counters.size() * 4, hipMemcpyHostToDevice));
        RT_REF_HIP(hipMemsetAsync(P.hostMark, 0, size_t(maxNodes) * 4, nullptr));
        RT_REF_HIP(hipEventCreate(&ev0));
        RT_REF_HIP(hipEventCreate(&ev1));
        RT_REF_HIP(hipEventRecord(ev0, nullptr));
        P.idxIn = levelIdx[0], P.idxOut = levelIdx[1];
        hipLaunchKernelGGL(ref_prim_kernel, grid, block, 0, nullptr, P);
        hipLaunchKernelGGL(ref_root_kernel, dim3(1), dim3(64), 0, nullptr, P);
        bool closed = false;
        while (!closed && levels < (uint32_t)kMaxRefLevels) {
            const uint32_t chunk = levels == 0 ? firstChunk : 4u;
            if (levels != 0 && !more_levels(chunk)) {
                ok = false;
                goto done;
            }
            for (uint32_t k2 = 0; k2 < chunk && levels < (uint32_t)kMaxRefLevels; k2++, levels++) {
                P.level = levels;
                P.idxIn = levelIdx[levels], P.idxOut = levelIdx[levels + 1];
                hipLaunchKernelGGL(ref_bounds_kernel, grid, block, 0, nullptr, P);
                hipLaunchKernelGGL(ref_keys_kernel, grid, block, 0, nullptr, P);
                size_t tb = sortBytes;
                RT_REF_HIP(hipcub::DeviceRadixSort::SortPairs(sortTemp, tb, P.keyIn, P.keyOut, P.idxIn, P.idxOut, (int)n, 0, int(32 + posBits), nullptr));
                hipLaunchKernelGGL(ref_ties_kernel, grid, block, 0, nullptr, P);
                if (sah) {
                    const size_t words = binWords;
                    hipLaunchKernelGGL(ref_decision_clear_kernel, grid, block, 0, nullptr, P);
                    hipLaunchKernelGGL(ref_bins_clear_kernel, dim3((unsigned)((words + kBlockR - 1) / kBlockR)), block, 0, nullptr, P);
                    hipLaunchKernelGGL(ref_bins_kernel, grid, block, 0, nullptr, P);
                }
                hipLaunchKernelGGL(ref_decide_kernel, grid, block, 0, nullptr, P);
                hipLaunchKernelGGL(ref_apply_kernel, grid, block, 0, nullptr, P);
            }
            RT_REF_HIP(hipGetLastError());
            RT_REF_HIP(hipMemcpy(counters.data(), P.counters, counters.size() * 4, hipMemcpyDeviceToHost)); // waits for the launches
            closed = counters[kLevelCounters + levels] == 0u;
        }
        RT_REF_HIP(hipEventRecord(ev1, nullptr));
        RT_REF_HIP(hipEventSynchronize(ev1));
        if (kernelMs) RT_REF_HIP(hipEventElapsedTime(kernelMs, ev0, ev1));
        if (!closed || counters[0] > maxNodes) {
            ok = false;
            goto done;
        }
        const uint32_t nodes = counters[0];
        tree.resize(nodes), order.resize(n), marks.resize(nodes);
        RT_REF_HIP(hipMemcpy(tree.data(), P.tree, size_t(nodes) * sizeof(RefBuildNode), hipMemcpyDeviceToHost));
        RT_REF_HIP(hipMemcpy(marks.data(), P.hostMark, size_t(nodes) * 4, hipMemcpyDeviceToHost));
        RT_REF_HIP(hipMemcpy(order.data(), levelIdx[levels], size_t(n) * 4, hipMemcpyDeviceToHost));
        // the marked spans that are not below another marked span: each gets the order it was given
        uint32_t hostSpans = 0;
        std::vector<uint32_t> todo { 0u };
        while (!todo.empty()) {
            const uint32_t t = todo.back();
            todo.pop_back();
            RefBuildNode& T = tree[t];
            if (marks[t]) {
                T.split = kRefBuildHost;
                hostSpans++;
                if (T.level > levels || uint64_t(T.begin) + T.count > n) {
                    ok = false;
                    goto done;
                }
                RT_REF_HIP(hipMemcpy(order.data() + T.begin, levelIdx[T.level] + T.begin, size_t(T.count) * 4, hipMemcpyDeviceToHost));
            } else if (T.split != 0u) {
                if (T.left + 1u >= nodes) {
                    ok = false;
                    goto done;
                }
                todo.push_back(T.left), todo.push_back(T.left + 1u);
            }
        }
        if (info) info[0] = nodes, info[1] = hostSpans, info[2] = levels, info[3] = 0u;
    }
done:
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    (void)hipFree(blob);
    for (void* b2 : levelBlocks) (void)hipFree(b2);
    if (!ok) (void)hipGetLastError();
    return ok;
}

} // namespace rt
