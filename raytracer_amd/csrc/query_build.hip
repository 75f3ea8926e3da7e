// Device build of the query tree's hierarchy: binned surface-area heuristic, top down, one level per launch and one
// workgroup per node of the level.
//
// The query tree is the library's own second tree over the scene's triangles (see QueryTree): its shape carries no
// parity constraint - a query's result is accepted only with a proof that the reference's walk reports the same
// triangle - so it is built for traversal cost alone, and may be built anywhere.  The reference's own tree
// (BVH::buildRecursive, src/bvh.cpp:429-512: median split, std::sort order) stays on the host: its node and primitive
// arrays are what BVHInterface::nodes()/primitives() hand out.
//
// The split rule is the host builder's (QueryBuilder::build): 16 bins per axis over the centroid bounds, cost =
// area(left) * count(left) + area(right) * count(right), the first minimum in (axis, bin) order; a node whose
// centroids give no plane is halved where it stands.  Bin bounds and counts do not depend on the order in which
// triangles are met, so the tree has the host builder's topology wherever that rule decides.
//
// A workgroup owns its node's range of the triangle permutation: bounds (pass A), bins in LDS (pass B), the best
// plane, a stable partition into a scratch array (pass C) and the copy back.  Children with more than `leafMax`
// triangles get a node index and a place in the next level's queue.  The top levels run on few workgroups (the root
// on one: 87,000 triangles take it about a millisecond); from the eighth level on the whole chip is busy.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <vector>

#include "rt_types.h"

namespace rt {
namespace {

constexpr int kSmallBlock = 256, kLargeBlock = 1024, kLargeLevels = 6; // levels 0 .. 5 (at most 32 nodes) run with the large workgroup
constexpr int kBins = 16;
constexpr int kMaxLevels = 64;

struct Work {
    uint32_t b, e, node;
};

struct BuildParams {
    const float4* lo; // per triangle: box corners (w unused)
    const float4* hi;
    uint32_t* idx;     // the permutation being refined
    uint32_t* scratch; // same size
    DeviceBuildNode* tree;
    const Work* in;
    Work* out;
    uint32_t* counters; // [0] nodes allocated, [1 + level] work items of that level
    uint32_t level, leafMax, maxNodes, maxWork;
};

// floats as unsigned integers of the same order (for atomicMin / atomicMax in LDS)
__device__ __forceinline__ uint32_t ordered(const float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(const uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ float half_area(const float lx, const float ly, const float lz, const float hx, const float hy, const float hz)
{
    const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
    return ex * ey + ey * ez + ez * ex;
}
__device__ __forceinline__ int bin_of(const float c, const float c0, const float scale)
{
    int k = int((c - c0) * scale);
    return k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
}

// BLOCK threads per node: the few nodes of the top levels hold tens of thousands of triangles each and are walked by one
// workgroup, which is bound by the latency of its gathers - sixteen waves hide four times as much of it as four.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void sah_level_kernel(const BuildParams P)
{
    constexpr int kBuildBlock = BLOCK;
    __shared__ uint32_t s_box[12];                 // ordered: box lo[3], box hi[3], centroid lo[3], centroid hi[3]
    __shared__ uint32_t s_binN[3][kBins];
    __shared__ uint32_t s_binLo[3][kBins][3], s_binHi[3][kBins][3];
    __shared__ float s_cost[48];
    __shared__ uint32_t s_scan[kBuildBlock / 64];
    __shared__ uint32_t s_run[2], s_pick[2], s_child[2];
    const uint32_t nWork = min(P.counters[1 + P.level], P.maxWork);
    if (blockIdx.x >= nWork) return;
    const Work w = P.in[blockIdx.x];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t n = w.e - w.b;

    // ---- pass A: bounds of the node's triangles and of their centroids ----
    if (tid < 12) s_box[tid] = (tid % 6) < 3 ? 0xffffffffu : 0u;
    if (tid < 48) s_binN[tid / kBins][tid % kBins] = 0;
    for (uint32_t i = tid; i < 3 * kBins * 3; i += kBuildBlock) (&s_binLo[0][0][0])[i] = 0xffffffffu, (&s_binHi[0][0][0])[i] = 0u;
    __syncthreads();
    {
        float bl[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, bh[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        float cl[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, ch[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (uint32_t i = w.b + tid; i < w.e; i += kBuildBlock) {
            const uint32_t t = P.idx[i];
            const float4 l = P.lo[t], h = P.hi[t];
            const float lv[3] = { l.x, l.y, l.z }, hv[3] = { h.x, h.y, h.z };
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float c = (lv[a] + hv[a]) * 0.5f;
                bl[a] = fminf(bl[a], lv[a]), bh[a] = fmaxf(bh[a], hv[a]);
                cl[a] = fminf(cl[a], c), ch[a] = fmaxf(ch[a], c);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            for (int off = 32; off > 0; off >>= 1) {
                bl[a] = fminf(bl[a], __shfl_xor(bl[a], off)), bh[a] = fmaxf(bh[a], __shfl_xor(bh[a], off));
                cl[a] = fminf(cl[a], __shfl_xor(cl[a], off)), ch[a] = fmaxf(ch[a], __shfl_xor(ch[a], off));
            }
            if (lane == 0) {
                atomicMin(&s_box[a], ordered(bl[a])), atomicMax(&s_box[3 + a], ordered(bh[a]));
                atomicMin(&s_box[6 + a], ordered(cl[a])), atomicMax(&s_box[9 + a], ordered(ch[a]));
            }
        }
    }
    __syncthreads();
    float c0[3], scale[3];
    bool axisOk[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        c0[a] = unordered(s_box[6 + a]);
        const float c1 = unordered(s_box[9 + a]);
        axisOk[a] = c1 > c0[a];
        scale[a] = axisOk[a] ? float(kBins) / (c1 - c0[a]) : 0.0f;
    }

    // ---- pass B: the bins of all three axes ----
    for (uint32_t i = w.b + tid; i < w.e; i += kBuildBlock) {
        const uint32_t t = P.idx[i];
        const float4 l = P.lo[t], h = P.hi[t];
        const float lv[3] = { l.x, l.y, l.z }, hv[3] = { h.x, h.y, h.z };
        const uint32_t ol[3] = { ordered(l.x), ordered(l.y), ordered(l.z) }, oh[3] = { ordered(h.x), ordered(h.y), ordered(h.z) };
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (!axisOk[a]) continue;
            const int k = bin_of((lv[a] + hv[a]) * 0.5f, c0[a], scale[a]);
            atomicAdd(&s_binN[a][k], 1u);
#pragma unroll
            for (int d = 0; d < 3; d++) atomicMin(&s_binLo[a][k][d], ol[d]), atomicMax(&s_binHi[a][k][d], oh[d]);
        }
    }
    __syncthreads();

    // ---- the best plane: thread (axis, k) prices the split between bin k-1 and bin k ----
    if (tid < 48) {
        const int a = tid / kBins, k = tid % kBins;
        float cost = FLT_MAX;
        if (k >= 1 && axisOk[a]) {
            float ll[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, lh[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            float rl[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, rh[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            uint32_t ln = 0, rn = 0;
            for (int j = 0; j < kBins; j++) {
                const uint32_t cnt = s_binN[a][j];
                if (cnt == 0) continue;
                for (int d = 0; d < 3; d++) {
                    const float lo = unordered(s_binLo[a][j][d]), hi = unordered(s_binHi[a][j][d]);
                    if (j < k) ll[d] = fminf(ll[d], lo), lh[d] = fmaxf(lh[d], hi);
                    else rl[d] = fminf(rl[d], lo), rh[d] = fmaxf(rh[d], hi);
                }
                if (j < k) ln += cnt;
                else rn += cnt;
            }
            if (ln != 0 && rn != 0)
                cost = half_area(ll[0], ll[1], ll[2], lh[0], lh[1], lh[2]) * float(ln) + half_area(rl[0], rl[1], rl[2], rh[0], rh[1], rh[2]) * float(rn);
        }
        s_cost[tid] = cost;
    }
    __syncthreads();
    if (tid == 0) {
        float best = FLT_MAX;
        int pick = -1;
        for (int i = 0; i < 48; i++)
            if (s_cost[i] < best) best = s_cost[i], pick = i; // the first minimum in (axis, bin) order
        uint32_t nLeft = n / 2; // no plane: the range is halved where it stands
        if (pick >= 0) {
            nLeft = 0;
            for (int j = 0; j < pick % kBins; j++) nLeft += s_binN[pick / kBins][j];
        }
        s_pick[0] = (uint32_t)pick, s_pick[1] = nLeft;
        s_run[0] = 0, s_run[1] = 0;
    }
    __syncthreads();
    const int pick = (int)s_pick[0];
    const uint32_t nLeft = s_pick[1];

    // ---- pass C: stable partition into the scratch array, then back ----
    if (pick >= 0) {
        const int axis = pick / kBins, bin = pick % kBins;
        const float ac0 = axis == 0 ? c0[0] : (axis == 1 ? c0[1] : c0[2]), asc = axis == 0 ? scale[0] : (axis == 1 ? scale[1] : scale[2]);
        for (uint32_t base = w.b; base < w.e; base += kBuildBlock) {
            const uint32_t i = base + tid;
            const bool valid = i < w.e;
            uint32_t t = 0;
            bool left = false;
            if (valid) {
                t = P.idx[i];
                const float4 l = P.lo[t], h = P.hi[t];
                const float c = axis == 0 ? (l.x + h.x) * 0.5f : (axis == 1 ? (l.y + h.y) * 0.5f : (l.z + h.z) * 0.5f);
                left = bin_of(c, ac0, asc) < bin;
            }
            const unsigned long long lm = __ballot(valid && left), vm = __ballot(valid);
            if (lane == 0) s_scan[wv] = (uint32_t)__popcll(lm) | ((uint32_t)__popcll(vm) << 16);
            __syncthreads();
            uint32_t leftBefore = 0, validBefore = 0, leftAll = 0, validAll = 0;
            for (uint32_t k = 0; k < kBuildBlock / 64; k++) {
                const uint32_t v = s_scan[k];
                if (k < wv) leftBefore += v & 0xffffu, validBefore += v >> 16;
                leftAll += v & 0xffffu, validAll += v >> 16;
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t lrank = leftBefore + (uint32_t)__popcll(lm & below), vrank = validBefore + (uint32_t)__popcll(vm & below);
            const uint32_t runL = s_run[0], runR = s_run[1];
            if (valid) {
                if (left) P.scratch[w.b + runL + lrank] = t;
                else P.scratch[w.b + nLeft + runR + (vrank - lrank)] = t;
            }
            __syncthreads();
            if (tid == 0) s_run[0] = runL + leftAll, s_run[1] = runR + (validAll - leftAll);
            __syncthreads();
        }
        for (uint32_t i = w.b + tid; i < w.e; i += kBuildBlock) P.idx[i] = P.scratch[i];
    }

    // ---- the node and its children ----
    if (tid < 2) {
        const uint32_t cb = tid == 0 ? w.b : w.b + nLeft, ce = tid == 0 ? w.b + nLeft : w.e;
        const uint32_t cn = ce - cb;
        uint32_t ref;
        if (cn <= P.leafMax) {
            ref = kLeafBit | (cn << kRefCountShift) | cb;
        } else {
            ref = atomicAdd(P.counters + 0, 1u);
            const uint32_t slot = atomicAdd(P.counters + 2 + P.level, 1u);
            if (ref < P.maxNodes && slot < P.maxWork) P.out[slot] = Work { cb, ce, ref };
        }
        s_child[tid] = ref;
    }
    __syncthreads();
    if (tid == 0 && w.node < P.maxNodes) {
        DeviceBuildNode nd;
        for (int a = 0; a < 3; a++) nd.lo[a] = unordered(s_box[a]), nd.hi[a] = unordered(s_box[3 + a]);
        nd.child[0] = s_child[0], nd.child[1] = s_child[1];
        P.tree[w.node] = nd;
    }
}

#define RT_BUILD_HIP(call)                    \
    do {                                      \
        const hipError_t e_ = (call);         \
        if (e_ != hipSuccess) {               \
            ok = false;                       \
            goto done;                        \
        }                                     \
    } while (0)

} // namespace

// Builds the hierarchy over `n` triangle boxes (lo4 / hi4: 4 floats per triangle, the fourth unused; host arrays, or with
// `resident` arrays already on the current device) on the current device.
// On success `tree` holds the nodes (node 0 = root; a child is a node index or a leaf reference kLeafBit | count << 25 |
// first place in `order`), `order` the triangles in leaf order and `depth` the number of node levels.  False when the
// device cannot be used, the tree came out deeper than 64 levels or an allocation failed: the caller builds on the host.
bool device_build_query_hierarchy(const float* lo4, const float* hi4, bool resident, uint32_t n, uint32_t leafMax, std::vector<DeviceBuildNode>& tree,
                                  std::vector<uint32_t>& order, uint32_t& depth, float* kernelMs)
{
    bool ok = true;
    float4 *dLo = nullptr, *dHi = nullptr;
    uint32_t *dIdx = nullptr, *dScratch = nullptr, *dCounters = nullptr;
    DeviceBuildNode* dTree = nullptr;
    Work* dWork[2] = { nullptr, nullptr };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const uint32_t maxNodes = n, maxWork = n / (leafMax + 1) + 2;
    std::vector<uint32_t> counters(2 + kMaxLevels + 1, 0u);
    std::vector<uint32_t> iota(n);
    int devCount = 0;
    if (n <= leafMax || hipGetDeviceCount(&devCount) != hipSuccess || devCount == 0) return false;
    for (uint32_t i = 0; i < n; i++) iota[i] = i;
    if (!resident) {
        RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dLo), size_t(n) * 16));
        RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dHi), size_t(n) * 16));
    }
    RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dIdx), size_t(n) * 4));
    RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dScratch), size_t(n) * 4));
    RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dTree), size_t(maxNodes) * sizeof(DeviceBuildNode)));
    RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dWork[0]), size_t(maxWork) * sizeof(Work)));
    RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dWork[1]), size_t(maxWork) * sizeof(Work)));
    RT_BUILD_HIP(hipMalloc(reinterpret_cast<void**>(&dCounters), counters.size() * 4));
    if (!resident) {
        RT_BUILD_HIP(hipMemcpy(dLo, lo4, size_t(n) * 16, hipMemcpyHostToDevice));
        RT_BUILD_HIP(hipMemcpy(dHi, hi4, size_t(n) * 16, hipMemcpyHostToDevice));
    }
    RT_BUILD_HIP(hipMemcpy(dIdx, iota.data(), size_t(n) * 4, hipMemcpyHostToDevice));
    counters[0] = 1, counters[1] = 1; // the root: node 0, the only work item of level 0
    RT_BUILD_HIP(hipMemcpy(dCounters, counters.data(), counters.size() * 4, hipMemcpyHostToDevice));
    {
        const Work root { 0u, n, 0u };
        RT_BUILD_HIP(hipMemcpy(dWork[0], &root, sizeof(root), hipMemcpyHostToDevice));
    }
    RT_BUILD_HIP(hipEventCreate(&ev0));
    RT_BUILD_HIP(hipEventCreate(&ev1));
    RT_BUILD_HIP(hipEventRecord(ev0, nullptr));
    for (int level = 0; level < kMaxLevels; level++) {
        BuildParams P;
        P.lo = resident ? reinterpret_cast<const float4*>(lo4) : dLo, P.hi = resident ? reinterpret_cast<const float4*>(hi4) : dHi, P.idx = dIdx, P.scratch = dScratch, P.tree = dTree;
        P.in = dWork[level & 1], P.out = dWork[(level + 1) & 1];
        P.counters = dCounters;
        P.level = (uint32_t)level, P.leafMax = leafMax, P.maxNodes = maxNodes, P.maxWork = maxWork;
        // a level holds at most 2^level nodes, and never more than there are ranges above the leaf size
        const uint32_t grid = level < 20 ? std::min<uint32_t>(1u << level, maxWork) : maxWork;
        if (level < kLargeLevels) hipLaunchKernelGGL(sah_level_kernel<kLargeBlock>, dim3(grid), dim3(kLargeBlock), 0, nullptr, P);
        else hipLaunchKernelGGL(sah_level_kernel<kSmallBlock>, dim3(grid), dim3(kSmallBlock), 0, nullptr, P);
    }
    RT_BUILD_HIP(hipGetLastError());
    RT_BUILD_HIP(hipEventRecord(ev1, nullptr));
    RT_BUILD_HIP(hipMemcpy(counters.data(), dCounters, counters.size() * 4, hipMemcpyDeviceToHost)); // waits for the launches
    if (kernelMs) RT_BUILD_HIP(hipEventElapsedTime(kernelMs, ev0, ev1));
    {
        const uint32_t nodes = counters[0];
        depth = 0;
        for (int level = 0; level <= kMaxLevels; level++) {
            if (counters[1 + level] > maxWork) ok = false;
            if (counters[1 + level]) depth = (uint32_t)level + 1;
        }
        if (nodes > maxNodes || counters[1 + kMaxLevels] != 0) ok = false; // ranges were still open after the last level
        if (!ok) goto done;
        tree.resize(nodes), order.resize(n);
        RT_BUILD_HIP(hipMemcpy(tree.data(), dTree, size_t(nodes) * sizeof(DeviceBuildNode), hipMemcpyDeviceToHost));
        RT_BUILD_HIP(hipMemcpy(order.data(), dIdx, size_t(n) * 4, hipMemcpyDeviceToHost));
    }
done:
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    (void)hipFree(dLo), (void)hipFree(dHi), (void)hipFree(dIdx), (void)hipFree(dScratch), (void)hipFree(dTree);
    (void)hipFree(dWork[0]), (void)hipFree(dWork[1]), (void)hipFree(dCounters);
    if (!ok) (void)hipGetLastError();
    return ok;
}

} // namespace rt
