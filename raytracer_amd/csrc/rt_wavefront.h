// Wavefront render pipeline (included by rt_device.hip inside its anonymous namespace).
//
// The reference evaluates each pixel as one depth-first recursion (src/recursive.cpp:25-83) whose
// glossy branch draws from the pixel's sampler in recursion order (src/extra.cpp:489-505).  On a
// GPU a pixel with a full glossy tree is a chain of several hundred dependent rays.  Only part of
// that chain is really sequential:
//   * shadow rays never draw from the sampler and never change control flow: they only scale a
//     term that is added to the pixel;
//   * rays spawned at rayDepth 6 cannot spawn anything (recursion stops at rayDepth < 6), so with
//     point lights they draw nothing either;
//   * the children of a glossy loop at rayDepth 4 each consume either nothing more or exactly 2N draws,
//     so child i starts at one of i+1 known sampler offsets: all candidates are traced in one round
//     and the valid ones picked afterwards ("batched" loop).
// So the path stage walks only the sequential part - the radiance rays whose outcome decides later
// sampler offsets - in the reference's exact order, one ray (or one candidate batch) per active pixel
// per round, and emits everything else as independent work.  Every closest-hit query of the frame goes
// through the same two lean kernels (ray records in, {t, triangle} out): trace_kernel, one ray per lane,
// for stages with many rays, and group_trace_kernel, several lanes per ray, for stages with few rays and
// for the rays trace_kernel hands over.  The kernels around them only generate rays and consume hits:
//   camera_kernel       pixel rays; pixels whose ray misses the root box end here -> states, round-1 rays
//   trace kernels       rays -> hits                                  (once per stage)
//   path_logic_kernel   hits of round r -> shade tasks, leaf rays, rays of round r+1
//   leaf_logic_kernel   hits of the deferred last-level rays -> shade tasks
//   shadow_gen_kernel   shade tasks -> one shadow ray per light sample
//   shade_logic_kernel  shadow hits -> visibility, shading model, weight * radiance added to the pixel
//   resolve_kernel      pixel accumulators -> float RGB in Screen order
// Pixel sums are kept in 64-bit fixed point so that the result does not depend on the order in which
// tasks finish (integer adds commute): a frame is reproducible bit for bit, and equal to the
// reference's nested float sums to ~1e-7 relative.
#pragma once

struct RayRec { // 32 B
    float ox, oy, oz, tmax;
    float dx, dy, dz;
    uint32_t tag; // free for the producer (leaf rays: pixel)
};
struct HitRec { // 8 B
    float t;
    uint32_t tri; // kNoHit, compact triangle index, or kSphereBase + sphere
};
struct ShadeTask { // 64 B
    uint32_t pixel;   // Screen index (H-1-y)*W + x
    uint32_t matInfo; // bit 31: sphere material, low bits: mesh / sphere index
    uint32_t smp;     // sampler state at the start of this hit's light loop (area lights draw from it)
    uint32_t tri;     // the triangle hit (reference leaf order; a sphere index beyond kSphereBase): textured materials look up kd
    float W[3], px;   // weight of this hit's direct light in the pixel; hit point p
    float py, pz, dx, dy; // incoming ray direction d
    float dz, nx, ny, nz; // shading normal
};
static_assert(sizeof(ShadeTask) == 64, "ShadeTask layout");
struct LeafTask { // 48 B: a RayRec (tag = pixel) followed by the weight of its radiance in the pixel
    RayRec ray;
    float W[3];
    float pad;
};
static_assert(sizeof(LeafTask) == 48, "LeafTask layout");

constexpr double kFixScale = 1099511627776.0; // 2^40
constexpr float kFixLimit = 4.0e6f;           // |value| beyond this is recorded as non-finite

struct WaveParams {
    SceneView S;
    rt_features F;
    int W, H, startDepth;
    int tilesX, myTileRows, stripeTileRows, rank, world;
    // Several frames in one launch sequence (rt_render_frames): the frames lie one after the other in rgb / primId / primT and
    // in the pixel sums, frame f takes work items [f, f + 1) * myTileRows * tilesX * 64 and the camera cams[f]; a pixel's
    // packed position holds f * H + y in its row half.
    int frames;
    const rt_camera* cams; // device memory, `frames` entries (store_cameras_kernel)
    uint32_t nWork;
    float* rgb;
    uint32_t* primId;
    float* primT;
    unsigned long long* stats;
    uint32_t* work;          // [0] pixel fetch, [1] leaf-task fetch, [2] shade-task fetch, [3] resolve fetch
    long long* accum;        // W*H*3 fixed point
    uint32_t* nonfinite;     // W*H flag words (bit c: NaN in channel c, bit 3+c: +inf, bit 6+c: -inf)
    ShadeTask* shadeQ;
    LeafTask* leafQ;
    uint32_t* qCount;        // [0] shade tasks appended, [1] leaf tasks appended (both keep counting past capacity),
                             // [2] path states allocated
    uint32_t shadeCap, leafCap;
    float* states;           // path states: a 64-byte header per state, stateStride floats apart
    uint32_t stateStride;    // 16 in the rounds (headers packed; loop frames and pass records in arrays of their own, below);
                             // kStateDwords where a persistent launch keeps header, frames and pass records of a state together
    // rounds: the pending glossy loop of rayDepth l of state s is frameBase + (l * stateCap + s) * kFrameDwords (all states'
    // frames of one depth lie together: the states of a round mostly sit at the same depth, and neighbours in the active list
    // are neighbours in memory), its pending pass-through ray passBase + (l * stateCap + s) * kPassDwords (null without
    // transparency: nothing is allocated for them)
    float* frameBase;
    float* passBase;
    uint32_t stateCap;
    const uint32_t* activeIn; // state indices whose rays were traced this round
    uint32_t* activeOut;      // survivors
    const uint32_t* activeInCount;
    uint32_t* activeOutCount;
    // The rounds' active lists are filled from both ends: states with a batch of candidate rays in flight from the front
    // (counted by activeOutCount), states with one ray in flight from the back (entry stateCap - 1 - k, counted here), so
    // that a wave of the next round's logic kernel holds states of one kind and runs one of the two long code paths.
    const uint32_t* activeInCountBack; // null: the list has a front only (the camera kernel's)
    uint32_t* activeOutCountBack;
    RayRec* pathRays;         // rays of the next round; activeRayOut[i] = first ray of state activeOut[i]
    const RayRec* pathRaysIn; // rays of the round whose hits this kernel consumes (a different buffer)
    uint32_t* rayCountOut;    // rays appended for the next round
    const uint32_t* activeRayIn;
    uint32_t* activeRayOut;
    uint32_t rayCap;          // capacity of pathRays
    uint32_t batchRays;       // most candidate rays of one batch of a glossy loop's children, w(w+1)/2 (0 = loops are walked one child per round)
    uint32_t batchWindow;     // w: children of such a loop traced per round
    const HitRec* hits;       // hits of the rays this kernel consumes
    RayRec* shadowRays;       // shadow_gen output, raysPerTask per shade task of the chunk
    uint32_t chunkBase, chunkTasks, raysPerTask;
    uint32_t lightDraws;     // sampler draws one hit's light loop consumes (0 with point lights only)
};

__device__ __forceinline__ void accumulate_pixel(const WaveParams& P, const uint32_t pixel, const vec3 c)
{
    const float v[3] = { c.x, c.y, c.z };
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float x = v[ch];
        if (x == 0.0f) continue;
        if (fabsf(x) < kFixLimit) {
            const long long q = __double2ll_rn((double)x * kFixScale);
            atomicAdd(reinterpret_cast<unsigned long long*>(P.accum + 3 * (size_t)pixel + ch), (unsigned long long)q);
        } else {
            const uint32_t bit = (x != x) ? (1u << ch) : (x > 0.0f ? (8u << ch) : (64u << ch));
            atomicOr(P.nonfinite + pixel, bit);
            // a finite value this large does not fit the fixed-point sum: the frame is reported incomplete (rt_sync)
            if (fabsf(x) <= FLT_MAX) atomicAdd(P.stats + (blockIdx.x % kStatCopies) * kStatStride + S_BIGVALUE, 1ull);
        }
    }
}

// Appends one record per lane with wantAppend; returns false for lanes whose slot lies beyond capacity.
template <typename T>
__device__ __forceinline__ bool queue_append(T* q, uint32_t* count, const uint32_t cap, const bool want, const T& rec, const int lane)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return true;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (!want) return true;
    const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (slot >= cap) return false;
    const float4* src = reinterpret_cast<const float4*>(&rec);
    float4* dst = reinterpret_cast<float4*>(q + slot);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 16); i++) dst[i] = src[i];
    return true;
}

__device__ __forceinline__ uint32_t wave_fetch(uint32_t* counter, const bool need, const int lane)
{
    const unsigned long long mask = __ballot(need);
    if (mask == 0ull) return 0xffffffffu;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return need ? base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) : 0xffffffffu;
}

// wave_fetch for a whole workgroup: one atomic per workgroup instead of one per wave (atomics on one address
// serialise).  Every thread of the workgroup must call it; s_n (4 words) and s_base are the caller's LDS.
__device__ __forceinline__ uint32_t block_fetch(uint32_t* counter, const bool need, const int lane, uint32_t* s_n, uint32_t* s_base)
{
    const unsigned long long mask = __ballot(need);
    const int wv = threadIdx.x >> 6;
    if (lane == 0) s_n[wv] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        *s_base = n ? atomicAdd(counter, n) : 0u;
    }
    __syncthreads();
    uint32_t pos = *s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; k++) pos += s_n[k];
    __syncthreads();
    return need ? pos : 0xffffffffu;
}

// block_fetch on two counters with one trip to them.  s_n: 8 words, s_base: 2 words.
__device__ __forceinline__ void block_fetch2(uint32_t* counterA, const bool needA, uint32_t* counterB, const bool needB, const int lane,
                                             uint32_t* s_n, uint32_t* s_base, uint32_t& posA, uint32_t& posB)
{
    const unsigned long long maskA = __ballot(needA), maskB = __ballot(needB);
    const int wv = threadIdx.x >> 6;
    if (lane == 0) s_n[wv] = (uint32_t)__popcll(maskA), s_n[4 + wv] = (uint32_t)__popcll(maskB);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nA = s_n[0] + s_n[1] + s_n[2] + s_n[3], nB = s_n[4] + s_n[5] + s_n[6] + s_n[7];
        uint32_t bA = 0, bB = 0;
        if (nA) bA = atomicAdd(counterA, nA);
        if (nB) bB = atomicAdd(counterB, nB);
        s_base[0] = bA, s_base[1] = bB;
    }
    __syncthreads();
    uint32_t pA = s_base[0] + (uint32_t)__popcll(maskA & ((1ull << lane) - 1ull));
    uint32_t pB = s_base[1] + (uint32_t)__popcll(maskB & ((1ull << lane) - 1ull));
    for (int k = 0; k < wv; k++) pA += s_n[k], pB += s_n[4 + k];
    __syncthreads();
    posA = needA ? pA : 0xffffffffu, posB = needB ? pB : 0xffffffffu;
}

__device__ __forceinline__ void flush_counters(unsigned long long* stats, const int lane, const uint32_t cPrimary, const uint32_t cShadow,
                                               const uint32_t cSecondary, const TraceCounters& tc, const uint32_t cPixels, const uint32_t cDrop)
{
    const uint32_t vals[10] = { cPrimary, cShadow, cSecondary, tc.nodePairs, tc.triTests, cPixels, tc.overflow, cDrop, 0u, tc.node4 };
#pragma unroll
    for (int i = 0; i < 10; i++) {
        unsigned long long v = vals[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && v) atomicAdd(stats + (blockIdx.x % kStatCopies) * kStatStride + i, v);
    }
}

__device__ __forceinline__ void store_ray(RayRec* r, const vec3 o, const vec3 d, const float tmax, const uint32_t tag)
{
    float4* q = reinterpret_cast<float4*>(r);
    q[0] = make_float4(o.x, o.y, o.z, tmax);
    q[1] = make_float4(d.x, d.y, d.z, __uint_as_float(tag));
}

// ---------------------------------------------------------------------------------------------
// trace_kernel: the closest-hit query of the reference (intersectRayWithBVH, src/bvh.cpp:235-370)
// for an array of rays.  Persistent waves; a lane whose ray ends takes the next ray from a shared
// counter at once instead of idling until the wave's longest ray ends (ray lengths on one model
// spread from 1 to several hundred steps).  One node-pair or one leaf per step and lane; the
// traversal stack lives in LDS (STACK entries per lane), everything else in registers.
// ---------------------------------------------------------------------------------------------
struct TraceParams {
    SceneView S;
    const char* rays; // RayRec at the start of each record
    HitRec* hits;
    const uint32_t* countPtr; // number of items = clamp(*countPtr - countSub, 0, countMaxItems) * countMul
    uint32_t countSub, countMaxItems, countMul;
    uint32_t rayStride;       // bytes between ray records
    uint32_t runLo, runHi;    // the launch does nothing unless runLo <= number of rays < runHi (several variants share one ray list)
    const uint32_t* indirect; // ray / hit index of item i (null: i itself)
    uint32_t indirectFrom;    // ... used only when the launch has at least this many items (an order that was made for big stages only)
    uint32_t stepBudget;      // one-ray-per-lane launches: a ray still unfinished after this many steps is handed over (0: never)
    uint32_t* longList;       // ... by appending its index here; a launch with several lanes per ray then traces it from the root
    uint32_t* longCount;
    uint32_t* fetch;          // zeroed work counter of this launch
    uint32_t* fallList;       // query-tree launches: rays whose result could not be proven equal to the reference's are
    uint32_t* fallCount;      // appended here and traced again in the reference's tree
    unsigned long long* stats;
    uint32_t useBvh;
    unsigned long long* waveTimes; // diagnostics (RT_AMD_WAVE_TIMES): per wave of a group launch, the clock at its start and at its end (null: off)
};

// One closest-hit query in flight on a lane.
struct TraceLane {
    RaySlabs R;
    float t;
    uint32_t tri;
    uint32_t cur;  // inner node (pair index), leaf reference, or kTraversalDone
    uint32_t pend; // a leaf reached but not tested yet (0: none)
    int sp;
    float tieT;    // query tree: distance at which two triangles tied (-1: none)
    uint32_t lost; // query tree: a stack entry was dropped
};

// Starts a query.  Returns true when traversal steps remain, false when the query has already ended
// (T.t / T.tri hold its result).
template <bool COUNT>
__device__ __forceinline__ bool trace_begin(const SceneView& S, const bool useBvh, TraceLane& T, const vec3 o, const vec3 d, const float tmax,
                                            TraceCounters& tc)
{
    T.R = make_slabs(S, o, d);
    T.t = tmax;
    T.tri = kNoHit;
    T.sp = 0;
    T.pend = 0;
    T.tieT = -1.0f, T.lost = 0;
    if (useBvh) {
        float tr;
        bool outside;
        // root box against the ray's own t; a miss ends the query before the sphere loop (src/bvh.cpp:264-268)
        if (!box_test_any(S.rootLo[0], S.rootLo[1], S.rootLo[2], S.rootHi[0], S.rootHi[1], S.rootHi[2], T.R, T.t, tr, outside)) return false;
        T.cur = S.rootRef;
        return true;
    }
    // enableAccelStructure off (src/bvh.cpp:351-363): every stored triangle, array order
    for (uint32_t k = 0; k < S.nTris; k++) {
        if (COUNT) tc.triTests++;
        tri_test(S, S.bruteOrder[k], o, d, T.t, T.tri);
    }
    for (uint32_t s = 0; s < S.nSpheres; s++)
        if (sphere_test(S.spheres[s], o, d, T.t)) T.tri = kSphereBase + s;
    return false;
}

// Next pending node from the lane's stack into T.cur (kTraversalDone when the stack is empty).
template <bool CULL, int STACK>
__device__ __forceinline__ void lane_pop(TraceLane& T, const uint32_t* const stack)
{
    for (;;) {
        if (T.sp == 0) {
            T.cur = kTraversalDone;
            return;
        }
        T.sp--;
        if (T.sp >= STACK) continue; // overflowed entries were dropped (counted)
        T.cur = stack[T.sp * kBlock];
        // a box entered beyond the closest hit found since it was pushed holds nothing closer
        if (!CULL || !(__uint_as_float(stack[(STACK + T.sp) * kBlock]) > T.t)) return;
    }
}

// One inner node: both child boxes, nearer child next, the other pushed.
template <bool CULL, bool COUNT, int STACK>
__device__ __forceinline__ void lane_node_step(const SceneView& S, TraceLane& T, uint32_t* const stack, TraceCounters& tc)
{
    const float4* q = reinterpret_cast<const float4*>(S.pairs + T.cur);
    const float4 a = q[0], b = q[1], c = q[2], e = q[3];
    if (COUNT) tc.nodePairs++;
    float tl, trr;
    bool outL, outR, hl, hr;
    pair_test(a, b, c, e, T.R, hl, hr, tl, trr, outL, outR);
    const uint32_t ref0 = __float_as_uint(b.z), ref1 = __float_as_uint(e.z);
    const bool leftNear = tl < trr; // only used when both boxes are hit; equal distances: right first (src/bvh.cpp:301-310)
    if (CULL) { // the distances may be the approximate ones: cull only beyond their error bound
        hl = hl && !(outL && tl - T.R.eps > T.t);
        hr = hr && !(outR && trr - T.R.eps > T.t);
    }
    if (hl && hr) {
        if (T.sp < STACK) {
            stack[T.sp * kBlock] = leftNear ? ref1 : ref0;
            // the far child's entry distance, to be compared again when it is popped (0: never culled)
            const bool farOut = leftNear ? outR : outL;
            stack[(STACK + T.sp) * kBlock] = __float_as_uint(farOut ? (leftNear ? trr : tl) - T.R.eps : 0.0f);
        } else {
            tc.overflow++;
        }
        T.sp++;
        T.cur = leftNear ? ref0 : ref1;
    } else if (hl || hr) {
        T.cur = hl ? ref0 : ref1;
    } else {
        lane_pop<CULL, STACK>(T, stack);
    }
}

// The same for a query-tree launch: one 4-wide node, nearest child next, the others pushed farthest first.
template <bool COUNT, int STACK>
__device__ __forceinline__ void lane_node4_step(const SceneView& S, TraceLane& T, uint32_t* const stack, TraceCounters& tc)
{
    if (COUNT) tc.node4++;
    const Node4Hits h = node4_test(S.nodes4 + T.cur, T.R, T.t);
    if (h.key[0] == kMissKey) {
        lane_pop<true, STACK>(T, stack);
        return;
    }
    T.cur = node4_ref(h, h.key[0]);
#pragma unroll
    for (int k = 3; k >= 1; k--) {
        if (h.key[k] == kMissKey) continue;
        if (T.sp < STACK) {
            stack[T.sp * kBlock] = node4_ref(h, h.key[k]);
            stack[(STACK + T.sp) * kBlock] = __float_as_uint(node4_entry(h.key[k]) - T.R.eps);
        } else {
            T.lost = 1;
        }
        T.sp++;
    }
}

// One step of every lane of a wave (`has`: the lane has a query in flight).  Lanes at an inner node test it.
// A lane that reaches a leaf parks it and goes on to its next node; parked leaves are tested together, when
// kLeafBatch lanes have one or a lane cannot go on without testing its own (it reached a second leaf, or its
// stack is empty) - a wave then runs the long triangle code for many lanes at once instead of for the one or
// two that happen to be at a leaf.  Leaves are still tested in the order they are reached, so equal-distance
// ties fall as in the reference.  Returns true for lanes whose query has ended.
// Shadow queries (ANY).  visibilityOfLightSampleBinary (src/light.cpp:66-90) only asks whether the closest hit of the ray from the
// light lies within 5e-4 of the shaded point's distance D.  A ray record whose tag word holds D lets a query-tree launch answer
// that with less work: nothing beyond any_bound(D) can be the hit that makes the sample visible, so the walk starts with that
// as its closest distance (after the root test, which the reference makes with the ray's own t), and a triangle closer than
// any_occluded(D) settles the answer at once - the reference's closest hit is then at most that far, and the distance test
// fails for it as it fails for this one.  The record that comes out is {t, triangle} of A hit the reference also finds (same
// proof as for every query-tree result), or no hit: the visibility test on it gives the reference's answer.  Equal distances
// need no care here (the answer depends on t alone).  Launches that ignore the tag return the closest hit itself.
__device__ __forceinline__ float any_bound(const float D) { return (D + 2e-3f) * 1.000001f; }
__device__ __forceinline__ float any_occluded(const float D) { return (D - 2e-3f) * 0.999999f; }

#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 20
#endif
constexpr int kLeafBatch = RT_LEAF_BATCH;
#ifndef RT_STUCK_LANES
#define RT_STUCK_LANES 8
#endif
constexpr int kStuckLanes = RT_STUCK_LANES; // lanes that cannot go on without their leaf test before the wave runs it
template <bool CULL, bool COUNT, int STACK, bool QUERY, bool ANY = false>
__device__ __forceinline__ bool trace_wave_step(const SceneView& S, const bool has, TraceLane& T, uint32_t* const stack, TraceCounters& tc)
{
    if (has && T.cur != kTraversalDone && !(T.cur & kLeafBit)) {
        if (QUERY) lane_node4_step<COUNT, STACK>(S, T, stack, tc);
        else lane_node_step<CULL, COUNT, STACK>(S, T, stack, tc);
    }
    if (has && T.cur != kTraversalDone && (T.cur & kLeafBit) && T.pend == 0) {
        T.pend = T.cur | kLeafBit;
        lane_pop<CULL, STACK>(T, stack);
    }
    const bool parked = has && T.pend != 0;
    const bool stuck = parked && (T.cur == kTraversalDone || (T.cur & kLeafBit));
    const unsigned long long stuckMask = __ballot(stuck);
    // a few stuck lanes wait for company as long as other lanes of the wave still have nodes to walk
    const bool test = __popcll(stuckMask) >= kStuckLanes || (stuckMask != 0ull && __ballot(has && !stuck) == 0ull)
        || __popcll(__ballot(parked)) >= kLeafBatch;
    if (test) {
        if (parked) {
            const uint32_t first = T.pend & kRefOffsetMask, cnt = QUERY ? (T.pend >> kRefCountShift) & kMaxLeafCount : leaf_count(S, T.pend);
            leaf_test_narrow<COUNT, QUERY && !ANY>(S, first, cnt, T.R.o, T.R.d, T.t, T.tri, tc, &T.tieT);
            T.pend = 0;
            if (ANY && T.t < T.tieT) T.cur = kTraversalDone, T.sp = 0; // (tieT holds any_occluded(D) here) an occluder: the answer is settled
        }
    }
    if (has && T.cur == kTraversalDone && T.pend == 0) {
        if (!QUERY) // a query-tree launch first settles which triangle it reports (see trace_kernel)
            for (uint32_t s = 0; s < S.nSpheres; s++) // src/bvh.cpp:366-367
                if (sphere_test(S.spheres[s], T.R.o, T.R.d, T.t)) T.tri = kSphereBase + s;
        return true;
    }
    return false;
}

// Query-tree launches: is the closest triangle found what the reference's walk reports?  Yes when no second triangle
// tied with it (the reference breaks ties by the order of its own walk), no stack entry was lost, and the reference
// certainly reaches the triangle's leaf - then both walks test that triangle, and the reference, which tests a subset
// of what the conservative walk tested, can find nothing closer.  On success `tri` becomes the index in reference order.
__device__ __forceinline__ bool query_result_proven(const SceneView& S, const RaySlabs& R, const float t, uint32_t& tri, const float tieT,
                                                    const uint32_t lost)
{
    if (lost) return false;
    if (tri == kNoHit) return true; // every triangle the reference tests was tested here too
    const float4* q = reinterpret_cast<const float4*>(S.cert + tri);
    const float4 c0 = q[0], c1 = q[1];
    TriCert c;
    c.leafLo[0] = c0.x, c.leafLo[1] = c0.y, c.leafLo[2] = c0.z, c.compact = __float_as_uint(c0.w);
    c.leafHi[0] = c1.x, c.leafHi[1] = c1.y, c.leafHi[2] = c1.z, c.route = __float_as_uint(c1.w);
    tri = c.compact;
    if (tieT == t) return false;
    return leaf_surely_visited(S, c, R) || leaf_visited_exactly(S, c.route, R);
}

__device__ __forceinline__ void trace_lane_clear(TraceLane& T)
{
    T.R.o = T.R.d = T.R.r = splat(0.0f);
    T.R.fast = T.R.approx = false;
    T.R.nor = splat(0.0f), T.R.eps = 0.0f;
    T.t = 0.0f, T.tri = kNoHit, T.cur = 0, T.pend = 0, T.sp = 0;
    T.tieT = -1.0f, T.lost = 0;
}

#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES
#endif
#ifndef RT_STEPS_PER_CHECK
#define RT_STEPS_PER_CHECK 6
#endif
constexpr int kStepsPerCheck = RT_STEPS_PER_CHECK; // traversal steps between two looks at the wave's idle lanes
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 1
#endif
constexpr int kRefillLanes = RT_REFILL_LANES; // idle lanes a wave waits for before it starts new rays on them
constexpr uint32_t kLaneChunk = 64;  // rays a wave takes from the shared counter at a time
constexpr uint32_t kGroupChunk = 32; // the same for a wave whose lanes share rays

template <bool CULL, bool COUNT, int STACK, bool QUERY = false, bool ANY = false>
__global__ __launch_bounds__(kBlock) RT_TRACE_WAVES void trace_kernel(const TraceParams P)
{
    __shared__ uint32_t s_stack[2 * STACK * kBlock]; // per lane: STACK node references, then their entry distances
    __shared__ uint32_t s_long[(kBlock / 64) * 128]; // per wave: rays to hand over, not yet on the list
    uint32_t nLong = 0;
    uint32_t* const stack = s_stack + threadIdx.x;
    const SceneView& S = P.S;
    const int lane = threadIdx.x & 63;
    const uint32_t have = *P.countPtr;
    const uint32_t items = have > P.countSub ? min(have - P.countSub, P.countMaxItems) : 0u;
    const uint32_t total = items * P.countMul;
    if (total < P.runLo || total >= P.runHi) return;
    TraceCounters tc { 0, 0, 0 };
    // Work distribution.  Atomics on one address serialise (about 10 ns each on this chip), so a wave must not pay one
    // per refill: the first ray of every lane is fixed by the lane's position in the grid, and later rays are
    // taken from the shared counter in chunks of kLaneChunk that the wave then hands to its idle lanes by itself.
    const uint32_t gridLanes = gridDim.x * kBlock;
    bool has = false, exhausted = total <= gridLanes, first = true;
    uint32_t idx = 0, steps = 0, chunkNext = 0, chunkEnd = 0; // [chunkNext, chunkEnd): rays this wave holds (wave-uniform)
    const uint32_t budget = P.stepBudget; // (0 for a pass that counts the reference's walk: every ray is walked exactly once)
    TraceLane T;
    trace_lane_clear(T);
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) >= total) return; // nothing for this wave, now or later
    // A lane whose query has ended stays idle until the wave next looks for new rays (the top of the loop), so what remains
    // to be done for it - the proof that its result is the reference's, its record - waits until then as well and runs once
    // for all the lanes that ended in these steps, instead of after every step for the two or three that ended in it; and it
    // runs between the fetch of the lanes' next rays and their set-up, so that the wave waits for those fetches once.
    bool settle = false;
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        // new rays are started for several idle lanes at a time: the set-up of a ray is long and runs with only those lanes
        const bool refill = idle != 0ull && (first || !exhausted || chunkNext < chunkEnd) && (first || __popcll(idle) >= kRefillLanes);
        uint32_t idxNew = 0xffffffffu;
        bool fetched = false;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
        if (refill) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = blockIdx.x * kBlock + threadIdx.x;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.fetch, kLaneChunk);
                    base = __shfl(base, 0) + gridLanes;
                    exhausted = base + kLaneChunk >= total;
                    chunkNext = min(base, total), chunkEnd = min(base + kLaneChunk, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t avail = chunkEnd - chunkNext, n = (uint32_t)__popcll(idle);
                if (rank < avail) next = chunkNext + rank;
                chunkNext += min(n, avail);
            }
            first = false;
            if (!has) idxNew = next;
            if (!has && idxNew < total) {
                if (P.indirect && total >= P.indirectFrom) idxNew = P.indirect[idxNew]; // from here on the ray's own index
                const float4* q = reinterpret_cast<const float4*>(P.rays + (size_t)idxNew * P.rayStride);
                a = q[0], b = q[1];
                fetched = true;
            }
        }
        if (__ballot(settle) != 0ull) { // (idx and T are still those of the query that ended)
            bool unproven = false;
            if (settle) {
                if (QUERY) {
                    unproven = !query_result_proven(S, T.R, T.t, T.tri, ANY ? -1.0f : T.tieT, T.lost);
                    for (uint32_t s = 0; s < S.nSpheres; s++) // src/bvh.cpp:366-367
                        if (sphere_test(S.spheres[s], T.R.o, T.R.d, T.t)) T.tri = kSphereBase + s;
                }
                if (!unproven) P.hits[idx] = HitRec { T.t, T.tri };
            }
            if (QUERY && __ballot(unproven) != 0ull) {
                const uint32_t pos = wave_fetch(P.fallCount, unproven, lane);
                if (unproven) P.fallList[pos] = idx;
            }
            settle = false;
        }
        if (refill) {
            bool unfit = false;
            if (!has) idx = idxNew;
            if (fetched) {
                has = trace_begin<COUNT>(S, P.useBvh != 0, T, make3(a.x, a.y, a.z), make3(b.x, b.y, b.z), a.w, tc);
                steps = 0;
                if (!has) P.hits[idx] = HitRec { T.t, T.tri };
                if (QUERY && has && !T.R.approx) unfit = true, has = false; // the conservative tests need the approximate slabs
                if (ANY && has) T.t = fminf(T.t, any_bound(b.w)), T.tieT = any_occluded(b.w); // the tag word holds the shaded point's distance
            }
            if (QUERY && __ballot(unfit) != 0ull) {
                const uint32_t pos = wave_fetch(P.fallCount, unfit, lane);
                if (unfit) P.fallList[pos] = idx;
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted && chunkNext == chunkEnd) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < kStepsPerCheck; it++) {
            const bool ended = trace_wave_step<CULL, COUNT, STACK, QUERY, ANY>(S, has, T, stack, tc);
            if (ended) has = false, settle = true;
        }
        if (budget != 0) { // the few long rays would hold the whole stage at one-lane speed: hand them over
            steps += kStepsPerCheck;
            const bool giveUp = has && steps > budget && exhausted && chunkNext == chunkEnd; // while idle lanes can be refilled a long ray costs nothing
            const unsigned long long gm = __ballot(giveUp);
            if (gm != 0ull) { // collected per wave in LDS, appended to the list 64 or more at a time (the list's counter is one address)
                if (giveUp) {
                    s_long[(threadIdx.x >> 6) * 128 + nLong + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull))] = idx;
                    has = false;
                }
                nLong += (uint32_t)__popcll(gm);
                if (nLong >= 64u) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.longCount, nLong);
                    base = __shfl(base, 0);
                    for (uint32_t k = lane; k < nLong; k += 64u) P.longList[base + k] = s_long[(threadIdx.x >> 6) * 128 + k];
                    nLong = 0;
                }
            }
        }
    }
    if (nLong != 0u) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(P.longCount, nLong);
        base = __shfl(base, 0);
        for (uint32_t k = lane; k < nLong; k += 64u) P.longList[base + k] = s_long[(threadIdx.x >> 6) * 128 + k];
    }
    if (COUNT || __ballot(tc.overflow != 0) != 0ull) flush_counters(P.stats, lane, 0, 0, 0, tc, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// packet_trace_kernel: the closest-hit query for rays that travel together - the camera rays of an 8 x 8 pixel tile share
// their origin and nearly their direction, and the path stage's first ray list holds them tile by tile.  A wave takes 64
// consecutive rays as one packet and walks the query tree ONCE for all of them: one stack per wave (node references only,
// in LDS), every lane tests the popped node's four boxes - or the popped leaf's triangles - for its own ray, and a child is
// pushed when any lane's ray enters it (the children in the order of the first such lane's entry distances, nearest on top).
// The node is the same for all lanes, so its 112 bytes are fetched once per wave instead of once per lane, no lane waits
// at a different place of the tree, and there is no per-lane stack.  A lane tests boxes and triangles its own ray would
// have skipped; that only adds candidates: the result is the closest triangle of a superset of what the lane's own walk
// tests, accepted under the same proof as every query-tree result (query_result_proven), else the ray goes to the list
// of rays traced again in the caller's tree.  Packets are dealt to the waves by position (no work counter).
// ---------------------------------------------------------------------------------------------
constexpr int kPacketStack = 96; // entries per wave: a walk needs at most three per level of the 4-wide tree and one
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void packet_trace_kernel(const TraceParams P)
{
    __shared__ uint32_t s_stack[(kBlock / 64) * kPacketStack];
    uint32_t* const stack = s_stack + (threadIdx.x >> 6) * kPacketStack;
    const SceneView& S = P.S;
    const int lane = threadIdx.x & 63;
    const uint32_t have = *P.countPtr;
    const uint32_t items = have > P.countSub ? min(have - P.countSub, P.countMaxItems) : 0u;
    const uint32_t total = items * P.countMul;
    if (total < P.runLo || total >= P.runHi) return;
    TraceCounters tc { 0, 0, 0 };
    const uint32_t nPackets = (total + 63u) / 64u, gridWaves = gridDim.x * (kBlock / 64);
    for (uint32_t packet = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); packet < nPackets; packet += gridWaves) {
        const uint32_t idx = packet * 64u + (uint32_t)lane;
        TraceLane T;
        trace_lane_clear(T);
        bool alive = false, unfit = false;
        if (idx < total) {
            const float4* q = reinterpret_cast<const float4*>(P.rays + (size_t)idx * P.rayStride);
            const float4 a = q[0], b = q[1];
            alive = trace_begin<COUNT>(S, true, T, make3(a.x, a.y, a.z), make3(b.x, b.y, b.z), a.w, tc);
            if (!alive) P.hits[idx] = HitRec { T.t, T.tri };
            if (alive && !T.R.approx) unfit = true, alive = false; // the conservative tests need the approximate slabs
        }
        int sp = 0;
        uint32_t cur = __ballot(alive) != 0ull ? S.rootRef : kTraversalDone; // the same on every lane
        const bool counting = COUNT && alive && lane == __ffsll((long long)__ballot(alive)) - 1;
        bool lost = false;
        while (cur != kTraversalDone) {
            if (cur & kLeafBit) {
                if (alive) {
                    const uint32_t first = cur & kRefOffsetMask, cnt = (cur >> kRefCountShift) & kMaxLeafCount;
                    TraceCounters one { 0, 0, 0 };
                    leaf_test_narrow<COUNT, true>(S, first, cnt, T.R.o, T.R.d, T.t, T.tri, one, &T.tieT);
                    if (COUNT && counting) tc.triTests += one.triTests; // (the wave fetches a leaf's triangles once: counted once)
                }
            } else {
                if (COUNT && counting) tc.node4++; // (the wave fetches the node once: counted once)
                Node4Hits h = node4_test(S.nodes4 + cur, T.R, T.t); // (every lane: the node is the wave's; a lane without a ray enters nothing)
                uint32_t slots = 0; // the children this lane's ray enters
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (!alive) h.key[k] = kMissKey;
                    if (h.key[k] != kMissKey) slots |= 1u << (h.key[k] & 3u);
                }
                const unsigned long long entering = __ballot(slots != 0u);
                if (entering != 0ull) {
                    const int lead = __ffsll((long long)entering) - 1;
                    uint32_t any = 0; // children some lane enters
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (__ballot((slots >> k) & 1u) != 0ull) any |= 1u << k;
                    // the leader's children by its entry distances (the keys are sorted per lane), and the node's references
                    const uint32_t lk0 = (uint32_t)__shfl((int)h.key[0], lead), lk1 = (uint32_t)__shfl((int)h.key[1], lead);
                    const uint32_t lk2 = (uint32_t)__shfl((int)h.key[2], lead), lk3 = (uint32_t)__shfl((int)h.key[3], lead);
                    const uint32_t r0 = (uint32_t)__shfl((int)h.ref[0], lead), r1 = (uint32_t)__shfl((int)h.ref[1], lead);
                    const uint32_t r2 = (uint32_t)__shfl((int)h.ref[2], lead), r3 = (uint32_t)__shfl((int)h.ref[3], lead);
#define RT_REF_OF(slot) ((slot) == 0u ? r0 : ((slot) == 1u ? r1 : ((slot) == 2u ? r2 : r3)))
#define RT_PUSH(r)                                   \
    do {                                             \
        if (sp < kPacketStack) {                     \
            if (lane == 0) stack[sp] = (r);          \
        } else {                                     \
            lost = true;                             \
        }                                            \
        sp++;                                        \
    } while (0)
                    uint32_t taken = 1u << (lk0 & 3u); // the leader enters at least one child
                    if (lk1 != kMissKey) taken |= 1u << (lk1 & 3u);
                    if (lk2 != kMissKey) taken |= 1u << (lk2 & 3u);
                    if (lk3 != kMissKey) taken |= 1u << (lk3 & 3u);
                    const uint32_t others = any & ~taken;
                    // pushed first, popped last: the children only other lanes enter; then the leader's, farthest first
                    if (others & 1u) RT_PUSH(r0);
                    if (others & 2u) RT_PUSH(r1);
                    if (others & 4u) RT_PUSH(r2);
                    if (others & 8u) RT_PUSH(r3);
                    if (lk3 != kMissKey) RT_PUSH(RT_REF_OF(lk3 & 3u));
                    if (lk2 != kMissKey) RT_PUSH(RT_REF_OF(lk2 & 3u));
                    if (lk1 != kMissKey) RT_PUSH(RT_REF_OF(lk1 & 3u));
                    cur = RT_REF_OF(lk0 & 3u);
#undef RT_PUSH
#undef RT_REF_OF
                    continue;
                }
            }
            // next pending node of the packet
            for (;;) {
                if (sp == 0) {
                    cur = kTraversalDone;
                    break;
                }
                sp--;
                if (sp >= kPacketStack) continue; // an entry that found no room (the packet's results are not used then)
                __builtin_amdgcn_wave_barrier();
                cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)stack[sp]);
                break;
            }
        }
        bool unproven = false;
        if (alive) {
            unproven = !query_result_proven(S, T.R, T.t, T.tri, T.tieT, lost ? 1u : 0u);
            for (uint32_t s_ = 0; s_ < S.nSpheres; s_++) // src/bvh.cpp:366-367
                if (sphere_test(S.spheres[s_], T.R.o, T.R.d, T.t)) T.tri = kSphereBase + s_;
            if (!unproven) P.hits[idx] = HitRec { T.t, T.tri };
        }
        const bool again = unfit || unproven;
        if (__ballot(again) != 0ull) {
            const uint32_t pos = wave_fetch(P.fallCount, again, lane);
            if (again) P.fallList[pos] = idx;
        }
    }
    if (COUNT) flush_counters(P.stats, lane, 0, 0, 0, tc, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Ray order for the big one-ray-per-lane launches.  A wave walks the tree fastest when its 64 rays walk the same nodes;
// the deferred rays of a frame (last-level glossy rays, shadow rays) are produced in the order their pixels happened to
// finish.  Three small kernels give a stage's rays a traversal order without moving them: a key per ray (origin cell and
// direction cell, the weight between the two chosen by the stage), a histogram of the keys, and a scatter of the ray
// indices into key order (counting sort; the order inside a bucket is whatever the atomics make it, which no result
// depends on: hits are stored by ray index).  The trace launches then take their rays through that index list.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kOrderKeyBits = 17;
constexpr uint32_t kOrderBuckets = 1u << kOrderKeyBits;
struct OrderParams {
    const char* rays;        // RayRec at the start of each record
    uint32_t rayStride;
    const uint32_t* countPtr; // number of items = clamp(*countPtr - countSub, 0, countMaxItems) * countMul (as TraceParams)
    uint32_t countSub, countMaxItems, countMul;
    uint32_t minRays;        // fewer rays than this: nothing is ordered (the launches then take the rays as they lie)
    float lo[3], inv[3];     // scene box: cell = (o - lo) * inv
    uint32_t dirMajor;       // 1: rays from few origins (shadow rays: direction first); 0: origin first
    uint32_t* keys;          // per ray
    uint32_t* hist;          // kOrderBuckets counters, zero on entry; holds each bucket's first place after order_scan_kernel
    uint32_t* order;         // ray indices in key order
};
__device__ __forceinline__ uint32_t order_total(const OrderParams& P)
{
    const uint32_t have = *P.countPtr;
    const uint32_t items = have > P.countSub ? min(have - P.countSub, P.countMaxItems) : 0u;
    return items * P.countMul;
}
__device__ __forceinline__ uint32_t spread3(uint32_t v) // bits of v (up to 5) to every third place
{
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
__device__ __forceinline__ uint32_t spread2(uint32_t v) // bits of v (up to 8) to every second place
{
    v = (v | (v << 4)) & 0x0f0fu;
    v = (v | (v << 2)) & 0x3333u;
    v = (v | (v << 1)) & 0x5555u;
    return v;
}
__device__ __forceinline__ uint32_t order_key(const OrderParams& P, const float4 a, const float4 b)
{
    if (a.w < 0.0f) return 0u; // no ray in this slot
    // octahedral map of the direction to [0, 1)^2
    const float n = fabsf(b.x) + fabsf(b.y) + fabsf(b.z);
    float u = n > 0.0f ? b.x / n : 0.0f, v = n > 0.0f ? b.y / n : 0.0f;
    if (b.z < 0.0f) {
        const float uu = (1.0f - fabsf(v)) * (u >= 0.0f ? 1.0f : -1.0f), vv = (1.0f - fabsf(u)) * (v >= 0.0f ? 1.0f : -1.0f);
        u = uu, v = vv;
    }
    u = fminf(fmaxf(u * 0.5f + 0.5f, 0.0f), 0.999999f), v = fminf(fmaxf(v * 0.5f + 0.5f, 0.0f), 0.999999f);
    const float cx = fminf(fmaxf((a.x - P.lo[0]) * P.inv[0], 0.0f), 0.999999f), cy = fminf(fmaxf((a.y - P.lo[1]) * P.inv[1], 0.0f), 0.999999f),
                cz = fminf(fmaxf((a.z - P.lo[2]) * P.inv[2], 0.0f), 0.999999f);
    if (P.dirMajor) { // 7 + 7 bits of direction (interleaved), 1 bit of origin per axis
        const uint32_t d = spread2((uint32_t)(u * 128.0f)) | (spread2((uint32_t)(v * 128.0f)) << 1);
        const uint32_t o = (uint32_t)(cx * 2.0f) | ((uint32_t)(cy * 2.0f) << 1) | ((uint32_t)(cz * 2.0f) << 2);
        return (d << 3) | o;
    }
    // 4 bits of origin per axis (interleaved), 3 + 2 bits of direction
    const uint32_t o = spread3((uint32_t)(cx * 16.0f)) | (spread3((uint32_t)(cy * 16.0f)) << 1) | (spread3((uint32_t)(cz * 16.0f)) << 2);
    return (o << 5) | ((uint32_t)(u * 8.0f) << 2) | (uint32_t)(v * 4.0f);
}
__global__ __launch_bounds__(kBlock) void order_key_kernel(const OrderParams P)
{
    const uint32_t total = order_total(P);
    if (total < P.minRays) return;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const float4* q = reinterpret_cast<const float4*>(P.rays + (size_t)i * P.rayStride);
        const uint32_t k = order_key(P, q[0], q[1]);
        P.keys[i] = k;
        atomicAdd(P.hist + k, 1u);
    }
}
// one workgroup: counts -> first places (exclusive prefix sums), in place
__global__ __launch_bounds__(1024) void order_scan_kernel(const OrderParams P)
{
    __shared__ uint32_t s_part[1024];
    if (order_total(P) < P.minRays) return;
    constexpr uint32_t kPer = kOrderBuckets / 1024;
    const uint32_t t = threadIdx.x;
    uint32_t local[kPer];
    uint32_t sum = 0;
    for (uint32_t k = 0; k < kPer; k++) local[k] = P.hist[t * kPer + k], sum += local[k];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) { // inclusive scan of the per-thread sums
        const uint32_t v = t >= off ? s_part[t - off] : 0u;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    uint32_t run = s_part[t] - sum;
    for (uint32_t k = 0; k < kPer; k++) {
        P.hist[t * kPer + k] = run;
        run += local[k];
    }
}
__global__ __launch_bounds__(kBlock) void order_scatter_kernel(const OrderParams P)
{
    const uint32_t total = order_total(P);
    if (total < P.minRays) return;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) P.order[atomicAdd(P.hist + P.keys[i], 1u)] = i;
}

// ---------------------------------------------------------------------------------------------
// Path stage: the sequential part of each pixel, one radiance ray per active pixel per round.
//
// A pixel's depth-first state lives in HBM between rounds (PathState: a 64-byte header with the ray
// to trace next + one 80-byte frame per pending glossy loop).  camera_kernel creates a state only for
// pixels that have a ray to trace; path_logic_kernel r takes the compact list of active states and the
// hits of their rays, advances each recursion to its next ray (or candidate batch) and appends it to the
// list of round r+1.  Every round therefore runs dense waves, however uneven the pixels' chains are.
// ---------------------------------------------------------------------------------------------
constexpr int kFrameDwords = 20;                                   // 80 B per pending glossy loop
constexpr int kPassDwords = 12;                                    // 48 B per pending pass-through ray: origin, direction, weight, flag
constexpr int kPassOff = RT_MAX_RAY_DEPTH * kFrameDwords;          // the pass records follow the loop frames
constexpr int kStateDwords = 16 + RT_MAX_RAY_DEPTH * (kFrameDwords + kPassDwords); // 832 B
enum PathStep : int { PS_DONE = 0, PS_TRACE = 1, PS_HIT, PS_GLOSSY, PS_POP, PS_BATCH };

struct PathLane {
    // header
    uint32_t pixel, smp, seed, pxy; // pxy = px | py << 16
    vec3 to, td, curW;
    int depth, loopDepth; // rayDepth of the ray in flight; rayDepth of the innermost pending glossy loop (-1: none)
    uint32_t sampleIdx;
    uint32_t nRays;   // rays in flight for this state: 1, or batchRays for a batched glossy loop
    // working
    uint32_t rayBase; // first of those rays / hits
    float tt;
    uint32_t ttri;
    int state;
};

__device__ __forceinline__ void load_header(const float* s, PathLane& L)
{
    const float4* q = reinterpret_cast<const float4*>(s);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    L.pixel = __float_as_uint(a.x), L.smp = __float_as_uint(a.y), L.seed = __float_as_uint(a.z), L.pxy = __float_as_uint(a.w);
    L.to = make3(b.x, b.y, b.z), L.td = make3(b.w, c.x, c.y), L.curW = make3(c.z, c.w, d.x);
    const uint32_t pk = __float_as_uint(d.y);
    L.depth = (int)(pk & 0xffu), L.loopDepth = (int)((pk >> 8) & 0xffu) - 1, L.sampleIdx = pk >> 16;
    L.nRays = __float_as_uint(d.z);
}
__device__ __forceinline__ void store_header(float* s, const PathLane& L)
{
    float4* q = reinterpret_cast<float4*>(s);
    const uint32_t pk = (uint32_t)L.depth | ((uint32_t)(L.loopDepth + 1) << 8) | (L.sampleIdx << 16);
    q[0] = make_float4(__uint_as_float(L.pixel), __uint_as_float(L.smp), __uint_as_float(L.seed), __uint_as_float(L.pxy));
    q[1] = make_float4(L.to.x, L.to.y, L.to.z, L.td.x);
    q[2] = make_float4(L.td.y, L.td.z, L.curW.x, L.curW.y);
    q[3] = make_float4(L.curW.z, __uint_as_float(pk), __uint_as_float(L.nRays), 0.0f);
}

__device__ __forceinline__ uint32_t pixel_ray_count(const rt_features& F)
{
    if (F.numPixelSamples <= 1) return 1;
    if (!F.enableJitteredSampling) return F.numPixelSamples;
    const uint32_t ns = (uint32_t)roundf(sqrtf((float)F.numPixelSamples));
    return ns * ns;
}

// generatePixelRays (src/render.cpp:56-123) for pixel ray `k` of the pixel in L
__device__ __forceinline__ void make_pixel_ray(const WaveParams& P, PathLane& L, const uint32_t k, const uint32_t nPixelRays)
{
    const rt_features& F = P.F;
    const uint32_t px = L.pxy & 0xffffu;
    uint32_t py = L.pxy >> 16, frame = 0;
    if (P.frames > 1) frame = py / (uint32_t)P.H, py -= frame * (uint32_t)P.H;
    float ndcX, ndcY;
    if (F.numPixelSamples > 1) { // offsets are drawn for every pixel ray before any is traced
        uint32_t s2 = L.seed;
        for (uint32_t j = 0; j < 2 * k; j++) (void)sampler_next(s2);
        float ox = sampler_next(s2);
        float oy = sampler_next(s2);
        if (F.enableJitteredSampling) {
            const uint32_t ns = (uint32_t)roundf(sqrtf((float)F.numPixelSamples));
            ox /= ns;
            oy /= ns; // the cell offset {i / ns, j / ns} is an integer division: always 0
        }
        ndcX = ((float)px + ox) / (float)P.W * 2.f - 1.f;
        ndcY = ((float)py + oy) / (float)P.H * 2.f - 1.f;
    } else { // pixel centre (src/render.cpp:68)
        ndcX = ((float)px + 0.5f) / (float)P.W * 2.f - 1.f;
        ndcY = ((float)py + 0.5f) / (float)P.H * 2.f - 1.f;
    }
    camera_ray(P.cams[frame], ndcX, ndcY, L.to, L.td);
    L.curW = splat(1.0f / static_cast<float>(nPixelRays)); // renderRays: L / rays.size()
    L.depth = P.startDepth;
    L.loopDepth = -1;
    L.sampleIdx = k;
    L.nRays = 1;
}

// Reserves `n` consecutive slots of a queue for every lane with n > 0: one atomic per wave.  Returns the
// lane's first slot (0xffffffff when the reservation does not fit the queue: nothing may be written then).
__device__ __forceinline__ uint32_t reserve_slots(uint32_t* count, const uint32_t cap, const uint32_t n, const int lane)
{
    uint32_t incl = n; // inclusive prefix sum over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    const uint32_t total = __shfl(incl, 63);
    if (total == 0) return 0xffffffffu;
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(count, total);
    base = __shfl(base, 63);
    const uint32_t first = base + incl - n;
    return (n != 0 && first + n <= cap) ? first : 0xffffffffu;
}

// The same for three queues at once and for a whole workgroup: every thread of the workgroup must call it.  Counters
// that every wave of a launch updates answer one update at a time (about 9 ns each: a thousand waves asking together
// wait 9 us), and a wave waits for the answer with all its lanes - so the workgroup's requests are summed in LDS, one
// thread issues the three updates together, and they are waited for once.  s_n: 12 words, s_base: 3 words.
__device__ __forceinline__ void reserve_slots3(uint32_t* c0, const uint32_t cap0, const uint32_t n0, uint32_t* c1, const uint32_t cap1,
                                               const uint32_t n1, uint32_t* c2, const uint32_t cap2, const uint32_t n2, const int lane,
                                               uint32_t* s_n, uint32_t* s_base, uint32_t& s0, uint32_t& s1, uint32_t& s2)
{
    uint32_t i0 = n0, i1 = n1, i2 = n2; // inclusive prefix sums over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o0 = __shfl_up(i0, off), o1 = __shfl_up(i1, off), o2 = __shfl_up(i2, off);
        if (lane >= off) i0 += o0, i1 += o1, i2 += o2;
    }
    const int wv = threadIdx.x >> 6;
    if (lane == 63) s_n[wv] = i0, s_n[4 + wv] = i1, s_n[8 + wv] = i2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t0 = s_n[0] + s_n[1] + s_n[2] + s_n[3], t1 = s_n[4] + s_n[5] + s_n[6] + s_n[7], t2 = s_n[8] + s_n[9] + s_n[10] + s_n[11];
        uint32_t b0 = 0, b1 = 0, b2 = 0;
        if (t0) b0 = atomicAdd(c0, t0);
        if (t1) b1 = atomicAdd(c1, t1);
        if (t2) b2 = atomicAdd(c2, t2);
        s_base[0] = b0, s_base[1] = b1, s_base[2] = b2;
    }
    __syncthreads();
    uint32_t f0 = s_base[0] + i0 - n0, f1 = s_base[1] + i1 - n1, f2 = s_base[2] + i2 - n2;
    for (int k = 0; k < wv; k++) f0 += s_n[k], f1 += s_n[4 + k], f2 += s_n[8 + k];
    __syncthreads();
    s0 = (n0 != 0 && f0 + n0 <= cap0) ? f0 : 0xffffffffu;
    s1 = (n1 != 0 && f1 + n1 <= cap1) ? f1 : 0xffffffffu;
    s2 = (n2 != 0 && f2 + n2 <= cap2) ? f2 : 0xffffffffu;
}

// One sample of renderRayGlossyComponent's loop (src/extra.cpp:489-505): sampleDisk (src/extra.cpp:443-450)
// draws twice from `smp`; false when the perturbed direction points into the surface.
__device__ __forceinline__ bool glossy_sample(uint32_t& smp, const float radius, const vec3 w, const vec3 u, const vec3 v, const vec3 nrm, vec3& pd)
{
    const float r = radius * sqrtf(sampler_next(smp));
    const float angle = sampler_next(smp) * 2 * 3.14159265358979323846264338327950288f;
    float sn, cs;
    libm_sincosf(angle, sn, cs);
    const float dx = r * cs, dy = r * sn;
    pd = normalize(w + dx * u + dy * v);
    return !(dot(pd, nrm) <= 0);
}

__device__ __forceinline__ void store_shade_task(ShadeTask* q, const uint32_t pixel, const uint32_t matInfo, const uint32_t smp,
                                                 const uint32_t tri, const vec3 W, const vec3 p, const vec3 d, const vec3 n)
{
    float4* o = reinterpret_cast<float4*>(q);
    o[0] = make_float4(__uint_as_float(pixel), __uint_as_float(matInfo), __uint_as_float(smp), __uint_as_float(tri));
    o[1] = make_float4(W.x, W.y, W.z, p.x);
    o[2] = make_float4(p.y, p.z, d.x, d.y);
    o[3] = make_float4(d.z, n.x, n.y, n.z);
}
// A hit of a batched glossy loop's child as path_logic_kernel leaves it: the ray and where it ended.  shade_expand_kernel
// turns it into the ShadeTask proper and, with leafSlot != 0xffffffff, draws the child's own glossy loop into the N LeafTask
// slots from there.
constexpr uint32_t kRawShade = 0x7fffffffu; // in the matInfo word: no mesh or sphere index reaches it
__device__ __forceinline__ void store_raw_shade_task(ShadeTask* q, const uint32_t pixel, const uint32_t smp, const uint32_t tri, const vec3 W,
                                                     const vec3 o, const vec3 d, const float t, const uint32_t leafSlot)
{
    float4* out = reinterpret_cast<float4*>(q);
    out[0] = make_float4(__uint_as_float(pixel), __uint_as_float(kRawShade), __uint_as_float(smp), __uint_as_float(tri));
    out[1] = make_float4(W.x, W.y, W.z, o.x);
    out[2] = make_float4(o.y, o.z, d.x, d.y);
    out[3] = make_float4(d.z, t, __uint_as_float(leafSlot), 0.0f);
}
__device__ __forceinline__ void store_leaf_task(LeafTask* q, const uint32_t pixel, const vec3 o, const vec3 d, const float tmax, const vec3 W)
{
    float4* out = reinterpret_cast<float4*>(q);
    out[0] = make_float4(o.x, o.y, o.z, tmax);
    out[1] = make_float4(d.x, d.y, d.z, __uint_as_float(pixel));
    out[2] = make_float4(W.x, W.y, W.z, 0.0f);
}
constexpr uint32_t kHolePixel = 0xffffffffu; // a reserved but unused ShadeTask slot
// Advances every lane from "its ray has been traced" (PS_HIT) to its next ray (PS_TRACE) or the end
// of its pixel (PS_DONE), appending the shade / leaf tasks met on the way.  Every thread of the workgroup must call it
// (the passes are workgroup-convergent: queue slots are reserved per workgroup).  s_n: 12 words, s_base: 3 words of LDS.
// frames: this lane's frame storage in HBM (null while the lane has none: lanes allocate lazily).
// TRANSP: the frame has transparency enabled (pass-through rays exist); compiled apart so that frames without it do not carry its registers
template <bool TRANSP, typename AllocFn>
__device__ __forceinline__ void advance_lanes(const WaveParams& P, PathLane& L, float*& frames, float* const passes, AllocFn&& allocFrames,
                                              const int lane, uint32_t& cPrimary, uint32_t& cSecondary, uint32_t& cDrop, uint32_t* s_n,
                                              uint32_t* s_base)
{
    // frames / passes: this lane's record of rayDepth 0; the record of rayDepth l lies l * levelStride further
    const size_t frameLevel = (size_t)P.stateCap * kFrameDwords, passLevel = (size_t)P.stateCap * kPassDwords;
    const SceneView& S = P.S;
    const rt_features& F = P.F;
    const bool deferLeaves = (P.lightDraws == 0);
    const uint32_t nPixelRays = pixel_ray_count(F);
    ShadeTask st;
    LeafTask lt;
    do {
        bool wantShade = false, wantLeaf = false, wantFrames = false, wantBatch = false, wantConsume = false;
        bool openLoop = false; // push a glossy frame after this pass's shade task has been appended
        bool openPass = false; // the frame (a loop's, or one of its own) also holds a pending pass-through ray
        vec3 passW = splat(0.0f);
        uint32_t hitMask = 0, opensMask = 0, nHits = 0, nOpened = 0; // of a batch whose candidates are back (PS_BATCH)
        uint32_t batchKids = 0; // children of the loop in the batch being emitted / consumed
        constexpr uint32_t kMaxN = 5; // N (N + 1) / 2 <= kMaxBatchRays
        while (L.state > PS_TRACE && !wantShade && !wantLeaf && !wantFrames && !wantBatch && !wantConsume) {
            switch (L.state) {
            case PS_HIT: {
                // renderRay after the intersection (src/recursive.cpp:29-36, 62-80)
                if (L.ttri == kNoHit) {
                    L.state = PS_POP;
                    break;
                }
                Hit h;
                fill_hit(S, F, L.ttri, L.to, L.td, L.tt, h);
                if (L.depth == P.startDepth && L.sampleIdx == 0 && L.loopDepth < 0) {
                    const uint32_t px = L.pxy & 0xffffu, py = L.pxy >> 16;
                    if (P.primId) P.primId[py * P.W + px] = h.primId;
                    if (P.primT) P.primT[py * P.W + px] = L.tt;
                }
                const vec3 p = L.to + L.tt * L.td;
                st.pixel = L.pixel;
                st.matInfo = h.mesh | (h.isSphere ? 0x80000000u : 0u);
                st.smp = L.smp;
                st.tri = L.ttri;
                st.W[0] = L.curW.x, st.W[1] = L.curW.y, st.W[2] = L.curW.z;
                st.px = p.x, st.py = p.y, st.pz = p.z;
                st.dx = L.td.x, st.dy = L.td.y, st.dz = L.td.z;
                st.nx = h.normal.x, st.ny = h.normal.y, st.nz = h.normal.z;
                wantShade = true; // the hit's direct light: deferred to the shade kernel
                for (uint32_t k = 0; k < P.lightDraws; k++) (void)sampler_next(L.smp); // the light loop's own draws
                L.state = PS_POP;
                if (L.depth < RT_MAX_RAY_DEPTH && (F.enableReflections || (TRANSP && F.enableTransparency))) {
                    const Mat m = load_material(S, h.mesh, h.isSphere);
                    // renderRayTransparentComponent (src/recursive.cpp:155-163): hitColor * (1 - transparency) + passColor *
                    // transparency, once the reflection has returned.  As weights: everything this hit adds itself (direct
                    // light, reflection subtree) counts (1 - transparency) times, the pass-through subtree transparency times.
                    const bool passes = (TRANSP && F.enableTransparency) && m.transparency != 1.f;
                    vec3 Wd = L.curW;
                    if (passes) {
                        Wd = L.curW * (1.f - m.transparency);
                        passW = L.curW * m.transparency;
                        st.W[0] = Wd.x, st.W[1] = Wd.y, st.W[2] = Wd.z;
                    }
                    const bool reflective = F.enableReflections && ((m.ks.x != 0.0f) || (m.ks.y != 0.0f) || (m.ks.z != 0.0f));
                    bool spawned = false;
                    if (reflective) {
                        // generateReflectionRay (src/recursive.cpp:91-103)
                        vec3 rorg = p;
                        const vec3 rdir = L.td - 2.f * dot(L.td, h.normal) * h.normal;
                        rorg = rorg + 0.00001f * rdir;
                        if (!F.enableGlossyReflection) { // renderRaySpecularComponent: weight * ks
                            L.to = rorg, L.td = rdir;
                            L.curW = Wd * m.ks;
                            L.depth++;
                            cSecondary++;
                            L.state = PS_TRACE;
                            openPass = passes; // the pass-through ray waits in a frame until this chain has returned
                            spawned = true;
                        } else if (F.numGlossySamples > 0) { // renderRayGlossyComponent (src/extra.cpp:464-508)
                            // the frame is parked in the task record's neighbours until storage exists
                            L.to = rorg, L.td = rdir; // reused as scratch: reflection ray of the new loop
                            L.tt = 0.5f / m.shininess;
                            L.curW = Wd;
                            openLoop = true;
                            openPass = passes;
                            spawned = true;
                            // ks and normal are re-read from st / material when the frame is written
                        }
                    }
                    if (!spawned && passes) { // generatePassthroughRay (src/recursive.cpp:112-123): nothing else is pending at this hit
                        L.to = p + 0.00001f * L.td;
                        L.curW = passW;
                        L.depth++;
                        cSecondary++;
                        L.state = PS_TRACE;
                    }
                }
                break;
            }
            case PS_GLOSSY: {
                float4* fq = reinterpret_cast<float4*>(frames + L.loopDepth * frameLevel);
                const float4 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3], f4 = fq[4];
                const vec3 fW = make3(f0.x, f0.y, f0.z), rorg = make3(f0.w, f1.x, f1.y), rdir = make3(f1.z, f1.w, f2.x);
                const vec3 nrm = make3(f2.y, f2.z, f2.w), ks = make3(f3.x, f3.y, f3.z);
                const float radius = f3.w;
                uint32_t i = __float_as_uint(f4.x);
                const int enclosing = (int)__float_as_uint(f4.y) - 1;
                vec3 w, u, v;
                ortho_basis(rdir, w, u, v);
                if (P.batchRays != 0 && L.loopDepth == RT_MAX_RAY_DEPTH - 2 && i + 2 <= F.numGlossySamples) {
                    wantBatch = true; // emitted below, once the wave has reserved the rays
                    batchKids = min(P.batchWindow, F.numGlossySamples - i);
                    break;
                }
                bool launched = false;
                while (i < F.numGlossySamples) {
                    // sampleDisk (src/extra.cpp:443-450): two draws per sample, consumed even if rejected
                    const float r = radius * sqrtf(sampler_next(L.smp));
                    const float angle = sampler_next(L.smp) * 2 * 3.14159265358979323846264338327950288f;
                    float sn, cs;
                    libm_sincosf(angle, sn, cs);
                    const float dx = r * cs, dy = r * sn;
                    const vec3 pd = normalize(w + dx * u + dy * v);
                    i++;
                    if (dot(pd, nrm) <= 0) continue;
                    // accumulatedColor += ks * L; hitColor += accumulatedColor / N  ->  weight * ks / N
                    const vec3 cw = fW * ks / float(F.numGlossySamples);
                    const vec3 corg = rorg + 0.0001f * pd;
                    cSecondary++;
                    launched = true;
                    if (deferLeaves && L.loopDepth + 1 >= RT_MAX_RAY_DEPTH) {
                        // a ray at rayDepth 6 spawns nothing and draws nothing: trace it later
                        lt.ray.ox = corg.x, lt.ray.oy = corg.y, lt.ray.oz = corg.z, lt.ray.tmax = FLT_MAX;
                        lt.ray.dx = pd.x, lt.ray.dy = pd.y, lt.ray.dz = pd.z, lt.ray.tag = L.pixel;
                        lt.W[0] = cw.x, lt.W[1] = cw.y, lt.W[2] = cw.z;
                        lt.pad = 0.0f;
                        wantLeaf = true; // leave to append; come back to PS_GLOSSY
                        break;
                    }
                    L.to = corg, L.td = pd;
                    L.curW = cw;
                    L.depth = L.loopDepth + 1;
                    L.state = PS_TRACE;
                    break;
                }
                frames[L.loopDepth * frameLevel + 16] = __uint_as_float(i);
                if (!launched) { // loop finished: the hit's pass-through ray if one waits, else resume the enclosing loop
                    const float* pr = passes + L.loopDepth * passLevel;
                    L.state = PS_POP;
                    if ((TRANSP && F.enableTransparency) && pr[9] != 0.0f) {
                        const float4* pq = reinterpret_cast<const float4*>(pr);
                        const float4 p0 = pq[0], p1 = pq[1], p2 = pq[2];
                        L.to = make3(p0.x, p0.y, p0.z), L.td = make3(p0.w, p1.x, p1.y), L.curW = make3(p1.z, p1.w, p2.x);
                        L.depth = L.loopDepth + 1;
                        cSecondary++;
                        L.state = PS_TRACE;
                    }
                    L.loopDepth = enclosing;
                }
                break;
            }
            case PS_POP: {
                // the ray in flight is finished: continue the innermost pending glossy loop, else the pixel
                if (L.loopDepth >= 0) {
                    L.state = PS_GLOSSY;
                    break;
                }
                if (L.sampleIdx + 1 < nPixelRays) {
                    make_pixel_ray(P, L, L.sampleIdx + 1, nPixelRays);
                    cPrimary++;
                    L.state = PS_TRACE;
                } else {
                    L.state = PS_DONE;
                }
                break;
            }
            case PS_BATCH: {
                // The traced candidates are back.  Only the sampler's progress is needed before the pixel's next ray can
                // start, and it depends on which candidates hit a reflecting surface: that bit per candidate, all loads
                // in flight together.
                batchKids = L.nRays == 3 ? 2u : (L.nRays == 6 ? 3u : (L.nRays == 10 ? 4u : 5u)); // nRays = kids (kids + 1) / 2
                uint32_t ctri[kMaxBatchRays], cword[kMaxBatchRays];
#pragma unroll
                for (uint32_t c = 0; c < kMaxBatchRays; c++) ctri[c] = (c < L.nRays) ? P.hits[L.rayBase + c].tri : kNoHit;
                // (triangles: the scene's bit per triangle, a table small enough to stay in the caches, instead of triangle -> mesh -> material)
#pragma unroll
                for (uint32_t c = 0; c < kMaxBatchRays; c++) cword[c] = (ctri[c] < kSphereBase) ? S.triReflects[ctri[c] >> 5] : 0u;
#pragma unroll
                for (uint32_t c = 0; c < kMaxBatchRays; c++) {
                    if (ctri[c] == kNoHit) continue;
                    hitMask |= 1u << c;
                    if (ctri[c] < kSphereBase) {
                        opensMask |= ((cword[c] >> (ctri[c] & 31u)) & 1u) << c;
                    } else {
                        const float* mks = S.spheres[ctri[c] - kSphereBase].ks;
                        if ((mks[0] != 0.0f) || (mks[1] != 0.0f) || (mks[2] != 0.0f)) opensMask |= 1u << c;
                    }
                }
                if (!F.enableReflections) opensMask = 0;
                // the candidate that stands for sample i lies one place further for every earlier sample with a loop of its own
#pragma unroll
                for (uint32_t i = 0; i < kMaxN; i++) {
                    const uint32_t k = i * (i + 1) / 2 + nOpened;
                    if (i < batchKids) nHits += (hitMask >> k) & 1u, nOpened += (opensMask >> k) & 1u;
                }
                wantConsume = true;
                break;
            }
            default: L.state = PS_DONE; break;
            }
        }
        // ---- queue slots for everything this pass produced (a lane asks for one kind): one trip to the counters ----
        // The two task queues are reserved without regard to their ends, and every slot is checked against the end when it is
        // written: the later stages walk a queue up to its end, so every slot below it that has been reserved must be written
        // (a queue's counter beyond its end reports the frame incomplete).
        uint32_t shadeSlot = 0xffffffffu, leafSlot = 0xffffffffu, raySlot = 0xffffffffu;
        {
            const uint32_t N = F.numGlossySamples;
            reserve_slots3(P.qCount + 0, 0xffffffffu, wantShade ? 1u : (wantConsume ? nHits : 0u), P.qCount + 1, 0xffffffffu,
                           wantLeaf ? 1u : (wantConsume ? N * nOpened : 0u), P.rayCountOut, P.rayCap, wantBatch ? batchKids * (batchKids + 1) / 2 : 0u, lane, s_n,
                           s_base, shadeSlot, leafSlot, raySlot);
        }
        // ---- append the tasks produced in this pass ----
        if (__ballot(wantShade) != 0ull) {
            const bool pushFrame = wantShade && (openLoop || openPass);
            if (wantShade) {
                if (shadeSlot < P.shadeCap) {
                    const float4* src = reinterpret_cast<const float4*>(&st);
                    float4* dst = reinterpret_cast<float4*>(P.shadeQ + shadeSlot);
                    dst[0] = src[0], dst[1] = src[1], dst[2] = src[2], dst[3] = src[3];
                } else {
                    cDrop++;
                }
            }
            // lanes that open a glossy loop need frame storage; allocate for those that have none yet
            if (__ballot(pushFrame && frames == nullptr) != 0ull) allocFrames(pushFrame && frames == nullptr);
            if (pushFrame) {
                const int level = openLoop ? L.depth : L.depth - 1; // rayDepth of the hit (a specular chain has already moved on)
                float4* fq = reinterpret_cast<float4*>(frames + level * frameLevel);
                if (openLoop) {
                    const Mat m = load_material(S, st.matInfo & 0x7fffffffu, (st.matInfo >> 31) != 0);
                    fq[0] = make_float4(L.curW.x, L.curW.y, L.curW.z, L.to.x);
                    fq[1] = make_float4(L.to.y, L.to.z, L.td.x, L.td.y);
                    fq[2] = make_float4(L.td.z, st.nx, st.ny, st.nz);
                    fq[3] = make_float4(m.ks.x, m.ks.y, m.ks.z, L.tt);
                    fq[4] = make_float4(__uint_as_float(0u), __uint_as_float((uint32_t)(L.loopDepth + 1)), 0.0f, 0.0f);
                    L.state = PS_GLOSSY;
                } else { // no loop at this hit: a frame whose samples are all taken
                    fq[4] = make_float4(__uint_as_float(0xffffffffu), __uint_as_float((uint32_t)(L.loopDepth + 1)), 0.0f, 0.0f);
                }
                if ((TRANSP && F.enableTransparency)) {
                    float4* pq = reinterpret_cast<float4*>(passes + level * passLevel);
                    // generatePassthroughRay (src/recursive.cpp:112-123): from just past the hit point, same direction
                    const vec3 d = make3(st.dx, st.dy, st.dz);
                    const vec3 porg = make3(st.px, st.py, st.pz) + 0.00001f * d;
                    pq[0] = make_float4(porg.x, porg.y, porg.z, d.x);
                    pq[1] = make_float4(d.y, d.z, passW.x, passW.y);
                    pq[2] = make_float4(passW.z, openPass ? 1.0f : 0.0f, 0.0f, 0.0f);
                }
                L.loopDepth = level;
            }
        }
        if (wantLeaf) {
            if (leafSlot < P.leafCap) {
                const float4* src = reinterpret_cast<const float4*>(&lt);
                float4* dst = reinterpret_cast<float4*>(P.leafQ + leafSlot);
                dst[0] = src[0], dst[1] = src[1], dst[2] = src[2];
            } else {
                cDrop++;
            }
        }
        // ---- batched glossy loop: the loop's children are the last rays that can spawn anything, and each of
        // them either consumes no further draws or exactly 2N (its own loop, whose children are deferred).  The
        // a-th child of a batch therefore starts at one of a+1 known sampler offsets: trace a candidate for each, pick
        // afterwards.  A batch holds the loop's next batchWindow children (the whole loop when the window allows).
        if (__ballot(wantBatch) != 0ull) {
            const uint32_t N = F.numGlossySamples;
            const uint32_t base = raySlot;
            if (wantBatch) {
                if (base == 0xffffffffu) {
                    cDrop++;
                    L.state = PS_DONE;
                } else {
                    const float4* fq = reinterpret_cast<const float4*>(frames + L.loopDepth * frameLevel);
                    const float4 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3];
                    const vec3 rorg = make3(f0.w, f1.x, f1.y), rdir = make3(f1.z, f1.w, f2.x), nrm = make3(f2.y, f2.z, f2.w);
                    const float radius = f3.w;
                    vec3 w, u, v;
                    ortho_basis(rdir, w, u, v);
                    uint32_t rowSmp = L.smp; // offset 2i: no earlier child of the batch consumed anything
                    uint32_t k = 0;
                    for (uint32_t i = 0; i < batchKids; i++) {
                        uint32_t candSmp = rowSmp;
                        for (uint32_t c = 0; c <= i; c++, k++) {
                            uint32_t s2 = candSmp;
                            vec3 pd;
                            if (glossy_sample(s2, radius, w, u, v, nrm, pd)) store_ray(P.pathRays + base + k, rorg + 0.0001f * pd, pd, FLT_MAX, 0u);
                            else store_ray(P.pathRays + base + k, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, 0u); // no ray: misses at once
                            for (uint32_t z = 0; z < 2 * N; z++) (void)sampler_next(candSmp); // one more earlier child with a loop of its own
                        }
                        (void)sampler_next(rowSmp);
                        (void)sampler_next(rowSmp);
                    }
                    L.nRays = batchKids * (batchKids + 1) / 2;
                    L.rayBase = base;
                    L.state = PS_TRACE;
                }
            }
        }
        if (__ballot(wantConsume) != 0ull) {
            // the traced candidates are back: walk the loop in the reference's order
            const uint32_t N = F.numGlossySamples;
            const uint32_t shadeBase = shadeSlot, leafBase = leafSlot;
            if (wantConsume) {
                const float4* fq = reinterpret_cast<const float4*>(frames + L.loopDepth * frameLevel);
                const float4 f0 = fq[0], f3 = fq[3], f4 = fq[4];
                const vec3 fW = make3(f0.x, f0.y, f0.z), ks = make3(f3.x, f3.y, f3.z);
                const int enclosing = (int)__float_as_uint(f4.y) - 1;
                const vec3 cw = fW * ks / float(N);
                // all of the batch's tasks or none: what was reserved below a queue's end and stays unused becomes holes
                const bool room = (nHits == 0 || shadeBase + nHits <= P.shadeCap) && (nOpened == 0 || leafBase + N * nOpened <= P.leafCap);
                if (!room) {
                    cDrop++;
                    for (uint32_t k = 0; k < nHits; k++)
                        if (shadeBase + k < P.shadeCap) reinterpret_cast<uint32_t*>(P.shadeQ + shadeBase + k)[0] = kHolePixel;
                    for (uint32_t k = 0; k < N * nOpened; k++)
                        if (leafBase + k < P.leafCap)
                            store_leaf_task(P.leafQ + leafBase + k, kHolePixel, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, splat(0.0f));
                }
                uint32_t ksel[kMaxN];
                {
                    uint32_t opened = 0;
#pragma unroll
                    for (uint32_t i = 0; i < kMaxN; i++) {
                        ksel[i] = i * (i + 1) / 2 + opened;
                        if (i < batchKids) opened += (opensMask >> ksel[i]) & 1u;
                    }
                }
                // the samples' rays and hits; what the hits look like (normal, material) is worked out by shade_expand_kernel
                float4 ra[kMaxN], rb[kMaxN];
                HitRec hr[kMaxN];
#pragma unroll
                for (uint32_t i = 0; i < kMaxN; i++) {
                    if (i >= batchKids) continue;
                    const float4* rq = reinterpret_cast<const float4*>(P.pathRaysIn + L.rayBase + ksel[i]);
                    ra[i] = rq[0], rb[i] = rq[1];
                    hr[i] = P.hits[L.rayBase + ksel[i]];
                }
                uint32_t nShade = 0, nLeaf = 0;
#pragma unroll
                for (uint32_t i = 0; i < kMaxN; i++) {
                    if (i >= batchKids) continue;
                    (void)sampler_next(L.smp); // the sample's own two draws (its direction is in the stored ray)
                    (void)sampler_next(L.smp);
                    if (ra[i].w < 0.0f) continue; // rejected sample
                    cSecondary++;
                    if (hr[i].tri == kNoHit) continue;
                    const bool opens = ((opensMask >> ksel[i]) & 1u) != 0;
                    if (room)
                        store_raw_shade_task(P.shadeQ + shadeBase + nShade, L.pixel, L.smp, hr[i].tri, cw, make3(ra[i].x, ra[i].y, ra[i].z),
                                             make3(rb[i].x, rb[i].y, rb[i].z), hr[i].t, opens ? leafBase + nLeaf : 0xffffffffu);
                    nShade++;
                    if (!opens) continue;
                    // the child's own glossy loop: its children sit at the recursion limit and draw nothing further
                    nLeaf += N;
                    for (uint32_t z = 0; z < 2 * N; z++) (void)sampler_next(L.smp);
                }
                L.nRays = 1;
                const uint32_t nextChild = __float_as_uint(f4.x) + batchKids;
                frames[L.loopDepth * frameLevel + 16] = __uint_as_float(nextChild);
                if (nextChild < N) { // the loop's remaining children: another batch, or a last single child as an ordinary ray
                    L.state = PS_GLOSSY;
                } else {
                    L.loopDepth = enclosing;
                    L.state = PS_POP;
                }
            }
        }
    } while (__syncthreads_or(L.state > PS_TRACE));
}

// The cameras of a launch sequence that renders several frames, from the kernel's arguments into device memory (a lane
// reads the camera of its pixel's frame; an argument array indexed per lane would be copied to scratch memory).
struct CameraPack {
    rt_camera c[RT_MAX_FRAMES];
};
__global__ __launch_bounds__(64) void store_cameras_kernel(const CameraPack pack, rt_camera* out)
{
    if (threadIdx.x < RT_MAX_FRAMES) out[threadIdx.x] = pack.c[threadIdx.x];
}

// Round 0: camera rays.  Most pixels of a frame look past the scene: only a pixel whose ray enters the root
// box (or that has further pixel samples to take) gets a recursion state and a ray to trace.
__global__ __launch_bounds__(kBlock) void camera_kernel(const WaveParams P)
{
    constexpr int kPer = 4; // pixels per thread and trip: the active list's counter is updated once per 1024 pixels
    __shared__ uint32_t s_n[4 * kPer], s_base;
    const rt_features& F = P.F;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nPixelRays = pixel_ray_count(F);
    uint32_t cPrimary = 0, cPixels = 0, cDrop = 0;
    TraceCounters tc { 0, 0, 0 };
    // pixel of work item w and its first ray; false outside the frame
    auto pixel_setup = [&](const uint32_t w, PathLane& L) -> bool {
        if (w >= P.nWork) return false;
        const uint32_t tile = w >> 6, l = w & 63;
        uint32_t m = tile / P.tilesX, frame = 0;
        const uint32_t tx = tile - m * P.tilesX;
        if (P.frames > 1) frame = m / (uint32_t)P.myTileRows, m -= frame * (uint32_t)P.myTileRows;
        const uint32_t blk = m / P.stripeTileRows, r = m - blk * P.stripeTileRows;
        const uint32_t ty = (blk * P.world + P.rank) * P.stripeTileRows + r;
        const uint32_t px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        if (!(px < (uint32_t)P.W && py < (uint32_t)P.H)) return false;
        // Screen::indexAt (src/screen.cpp:119-122), in the frame's own image
        L.pixel = frame * (uint32_t)(P.W * P.H) + (uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px;
        L.seed = static_cast<uint32_t>(P.H * (int)px + (int)py);      // src/render.cpp:38
        L.smp = L.seed;
        L.pxy = px | ((frame * (uint32_t)P.H + py) << 16);
        for (uint32_t k = 0; k < (F.numPixelSamples > 1 ? 2 * nPixelRays : 0u); k++) (void)sampler_next(L.smp);
        make_pixel_ray(P, L, 0, nPixelRays);
        return true;
    };
    for (uint32_t base = blockIdx.x * (kBlock * kPer); base < P.nWork; base += gridDim.x * (kBlock * kPer)) { // block-uniform trip count
        unsigned long long mask[kPer];
        bool live[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            PathLane L;
            L.nRays = 1, L.rayBase = 0;
            live[k] = false;
            if (pixel_setup(base + k * kBlock + threadIdx.x, L)) {
                cPixels++;
                cPrimary++;
                const uint32_t px = L.pxy & 0xffffu, py = L.pxy >> 16;
                if (P.primId) P.primId[py * P.W + px] = kNoHit;
                if (P.primT) P.primT[py * P.W + px] = FLT_MAX;
                live[k] = nPixelRays > 1 || !F.enableAccelStructure;
                if (!live[k]) { // the root test every query starts with (src/bvh.cpp:264-268)
                    const RaySlabs R = make_slabs(P.S, L.to, L.td);
                    float tr;
                    bool outside;
                    live[k] = box_test_any(P.S.rootLo[0], P.S.rootLo[1], P.S.rootLo[2], P.S.rootHi[0], P.S.rootHi[1], P.S.rootHi[2], R, FLT_MAX,
                                           tr, outside);
                }
            }
            mask[k] = __ballot(live[k]);
            if (lane == 0) s_n[k * 4 + wv] = (uint32_t)__popcll(mask[k]);
        }
        // one counter update per workgroup and trip: state slot = ray slot = position in the first active list
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t n = 0;
            for (int k = 0; k < 4 * kPer; k++) n += s_n[k];
            s_base = n ? atomicAdd(P.activeOutCount, n) : 0u; // also the number of rays: the host reads round 0's from here
        }
        __syncthreads();
        uint32_t run = s_base;
        uint32_t pos[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++)
            for (int w = 0; w < 4; w++) {
                if (w == wv) pos[k] = run + (uint32_t)__popcll(mask[k] & ((1ull << lane) - 1ull));
                run += s_n[k * 4 + w];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            if (!live[k]) continue;
            PathLane L;
            L.nRays = 1;
            (void)pixel_setup(base + k * kBlock + threadIdx.x, L);
            L.state = PS_TRACE;
            L.rayBase = pos[k];
            if (pos[k] < P.rayCap) store_ray(P.pathRays + pos[k], L.to, L.td, FLT_MAX, 0u);
            else cDrop++;
            store_header(P.states + (size_t)pos[k] * P.stateStride, L);
            P.activeOut[pos[k]] = pos[k];
            P.activeRayOut[pos[k]] = pos[k];
        }
    }
    flush_counters(P.stats, lane, cPrimary, 0, 0, tc, cPixels, cDrop);
}

// Round r >= 1, after trace_kernel: consume the hit of every active state's ray, advance its recursion.
#ifndef RT_LOGIC_WAVES
#define RT_LOGIC_WAVES 2 // waves per SIMD the path logic is compiled for (3: 168 registers, 444 B per lane spilled, a pooled frame 7 % slower)
#endif
template <bool TRANSP>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_LOGIC_WAVES))) void path_logic_kernel(const WaveParams P)
{
    __shared__ uint32_t s_n[12], s_base[3];
    const int lane = threadIdx.x & 63;
    const uint32_t front = *P.activeInCount, total = front + (P.activeInCountBack ? *P.activeInCountBack : 0u);
    uint32_t cPrimary = 0, cSecondary = 0, cDrop = 0;
    TraceCounters tc { 0, 0, 0 };
    for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        PathLane L;
        L.state = PS_DONE;
        L.nRays = 1, L.rayBase = 0;
        float* frames = nullptr;
        float* passes = nullptr;
        uint32_t slot = 0;
        if (i < total) {
            const uint32_t at = i < front ? i : P.stateCap - 1u - (i - front); // (total <= stateCap: the two ends do not meet)
            slot = P.activeIn[at];
            load_header(P.states + (size_t)slot * P.stateStride, L);
            frames = P.frameBase + (size_t)slot * kFrameDwords;
            if (TRANSP) passes = P.passBase + (size_t)slot * kPassDwords;
            L.rayBase = P.activeRayIn[at];
            if (L.nRays > 1) {
                L.state = PS_BATCH; // (a batch's rays were reserved as a whole: all of them lie below rayCap)
            } else if (L.rayBase < P.rayCap) {
                const HitRec h = P.hits[L.rayBase];
                L.tt = h.t;
                L.ttri = h.tri;
                L.state = PS_HIT;
            } // else: the ray found no room in the last round (counted there, the frame is reported incomplete) - no hit record exists
              // for it, the pixel ends here
        }
        auto alloc = [&](const bool) {}; // every state of a round already owns its frame storage
        advance_lanes<TRANSP>(P, L, frames, passes, alloc, lane, cPrimary, cSecondary, cDrop, s_n, s_base);
        const bool active = (L.state == PS_TRACE);
        const bool single = active && L.nRays <= 1; // a batch reserved its rays when it was emitted
        uint32_t posFront, posBack, rpos;
        reserve_slots3(P.activeOutCount, 0xffffffffu, (active && !single) ? 1u : 0u, P.activeOutCountBack, 0xffffffffu, single ? 1u : 0u, P.rayCountOut,
                       0xffffffffu, single ? 1u : 0u, lane, s_n, s_base, posFront, posBack, rpos);
        if (single) {
            L.nRays = 1;
            L.rayBase = rpos;
            if (rpos < P.rayCap) store_ray(P.pathRays + rpos, L.to, L.td, FLT_MAX, 0u);
            else cDrop++;
        }
        if (active) { // (every state is on at most one list: front + back <= stateCap)
            const uint32_t at = single ? P.stateCap - 1u - posBack : posFront;
            store_header(P.states + (size_t)slot * P.stateStride, L);
            P.activeOut[at] = slot;
            P.activeRayOut[at] = L.rayBase;
        }
    }
    flush_counters(P.stats, lane, cPrimary, 0, cSecondary, tc, 0, cDrop);
}

// ---------------------------------------------------------------------------------------------
// Group traversal: G lanes share one ray.
//
// The closest hit of the reference is a pure function of the set of leaves whose ancestors' boxes the
// ray enters, and of the order in which its depth-first walk meets them: the smallest t wins, and among
// equal t the triangle tested LAST (intersectRayTriangle accepts t <= ray.t, src/bvh.cpp:235-370).  The
// walk's order does not depend on what was hit before, so the G lanes may visit nodes in any order as
// long as every candidate carries its place in the reference's order.  That place is the path from the
// root written as a heap index: 1 for the root, 2k for the child visited first, 2k+1 for the child
// visited second; left-aligned, two leaves compare like their positions in the depth-first walk.
// A group keeps a stack of pending inner nodes and a queue of pending leaves in LDS.  A node step pops
// up to G nodes (one pair of boxes per lane) and pushes the children the ray enters; a leaf step pops up
// to G leaves (one per lane) and is taken when G leaves wait or no node is left, so that each step runs
// one kind of work on all lanes.  A ray that needs several hundred dependent steps on one lane needs a
// few tens here - the latency of the frame's longest pixel chain shrinks by that factor.
// ---------------------------------------------------------------------------------------------
template <int G, bool KEYS = true>
struct GroupLayout {
    // node entries per group; a query-tree launch keeps no order keys and affords 4-lane groups a deeper stack
    static constexpr int kQ = (G <= 4) ? (KEYS ? 32 : 48) : (G <= 16) ? 64 : 4 * G;
    static constexpr int kStride = kQ + 8;             // skewed so that groups start on different banks
    static constexpr int kLeafQ = 3 * G + 8;           // a node step adds at most 2G leaves to fewer than G
    static constexpr int kEntryOff = KEYS ? 2 * kStride : kStride; // ref[], (key[],) entry distance[]
    static constexpr int kNodeWords = kEntryOff + kStride;
    static constexpr int kLeafWords = (KEYS ? 2 : 1) * kLeafQ; // ref[] (, key[])
    static constexpr int kGroupWords = kNodeWords + kLeafWords;
    static constexpr int kGroupsPerBlock = kBlock / G;
    static constexpr int kWords = kGroupsPerBlock * kGroupWords;
    // deepest tree whose reserve (depth + 2 entries) leaves room to pop G at a time; keys are 32-bit heap indices
    static constexpr int kMaxTreeDepth = KEYS ? (kQ - 2 - G < 30 ? kQ - 2 - G : 30) : kQ - 2 - G;
};

struct GroupLane {
    RaySlabs R;
    float bestT, cullT; // this lane's best hit; the group's best (for culling)
    float tieT;         // query tree: distance at which this lane saw two triangles tie (-1: none)
    uint32_t bestKey, bestTri;
    int nq, lq;         // pending nodes / leaves of the group (same value on every lane of the group)
};

// Starts a query on every lane of a group.  Returns true when traversal steps remain.
template <int G, bool QUERY>
__device__ __forceinline__ bool group_begin(const SceneView& S, GroupLane& T, uint32_t* const q, const int j, const vec3 o, const vec3 d,
                                            const float tmax)
{
    using Lay = GroupLayout<G, !QUERY>;
    T.R = make_slabs(S, o, d);
    T.bestT = tmax, T.cullT = tmax;
    T.bestKey = 0, T.bestTri = kNoHit;
    T.tieT = -1.0f;
    T.nq = 0, T.lq = 0;
    float tr;
    bool outside;
    // root box against the ray's own t; a miss ends the query before the sphere loop (src/bvh.cpp:264-268)
    if (!box_test_any(S.rootLo[0], S.rootLo[1], S.rootLo[2], S.rootHi[0], S.rootHi[1], S.rootHi[2], T.R, tmax, tr, outside)) return false;
    if (S.rootRef & kLeafBit) {
        if (j == 0) {
            q[Lay::kNodeWords] = S.rootRef;
            if (!QUERY) q[Lay::kNodeWords + Lay::kLeafQ] = 1u;
        }
        T.lq = 1;
    } else {
        if (j == 0) {
            q[0] = S.rootRef, q[Lay::kEntryOff] = __float_as_uint(0.0f);
            if (!QUERY) q[Lay::kStride] = 1u;
        }
        T.nq = 1;
    }
    return true;
}

// Node step for the groups with `active`: pop up to G inner nodes, push the children the ray enters.
template <bool CULL, bool COUNT, int G, bool QUERY>
__device__ __forceinline__ void group_node_step(const SceneView& S, GroupLane& T, uint32_t* const q, const int j, const int groupShift,
                                                const int reserve, const bool active, TraceCounters& tc)
{
    using Lay = GroupLayout<G, !QUERY>;
    constexpr uint32_t kGroupMask = (uint32_t)((1ull << G) - 1ull);
    uint32_t* const lq = q + Lay::kNodeWords;
    int n = T.nq < G ? T.nq : G;
    const int room = Lay::kQ - reserve - T.nq;
    if (n > room) n = room < 1 ? 1 : room; // a nearly full stack is walked one entry per step: growth is then bounded by the tree depth
    const int base = T.nq - n;
    bool take = active && j < n;
    uint32_t ref = 0, key = 0;
    if (take) {
        const int e = T.nq - 1 - j;
        ref = q[e];
        if (!QUERY) key = q[Lay::kStride + e];
        const float tentry = __uint_as_float(q[Lay::kEntryOff + e]);
        if (CULL && tentry > T.cullT) take = false;
    }
    bool pushFar = false, pushNear = false;
    uint32_t farRef = 0, farKey = 0, nearRef = 0, nearKey = 0;
    float farT = 0.0f, nearT = 0.0f;
    if (take) {
        const float4* p = reinterpret_cast<const float4*>(S.pairs + ref);
        const float4 a = p[0], b = p[1], c = p[2], e = p[3];
        if (COUNT) tc.nodePairs++;
        float tl, trr;
        bool outL = true, outR = true, hl, hr;
        if (QUERY) pair_test_loose(a, b, c, e, T.R, hl, hr, tl, trr);
        else pair_test(a, b, c, e, T.R, hl, hr, tl, trr, outL, outR);
        const uint32_t ref0 = __float_as_uint(b.z), ref1 = __float_as_uint(e.z);
        // the child the reference visits first (src/bvh.cpp:301-310: nearer first, equal distances: right first)
        const bool leftFirst = hl && (!hr || tl < trr);
        // entry distances kept for culling: a box holding the origin is never culled; the others only beyond the error
        // bound of pair_test's approximate distances
        const float el = outL ? tl - T.R.eps : 0.0f, er = outR ? trr - T.R.eps : 0.0f;
        const bool keepL = hl && !(CULL && el > T.cullT), keepR = hr && !(CULL && er > T.cullT);
        const uint32_t keyFirst = 2u * key, keySecond = (hl && hr) ? 2u * key + 1u : 2u * key;
        if (leftFirst) {
            pushNear = keepL, nearRef = ref0, nearKey = keyFirst, nearT = el;
            pushFar = keepR, farRef = ref1, farKey = keySecond, farT = er;
        } else {
            pushNear = keepR, nearRef = ref1, nearKey = keyFirst, nearT = er;
            pushFar = keepL, farRef = ref0, farKey = keySecond, farT = el;
        }
    }
    const bool farLeaf = pushFar && (farRef & kLeafBit), nearLeaf = pushNear && (nearRef & kLeafBit);
    const bool farNode = pushFar && !farLeaf, nearNode = pushNear && !nearLeaf;
    const uint32_t mFarN = (uint32_t)(__ballot(farNode) >> groupShift) & kGroupMask;
    const uint32_t mNearN = (uint32_t)(__ballot(nearNode) >> groupShift) & kGroupMask;
    const uint32_t mFarL = (uint32_t)(__ballot(farLeaf) >> groupShift) & kGroupMask;
    const uint32_t mNearL = (uint32_t)(__ballot(nearLeaf) >> groupShift) & kGroupMask;
    if (!active) return;
    const uint32_t below = (1u << j) - 1u;
    const int nFarN = __popc(mFarN), nFarL = __popc(mFarL);
    __builtin_amdgcn_wave_barrier(); // every pop of this step precedes its pushes
    if (farNode) {
        const int e = base + __popc(mFarN & below);
        q[e] = farRef, q[Lay::kEntryOff + e] = __float_as_uint(farT);
        if (!QUERY) q[Lay::kStride + e] = farKey;
    }
    if (nearNode) { // nearer children on top: they are popped first
        const int e = base + nFarN + __popc(mNearN & below);
        q[e] = nearRef, q[Lay::kEntryOff + e] = __float_as_uint(nearT);
        if (!QUERY) q[Lay::kStride + e] = nearKey;
    }
    if (farLeaf) {
        const int e = T.lq + __popc(mFarL & below);
        lq[e] = farRef;
        if (!QUERY) lq[Lay::kLeafQ + e] = farKey;
    }
    if (nearLeaf) {
        const int e = T.lq + nFarL + __popc(mNearL & below);
        lq[e] = nearRef;
        if (!QUERY) lq[Lay::kLeafQ + e] = nearKey;
    }
    __builtin_amdgcn_wave_barrier();
    T.nq = base + nFarN + __popc(mNearN);
    T.lq = T.lq + nFarL + __popc(mNearL);
}

// The same step in the query tree's 4-wide form (query-tree launches only): every lane that pops an entry tests the four
// boxes of one Node4 (as the one-ray-per-lane kernel does), so a ray's walk is half as many dependent steps deep.  Children
// go on the stack farthest first - the fourth-nearest children of all lanes, then the third-nearest, ... - so the nearest are
// popped next.  `reserve` = 3 * (depth of the 4-wide tree) + 3: a pop adds at most three entries net, a step pops only as
// many entries as leave that room, and a nearly full stack is walked one entry per step, when the growth is bounded by
// three per level.  Leaves: a pop adds at most four, and a step pops only as many entries as the leaf queue has room for.
template <bool COUNT, int G>
__device__ __forceinline__ void group_node4_step(const SceneView& S, GroupLane& T, uint32_t* const q, const int j, const int groupShift,
                                                 const int reserve, const bool active, TraceCounters& tc)
{
    using Lay = GroupLayout<G, false>;
    constexpr uint32_t kGroupMask = (uint32_t)((1ull << G) - 1ull);
    uint32_t* const lq = q + Lay::kNodeWords;
    int n = T.nq < G ? T.nq : G;
    const int room = (Lay::kQ - reserve - T.nq) / 3;
    if (n > room) n = room < 1 ? 1 : room;
    const int leafRoom = (Lay::kLeafQ - T.lq) / 4; // >= 1: the caller checks
    if (n > leafRoom) n = leafRoom;
    const int base = T.nq - n;
    bool take = active && j < n;
    uint32_t ref = 0;
    if (take) {
        const int e = T.nq - 1 - j;
        ref = q[e];
        if (__uint_as_float(q[Lay::kEntryOff + e]) > T.cullT) take = false;
    }
    Node4Hits h;
    h.key[0] = h.key[1] = h.key[2] = h.key[3] = kMissKey;
    h.ref[0] = h.ref[1] = h.ref[2] = h.ref[3] = kEmptyRef;
    if (take) {
        if (COUNT) tc.node4++;
        h = node4_test(S.nodes4 + ref, T.R, T.cullT);
    }
    __builtin_amdgcn_wave_barrier(); // every pop of this step precedes its pushes
    int nq = base, nl = T.lq;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        const bool hit = h.key[k] != kMissKey;
        const uint32_t cref = hit ? node4_ref(h, h.key[k]) : 0u;
        const bool isLeaf = hit && (cref & kLeafBit), isNode = hit && !(cref & kLeafBit);
        const uint32_t mN = (uint32_t)(__ballot(isNode) >> groupShift) & kGroupMask;
        const uint32_t mL = (uint32_t)(__ballot(isLeaf) >> groupShift) & kGroupMask;
        const uint32_t below = (1u << j) - 1u;
        if (isNode) {
            const int e = nq + __popc(mN & below);
            q[e] = cref, q[Lay::kEntryOff + e] = __float_as_uint(node4_entry(h.key[k]) - T.R.eps);
        }
        if (isLeaf) lq[nl + __popc(mL & below)] = cref;
        nq += __popc(mN), nl += __popc(mL);
    }
    __builtin_amdgcn_wave_barrier();
    if (active) T.nq = nq, T.lq = nl;
}

// Leaf step for the groups with `active`: pop up to G leaves, test their triangles, refresh the group's best.
template <bool COUNT, int G, bool QUERY, bool ANY = false>
__device__ __forceinline__ void group_leaf_step(const SceneView& S, GroupLane& T, uint32_t* const q, const int j, const bool active,
                                                TraceCounters& tc)
{
    using Lay = GroupLayout<G, !QUERY>;
    uint32_t* const lq = q + Lay::kNodeWords;
    const int n = T.lq < G ? T.lq : G;
    if (active && j < n) {
        const int e = T.lq - 1 - j;
        const uint32_t ref = lq[e], key = QUERY ? 1u : lq[Lay::kLeafQ + e];
        const uint32_t first = ref & kRefOffsetMask, cnt = QUERY ? (ref >> kRefCountShift) & kMaxLeafCount : leaf_count(S, ref);
        float tl = T.cullT; // the closest hit any lane of the group holds: nothing beyond it can win (equal distances stay in)
        uint32_t tril = kNoHit;
        // query-tree launches: one triangle in flight at a time keeps the kernel free of register spills at 4 waves per SIMD
        if (QUERY) leaf_test_narrow<COUNT, !ANY>(S, first, cnt, T.R.o, T.R.d, tl, tril, tc, &T.tieT);
        else leaf_test<COUNT, false>(S, first, cnt, T.R.o, T.R.d, tl, tril, tc, &T.tieT);
        if (tril != kNoHit) {
            if (QUERY) { // the walk's order means nothing here: equal distances are only noted
                if (!ANY && tl == T.bestT && T.bestTri != kNoHit) T.tieT = tl;
                T.bestT = tl, T.bestTri = tril;
            } else {
                const uint32_t aligned = key << __clz(key);
                // tl <= bestT here: a closer hit wins, an equal one wins if the reference meets this leaf later
                if (tl < T.bestT || aligned > T.bestKey) T.bestT = tl, T.bestKey = aligned, T.bestTri = tril;
            }
        }
    }
    if (active) T.lq -= n;
    float bt = T.bestT;
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) bt = fminf(bt, __shfl_xor(bt, off));
    if (active) T.cullT = bt;
    if (ANY && active && bt < T.tieT) T.nq = 0, T.lq = 0; // (tieT holds any_occluded(D) here) an occluder: the answer is settled
}

// The group's result on every lane of the group (all lanes of the wave must call).
template <int G>
__device__ __forceinline__ void group_finish(const SceneView& S, GroupLane& T, const bool active)
{
    float bt = T.bestT;
    uint32_t bk = T.bestKey, bi = T.bestTri;
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const float ot = __shfl_xor(bt, off);
        const uint32_t ok = __shfl_xor(bk, off), oi = __shfl_xor(bi, off);
        if (ot < bt || (ot == bt && ok > bk)) bt = ot, bk = ok, bi = oi;
    }
    if (!active) return;
    T.bestT = bt, T.bestKey = bk, T.bestTri = bi;
    for (uint32_t s = 0; s < S.nSpheres; s++) // src/bvh.cpp:366-367
        if (sphere_test(S.spheres[s], T.R.o, T.R.d, T.bestT)) T.bestTri = kSphereBase + s;
}

// The same for a query-tree launch: the closest triangle of the group, and whether the reference provably reports it
// (query_result_proven; two lanes holding different triangles at the closest distance are a tie as well).
template <int G, bool ANY = false>
__device__ __forceinline__ bool group_finish_query(const SceneView& S, GroupLane& T, const int groupShift, const bool active)
{
    constexpr uint32_t kGroupMask = (uint32_t)((1ull << G) - 1ull);
    float bt = T.bestT;
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) bt = fminf(bt, __shfl_xor(bt, off));
    const uint32_t holders = (uint32_t)(__ballot(T.bestTri != kNoHit && T.bestT == bt) >> groupShift) & kGroupMask;
    const uint32_t tied = ANY ? 0u : (uint32_t)(__ballot(T.tieT == bt) >> groupShift) & kGroupMask;
    const int winner = holders ? __ffs(holders) - 1 : 0;
    uint32_t bi = __shfl(T.bestTri, groupShift + winner);
    if (!holders) bi = kNoHit;
    if (!active) return true;
    // (a shadow query asks for the distance alone: which of several triangles at the closest distance is reported does not matter)
    const bool proven = (ANY || __popc(holders) <= 1) && tied == 0u && query_result_proven(S, T.R, bt, bi, -1.0f, 0u);
    T.bestT = bt, T.bestTri = bi;
    for (uint32_t s = 0; s < S.nSpheres; s++) // src/bvh.cpp:366-367
        if (sphere_test(S.spheres[s], T.R.o, T.R.d, T.bestT)) T.bestTri = kSphereBase + s;
    return proven;
}

// trace_kernel with G lanes per ray.  `reserve` = tree depth + 2: the stack room kept free so that a nearly
// full stack can always be walked one entry at a time.
#ifndef RT_GROUP_WAVES
#define RT_GROUP_WAVES 4 // waves per SIMD the group kernel is compiled for (5: 96 registers, 49 spilled, a frame 18 % slower)
#endif
template <bool CULL, bool COUNT, int G, bool QUERY = false, bool ANY = false, bool N4 = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_GROUP_WAVES))) void group_trace_kernel(const TraceParams P, const int reserve)
{
    using Lay = GroupLayout<G, !QUERY>;
    __shared__ uint32_t s_q[Lay::kWords];
    const SceneView& S = P.S;
    const int lane = threadIdx.x & 63;
    const int j = lane & (G - 1), leaderLane = lane - j, groupShift = leaderLane;
    uint32_t* const q = s_q + (threadIdx.x / G) * Lay::kGroupWords;
    const uint32_t have = *P.countPtr;
    const uint32_t items = have > P.countSub ? min(have - P.countSub, P.countMaxItems) : 0u;
    const uint32_t total = items * P.countMul;
    if (total < P.runLo || total >= P.runHi) return;
    TraceCounters tc { 0, 0, 0 };
    // The first ray of every group is fixed by the group's position in the grid; later rays come from the shared
    // counter in chunks of kGroupChunk per wave (see trace_kernel).
    const uint32_t gridGroups = gridDim.x * (kBlock / G);
    const uint32_t myGroup = blockIdx.x * (kBlock / G) + threadIdx.x / G;
    bool has = false, exhausted = total <= gridGroups, first = true; // has: same on every lane of a group
    uint32_t idx = 0, chunkNext = 0, chunkEnd = 0;
    GroupLane T;
    T.R.o = T.R.d = T.R.r = splat(0.0f);
    T.R.fast = T.R.approx = false;
    T.R.nor = splat(0.0f), T.R.eps = 0.0f;
    T.bestT = T.cullT = 0.0f, T.bestKey = 0, T.bestTri = kNoHit, T.nq = T.lq = 0;
    T.tieT = -1.0f;
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) / G >= total) return; // nothing for this wave, now or later
    const unsigned long long waveStart = P.waveTimes ? wall_clock64() : 0ull;
    for (;;) {
        const unsigned long long idle = __ballot(!has && j == 0);
        if (idle != 0ull && (first || !exhausted || chunkNext < chunkEnd)) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = myGroup;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.fetch, kGroupChunk);
                    base = __shfl(base, 0) + gridGroups;
                    exhausted = base + kGroupChunk >= total;
                    chunkNext = min(base, total), chunkEnd = min(base + kGroupChunk, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << leaderLane) - 1ull));
                const uint32_t avail = chunkEnd - chunkNext, n = (uint32_t)__popcll(idle);
                if (rank < avail) next = chunkNext + rank;
                chunkNext += min(n, avail);
            }
            first = false;
            bool valid = false;
            if (!has) {
                valid = next < total;
                idx = (valid && P.indirect && total >= P.indirectFrom) ? P.indirect[next] : next;
            }
            bool unfit = false;
            if (valid) {
                const float4* r = reinterpret_cast<const float4*>(P.rays + (size_t)idx * P.rayStride);
                const float4 a = r[0], b = r[1];
                has = group_begin<G, QUERY>(S, T, q, j, make3(a.x, a.y, a.z), make3(b.x, b.y, b.z), a.w);
                if (!has && j == 0) P.hits[idx] = HitRec { T.bestT, kNoHit };
                if (QUERY && has && !T.R.approx) unfit = true, has = false; // the conservative tests need the approximate slabs
                if (ANY && has) T.bestT = T.cullT = fminf(T.bestT, any_bound(b.w)), T.tieT = any_occluded(b.w); // the tag word holds the shaded point's distance
            }
            if (QUERY && __ballot(unfit) != 0ull) {
                const uint32_t pos = wave_fetch(P.fallCount, unfit && j == 0, lane);
                if (unfit && j == 0) P.fallList[pos] = idx;
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted && chunkNext == chunkEnd) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < 4; it++) {
            // A step costs the wave the same however many groups take part: once one group needs a kind of step, every
            // group with work of that kind joins in (nodes only while the leaf queue has room for 2G more).
            const bool needLeaf = has && (T.lq >= G || (T.nq == 0 && T.lq > 0));
            const bool needNode = has && !needLeaf && T.nq > 0;
            if (__ballot(needNode) != 0ull) {
                if (N4) group_node4_step<COUNT, G>(S, T, q, j, groupShift, reserve, has && T.nq > 0 && T.lq + 4 <= Lay::kLeafQ, tc);
                else group_node_step<CULL, COUNT, G, QUERY>(S, T, q, j, groupShift, reserve, has && T.nq > 0 && T.lq + 2 * G <= Lay::kLeafQ, tc);
            }
            if (__ballot(has && (T.lq >= G || (T.nq == 0 && T.lq > 0))) != 0ull)
                group_leaf_step<COUNT, G, QUERY, ANY>(S, T, q, j, has && T.lq > 0, tc);
        }
        // (groups whose walk has ended are settled together here, once per four steps: a group idles until the wave next looks for
        // new rays anyway, and the settling code runs for however few groups need it at the cost of the whole wave)
        {
            const bool done = has && T.nq == 0 && T.lq == 0;
            if (__ballot(done) != 0ull) {
                bool proven = true;
                if (QUERY) proven = group_finish_query<G, ANY>(S, T, groupShift, done);
                else group_finish<G>(S, T, done);
                if (done) {
                    if (j == 0 && proven) P.hits[idx] = HitRec { T.bestT, T.bestTri };
                    has = false;
                }
                if (QUERY && __ballot(done && !proven) != 0ull) {
                    const uint32_t pos = wave_fetch(P.fallCount, done && !proven && j == 0, lane);
                    if (done && !proven && j == 0) P.fallList[pos] = idx;
                }
            }
        }
    }
    if (P.waveTimes && lane == 0) {
        const size_t wv = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        P.waveTimes[2 * wv] = waveStart, P.waveTimes[2 * wv + 1] = wall_clock64();
    }
    if (COUNT) flush_counters(P.stats, lane, 0, 0, 0, tc, 0, 0);
}

// Completes the raw tasks of the shade queue (store_raw_shade_task): renderRay after the intersection for a child of a
// batched glossy loop (src/recursive.cpp:29-36, 62-80; src/extra.cpp:464-508), everything but the sampler's progress.
__global__ __launch_bounds__(kBlock) void shade_expand_kernel(const WaveParams P)
{
    const SceneView& S = P.S;
    const uint32_t total = min(P.qCount[0], P.shadeCap);
    const uint32_t N = P.F.numGlossySamples;
    TraceCounters tc { 0, 0, 0 };
    uint32_t cSecondary = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const float4* q = reinterpret_cast<const float4*>(P.shadeQ + i);
        const float4 a = q[0];
        if (__float_as_uint(a.y) != kRawShade || __float_as_uint(a.x) == kHolePixel) continue;
        const float4 b = q[1], c = q[2], d = q[3];
        const uint32_t pixel = __float_as_uint(a.x), smp = __float_as_uint(a.z), tri = __float_as_uint(a.w);
        const vec3 W = make3(b.x, b.y, b.z), co = make3(b.w, c.x, c.y), cd = make3(c.z, c.w, d.x);
        const float t = d.y;
        const uint32_t leafSlot = __float_as_uint(d.z);
        Hit h;
        fill_hit(S, P.F, tri, co, cd, t, h);
        const vec3 p = co + t * cd;
        store_shade_task(P.shadeQ + i, pixel, h.mesh | (h.isSphere ? 0x80000000u : 0u), smp, tri, W, p, cd, h.normal);
        if (leafSlot == 0xffffffffu) continue;
        const Mat m = load_material(S, h.mesh, h.isSphere);
        // generateReflectionRay (src/recursive.cpp:91-103)
        vec3 rorg = p;
        const vec3 rdir = cd - 2.f * dot(cd, h.normal) * h.normal;
        rorg = rorg + 0.00001f * rdir;
        // renderRayGlossyComponent's loop (src/extra.cpp:489-505): sample s to slot s, a hole where it is rejected
        const float radius = 0.5f / m.shininess;
        const vec3 lw = W * m.ks / float(N);
        vec3 w, u, v;
        ortho_basis(rdir, w, u, v);
        uint32_t s2 = smp;
        for (uint32_t k = 0; k < N; k++) {
            vec3 pd;
            if (glossy_sample(s2, radius, w, u, v, h.normal, pd)) { // (a raw task's leaf slots all lie below the queue's end: path_logic_kernel)
                cSecondary++;
                store_leaf_task(P.leafQ + leafSlot + k, pixel, rorg + 0.0001f * pd, pd, FLT_MAX, lw);
            } else {
                store_leaf_task(P.leafQ + leafSlot + k, kHolePixel, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, splat(0.0f));
            }
        }
    }
    flush_counters(P.stats, threadIdx.x & 63, 0, 0, cSecondary, tc, 0, 0);
}

// Deferred rays at the recursion limit: a hit becomes a ShadeTask.
__global__ __launch_bounds__(kBlock) void leaf_logic_kernel(const WaveParams P)
{
    constexpr int kPer = 4; // tasks per thread and trip: the shade queue's counter is updated once per 1024 tasks
    __shared__ uint32_t s_n[4 * kPer], s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t total = min(P.qCount[1], P.leafCap);
    TraceCounters tc { 0, 0, 0 };
    uint32_t cDrop = 0;
    for (uint32_t base = blockIdx.x * (kBlock * kPer); base < total; base += gridDim.x * (kBlock * kPer)) { // block-uniform trip count
        HitRec hr[kPer];
        unsigned long long mask[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const uint32_t i = base + k * kBlock + threadIdx.x;
            hr[k] = HitRec { 0.0f, kNoHit };
            if (i < total) hr[k] = P.hits[i];
            mask[k] = __ballot(hr[k].tri != kNoHit);
            if (lane == 0) s_n[k * 4 + wv] = (uint32_t)__popcll(mask[k]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t n = 0;
            for (int k = 0; k < 4 * kPer; k++) n += s_n[k];
            s_base = n ? atomicAdd(P.qCount + 0, n) : 0u;
        }
        __syncthreads();
        uint32_t run = s_base;
        uint32_t slot[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++)
            for (int w = 0; w < 4; w++) {
                if (w == wv) slot[k] = run + (uint32_t)__popcll(mask[k] & ((1ull << lane) - 1ull));
                run += s_n[k * 4 + w];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            if (hr[k].tri == kNoHit) continue;
            if (slot[k] >= P.shadeCap) {
                cDrop++;
                continue;
            }
            const uint32_t i = base + k * kBlock + threadIdx.x;
            const float4* q = reinterpret_cast<const float4*>(P.leafQ + i);
            const float4 a = q[0], b = q[1], c = q[2];
            const vec3 o = make3(a.x, a.y, a.z), d = make3(b.x, b.y, b.z), W = make3(c.x, c.y, c.z);
            Hit h;
            fill_hit(P.S, P.F, hr[k].tri, o, d, hr[k].t, h);
            store_shade_task(P.shadeQ + slot[k], __float_as_uint(b.w), h.mesh | (h.isSphere ? 0x80000000u : 0u), 0u, hr[k].tri, W,
                             o + hr[k].t * d, d, h.normal);
        }
    }
    flush_counters(P.stats, lane, 0, 0, 0, tc, 0, cDrop);
}

// Position and colour of light sample (li, s) for a hit whose light loop starts at sampler state smp.
// Walks the lights in order so that area-light draws are consumed exactly as computeLightContribution does
// (src/light.cpp:262-276).  Calls fn(rayIndex, light, lpos, lcol) for every sample.
template <typename Fn>
__device__ __forceinline__ void for_each_light_sample(const SceneView& S, const rt_features& F, uint32_t smp, Fn&& fn)
{
    uint32_t j = 0;
    for (uint32_t li = 0; li < S.nLights; li++) {
        const GpuLight& L = S.lights[li];
        const int type = (int)L.type;
        const uint32_t nSamples = (type == 0) ? 1u : F.numShadowSamples;
        for (uint32_t s = 0; s < nSamples; s++, j++) {
            vec3 lpos, lcol;
            if (type == 0) {
                lpos = load3(L.f);
                lcol = load3(L.f + 3);
            } else if (type == 1) { // sampleSegmentLight (src/light.cpp:25-29)
                const float u = sampler_next(smp);
                const vec3 e0 = load3(L.f), e1 = load3(L.f + 3), c0 = load3(L.f + 6), c1 = load3(L.f + 9);
                lpos = e0 + u * (e1 - e0);
                lcol = c0 + u * (c1 - c0);
            } else { // sampleParallelogramLight (src/light.cpp:40-53)
                const float sx = sampler_next(smp);
                const float sy = sampler_next(smp);
                lpos = load3(L.f) + sx * load3(L.f + 3) + sy * load3(L.f + 6);
                const float a0 = (1.0f - sx) * (1.0f - sy), a1 = (1.0f - sx) * sy, a2 = sx * (1.0f - sy), a3 = sx * sy;
                lcol = a0 * load3(L.f + 9) + a1 * load3(L.f + 12) + a2 * load3(L.f + 15) + a3 * load3(L.f + 18);
            }
            fn(j, type, s + 1 == nSamples, lpos, lcol);
        }
    }
}

// Shadow rays of the shade tasks [chunkBase, chunkBase + chunkTasks): visibilityOfLightSampleBinary
// (src/light.cpp:66-90) traces from the light towards the hit point.
__global__ __launch_bounds__(kBlock) void shadow_gen_kernel(const WaveParams P)
{
    const int lane = threadIdx.x & 63;
    const uint32_t have = min(P.qCount[0], P.shadeCap);
    const uint32_t total = have > P.chunkBase ? min(have - P.chunkBase, P.chunkTasks) : 0u;
    uint32_t cShadow = 0;
    TraceCounters tc { 0, 0, 0 };
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const float4* q = reinterpret_cast<const float4*>(P.shadeQ + P.chunkBase + i);
        const float4 a = q[0], b = q[1], c = q[2];
        const vec3 p = make3(b.w, c.x, c.y);
        RayRec* out = P.shadowRays + (size_t)i * P.raysPerTask;
        if (__float_as_uint(a.x) == kHolePixel) { // unused slot: rays that miss at once
            for (uint32_t j = 0; j < P.raysPerTask; j++) store_ray(out + j, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, 0u);
            continue;
        }
        for_each_light_sample(P.S, P.F, __float_as_uint(a.z), [&](const uint32_t j, int, bool, const vec3 lpos, const vec3) {
            // the tag word: the distance shade_logic_kernel will compare the hit with (see any_bound)
            store_ray(out + j, lpos, normalize(p - lpos), FLT_MAX, __float_as_uint(length(p - lpos)));
            cShadow++;
        });
    }
    flush_counters(P.stats, lane, 0, cShadow, 0, tc, 0, 0);
}

// visibilityOfLightSampleTransparency (src/light.cpp:107-136) for the shadow rays of the chunk: a lane follows its light
// sample from the light through every surface on the way to the shaded point; each surface tints and dims what gets past
// it, and what reached the LAST surface met is what the sample contributes.  The walk is a chain of closest-hit queries
// in the caller's tree, each starting just past the previous hit.  The reference's walk ends only when a surface is met
// within 1e-5 of the shaded point or nothing is hit; where it would oscillate around the shaded surface for ever, this
// one stops after kMaxShadowSegments surfaces.  The colour that arrives replaces the ray's origin in its record.
template <bool CULL, bool COUNT, int STACK>
__global__ __launch_bounds__(kBlock) void shadow_walk_kernel(const WaveParams P, const TraceParams Q)
{
    __shared__ uint32_t s_stack[2 * STACK * kBlock];
    uint32_t* const stack = s_stack + threadIdx.x;
    const SceneView& S = Q.S;
    const int lane = threadIdx.x & 63;
    const uint32_t have = min(P.qCount[0], P.shadeCap);
    const uint32_t tasks = have > P.chunkBase ? min(have - P.chunkBase, P.chunkTasks) : 0u;
    const uint32_t total = tasks * P.raysPerTask;
    TraceCounters tc { 0, 0, 0 };
    uint32_t cShadow = 0;
    const uint32_t gridLanes = gridDim.x * kBlock;
    bool has = false, exhausted = total <= gridLanes, first = true;
    uint32_t idx = 0, chunkNext = 0, chunkEnd = 0, segments = 0;
    vec3 ip = splat(0.0f), incoming = splat(0.0f), prev = splat(0.0f), org = splat(0.0f);
    TraceLane T;
    trace_lane_clear(T);
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) >= total) return; // nothing for this wave, now or later
    // one surface met (or none): true when the walk goes on with another query
    auto walk_step = [&]() -> bool {
        const bool intersects = T.tri != kNoHit;
        const bool finalRayT = fabsf(T.t - length(ip - org)) <= 1e-5f;
        if (intersects) {
            const bool sph = T.tri >= kSphereBase;
            const Mat hm = load_material(S, sph ? T.tri - kSphereBase : S.triAttr[T.tri].meshID, sph);
            prev = incoming;
            incoming = incoming * hm.kd * (1.0f - hm.transparency);
            org = org + (T.t + 1e-6f) * T.R.d;
        }
        return intersects && !finalRayT && segments < kMaxShadowSegments;
    };
    for (;;) {
        const unsigned long long idle = __ballot(!has);
        if (idle != 0ull && (first || !exhausted || chunkNext < chunkEnd)) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = blockIdx.x * kBlock + threadIdx.x;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(Q.fetch, kLaneChunk);
                    base = __shfl(base, 0) + gridLanes;
                    exhausted = base + kLaneChunk >= total;
                    chunkNext = min(base, total), chunkEnd = min(base + kLaneChunk, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t avail = chunkEnd - chunkNext, n = (uint32_t)__popcll(idle);
                if (rank < avail) next = chunkNext + rank;
                chunkNext += min(n, avail);
            }
            first = false;
            if (!has) idx = next;
            if (!has && idx < total) {
                const float4* q = reinterpret_cast<const float4*>(P.shadowRays + idx);
                const float4 a = q[0], b = q[1];
                if (a.w < 0.0f) { // the slot of an unused task: nothing to follow
                } else {
                    // the shaded point and the sample's colour, as shadow_gen_kernel saw them
                    const uint32_t task = idx / P.raysPerTask, j = idx - task * P.raysPerTask;
                    const float4* tq = reinterpret_cast<const float4*>(P.shadeQ + P.chunkBase + task);
                    const float4 ta = tq[0], tb = tq[1], tcw = tq[2];
                    ip = make3(tb.w, tcw.x, tcw.y);
                    for_each_light_sample(S, P.F, __float_as_uint(ta.z), [&](const uint32_t jj, int, bool, const vec3, const vec3 lcol) {
                        if (jj == j) incoming = prev = lcol;
                    });
                    org = make3(a.x, a.y, a.z);
                    segments = 1;
                    for (;;) { // queries that end at the root box are answered on the spot
                        has = trace_begin<COUNT>(S, Q.useBvh != 0, T, org, segments == 1 ? make3(b.x, b.y, b.z) : normalize(ip - org), FLT_MAX, tc);
                        if (has) break;
                        if (!walk_step()) break;
                        segments++, cShadow++;
                    }
                    if (!has) {
                        float4* o = reinterpret_cast<float4*>(P.shadowRays + idx);
                        o[0] = make_float4(prev.x, prev.y, prev.z, 0.0f);
                    }
                }
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted && chunkNext == chunkEnd) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < kStepsPerCheck; it++) {
            const bool ended = trace_wave_step<CULL, COUNT, STACK, false>(S, has, T, stack, tc);
            if (ended) {
                has = false;
                while (walk_step()) {
                    segments++, cShadow++;
                    has = trace_begin<COUNT>(S, Q.useBvh != 0, T, org, normalize(ip - org), FLT_MAX, tc);
                    if (has) break;
                }
                if (!has) {
                    float4* o = reinterpret_cast<float4*>(P.shadowRays + idx);
                    o[0] = make_float4(prev.x, prev.y, prev.z, 0.0f);
                }
            }
        }
    }
    flush_counters(P.stats, lane, 0, cShadow, 0, tc, 0, 0);
}

// computeLightContribution (src/light.cpp:262-276) for the shade tasks of the chunk, with the shadow
// queries answered by P.hits (null when shadows are off).
__global__ __launch_bounds__(kBlock) void shade_logic_kernel(const WaveParams P)
{
    const SceneView& S = P.S;
    const rt_features& F = P.F;
    const uint32_t have = min(P.qCount[0], P.shadeCap);
    const uint32_t total = have > P.chunkBase ? min(have - P.chunkBase, P.chunkTasks) : 0u;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const float4* q = reinterpret_cast<const float4*>(P.shadeQ + P.chunkBase + i);
        const float4 a = q[0], b = q[1], c = q[2], e = q[3];
        const uint32_t pixel = __float_as_uint(a.x), matInfo = __float_as_uint(a.y);
        if (pixel == kHolePixel) continue;
        const vec3 W = make3(b.x, b.y, b.z), p = make3(b.w, c.x, c.y), rd = make3(c.z, c.w, e.x), hn = make3(e.y, e.z, e.w);
        Mat m = load_material(S, matInfo & 0x7fffffffu, (matInfo >> 31) != 0);
        if (F.enableTextureMapping && !(matInfo >> 31)) // sampleMaterialKd (src/shading.cpp:13-24) at the hit's texture coordinates
            m.kd = sample_material_kd(S, F, matInfo, tri_uv_at(S, __float_as_uint(a.w), p), m.kd);
        const HitRec* hits = P.hits ? P.hits + (size_t)i * P.raysPerTask : nullptr;
        vec3 Lo = splat(0.0f), areaAcc = splat(0.0f);
        for_each_light_sample(S, F, __float_as_uint(a.z), [&](const uint32_t j, const int type, const bool last, const vec3 lpos, const vec3 lcol) {
            vec3 visible = lcol;
            if (hits && F.enableTransparency) { // what shadow_walk_kernel left in the sample's ray record
                const float4 v = reinterpret_cast<const float4*>(P.shadowRays + (size_t)i * P.raysPerTask + j)[0];
                visible = make3(v.x, v.y, v.z);
            } else if (hits) {
                const HitRec h = hits[j];
                const bool expected = fabsf(h.t - length(p - lpos)) <= 5e-4f;
                if (!(h.tri != kNoHit && expected)) visible = splat(0.0f);
            }
            if (type == 0) { // computeContributionPointLight (src/light.cpp:151-159)
                Lo = Lo + shade(F, -rd, normalize(lpos - p), visible, hn, m);
            } else { // the sampled POSITION is handed over as the light direction (src/light.cpp:191, 228)
                areaAcc = areaAcc + shade(F, -rd, lpos, visible, hn, m) / (1.0f * F.numShadowSamples);
                if (last) {
                    Lo = Lo + areaAcc;
                    areaAcc = splat(0.0f);
                }
            }
        });
        accumulate_pixel(P, pixel, W * Lo);
    }
}

// ---------------------------------------------------------------------------------------------
// Resolve: fixed-point sums -> float RGB (Screen order) for this rank's stripes; clears the sums.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void resolve_kernel(const WaveParams P)
{
    for (uint32_t w = blockIdx.x * kBlock + threadIdx.x; w < P.nWork; w += gridDim.x * kBlock) {
        const uint32_t tile = w >> 6, l = w & 63;
        uint32_t m = tile / P.tilesX, frame = 0;
        const uint32_t tx = tile - m * P.tilesX;
        if (P.frames > 1) frame = m / (uint32_t)P.myTileRows, m -= frame * (uint32_t)P.myTileRows;
        const uint32_t blk = m / P.stripeTileRows, r = m - blk * P.stripeTileRows;
        const uint32_t ty = (blk * P.world + P.rank) * P.stripeTileRows + r;
        const uint32_t px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        if (px >= (uint32_t)P.W || py >= (uint32_t)P.H) continue;
        const uint32_t pixel = frame * (uint32_t)(P.W * P.H) + (uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px;
        long long* a = P.accum + 3 * (size_t)pixel;
        const uint32_t nf = P.nonfinite[pixel];
        float out[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            out[ch] = (float)((double)a[ch] * (1.0 / kFixScale));
            a[ch] = 0;
        }
        if (nf) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const bool nan = (nf >> ch) & 1u, pinf = (nf >> (3 + ch)) & 1u, ninf = (nf >> (6 + ch)) & 1u;
                if (nan || (pinf && ninf)) out[ch] = __uint_as_float(0x7fc00000u);
                else if (pinf) out[ch] = __uint_as_float(0x7f800000u);
                else if (ninf) out[ch] = __uint_as_float(0xff800000u);
            }
            P.nonfinite[pixel] = 0;
        }
        float* o = P.rgb + 3 * (size_t)pixel;
        o[0] = out[0], o[1] = out[1], o[2] = out[2];
    }
}
