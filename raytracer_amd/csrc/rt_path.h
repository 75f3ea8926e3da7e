// Path stage without rounds (included by rt_device.hip after rt_wavefront.h, inside its anonymous namespace).
//
// The sequential part of a pixel - the radiance rays whose outcome decides where the pixel's sampler stands when the
// next glossy sample is drawn (src/extra.cpp:489-505) - is a chain of dependent closest-hit queries: up to 85 for glossy
// x4 from rayDepth 2.  Walking that chain in frame-wide rounds makes every link as slow as the slowest ray of the whole
// frame plus a kernel boundary.  Here a pixel is owned by one group of G lanes of a persistent wave from its first ray to
// its last: the group traces the ray together (group_node_step / group_leaf_step), then advances the pixel's recursion to
// its next ray itself (renderRay, src/recursive.cpp:25-83, as a state machine) and goes on, with no synchronisation
// beyond its own wave.  A group whose pixel has ended takes the next pixel from a shared counter.  Everything that is not
// sequential leaves the kernel as independent work, exactly as in the round-based stage: one shade task per hit (its
// shadow rays and shading model run later) and the rays at the recursion limit (leaf tasks).
//
// Per group, in LDS: the traversal stack and leaf queue (GroupLayout), the pixel's context (8 words) and one frame per
// pending glossy loop (kFrameDwords words per rayDepth).  All lanes of a group run the recursion's arithmetic
// redundantly on the same values and write the same LDS words; only lane 0 of the group stores to global memory.
#pragma once

struct PathKParams {
    SceneView T;              // the tree the rays walk: the query view or the reference view
    const uint32_t* list;     // state slot of item i (null: i itself)
    const uint32_t* countPtr; // number of items
    uint32_t* fetch;          // zeroed work counter of the launch
    uint32_t* suspList;       // query-tree launches: state slots of the pixels whose next ray the reference's tree must answer
    uint32_t* suspCount;
    int reserve;              // stack room a group keeps free (tree depth + 2, see group_node_step)
    uint32_t shadeChunk, leafChunk; // queue slots a wave reserves at a time (leafChunk: a multiple of numGlossySamples)
    unsigned long long* dbg;  // profiling builds: end time of every wave
    uint32_t scatter;         // one pixel per lane: spread the lanes of a wave over the list
    // One pixel per lane, in slices: a pixel that has traced linkBudget rays in this launch, and every pixel of a wave that
    // can no longer fill minLanes lanes, is set aside with its next ray (state record + contList) for the next launch, which
    // packs what is left of the frame into full waves again.  contList null: every pixel is walked to its end.
    uint32_t* contList;
    uint32_t* contCount;
    uint32_t linkBudget, minLanes;
    uint32_t advanceAt, advanceEvery; // one pixel per lane: the recursion is advanced once this many lanes wait for it, or after this many step batches
};

constexpr int kCtxDwords = 8; // slot, sampler state, px | py << 16, weight[3], depth | loopDepth+1 << 8 | sampleIdx << 16, unused
constexpr int kGroupStateDwords = kCtxDwords + RT_MAX_RAY_DEPTH * kFrameDwords; // 128
enum PixelState : int { PX_NONE = 0, PX_BEGIN, PX_TRACE, PX_HIT };

// A run of queue slots a wave has reserved for itself: one trip to the queue's counter per chunk instead of one per task
// (additions to one address are served one at a time; see kStatCopies).
struct QueueChunk {
    uint32_t next, end;
};
// `per` consecutive slots for every lane with `want` (the chunk size is a multiple of `per`, so a chunk is used up
// exactly).  0xffffffff: the queue is full (or the lane wanted nothing).
__device__ __forceinline__ uint32_t chunk_alloc(uint32_t* counter, const uint32_t cap, const uint32_t chunkSize, QueueChunk& c, const bool want,
                                                const uint32_t per, const int lane)
{
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0xffffffffu;
    const uint32_t nReq = (uint32_t)__popcll(mask), r = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t availReq = (c.end - c.next) / per;
    uint32_t slot;
    if (nReq > availReq) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(counter, chunkSize);
        base = __shfl(base, 0);
        slot = r < availReq ? c.next + r * per : base + (r - availReq) * per;
        c.next = base + (nReq - availReq) * per, c.end = base + chunkSize;
    } else {
        slot = c.next + r * per;
        c.next += nReq * per;
    }
    return (want && slot + per <= cap && slot + per > slot) ? slot : 0xffffffffu;
}

// Pixel ray k of pixel (px, py): generatePixelRays (src/render.cpp:56-123) + Trackball::generateRay
__device__ __forceinline__ void pixel_ray(const WaveParams& P, const uint32_t pxy, const uint32_t k, vec3& o, vec3& d)
{
    const rt_features& F = P.F;
    const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
    float ndcX, ndcY;
    if (F.numPixelSamples > 1) { // offsets are drawn for every pixel ray before any is traced
        uint32_t s2 = static_cast<uint32_t>(P.H * (int)px + (int)py); // src/render.cpp:38
        for (uint32_t i = 0; i < 2 * k; i++) (void)sampler_next(s2);
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
    camera_ray(P.cams[0], ndcX, ndcY, o, d);
}

// Advances the pixel of every group with `pending` from "its ray (ro, rd) ended at tt on ttri" to its next ray
// (returned in ro, rd; newState = PX_BEGIN) or to the pixel's end (PX_NONE), appending the shade task and the leaf tasks
// met on the way.  Wave-convergent: every lane of the wave must call it.
template <int G, int CS>
__device__ __forceinline__ int path_advance(const WaveParams& P, const PathKParams& K, const bool pending, uint32_t* const ctx, float* const fr,
                                            vec3& ro, vec3& rd, const float tt, const uint32_t ttri, const int j, const int lane,
                                            QueueChunk& shadeC, QueueChunk& leafC, uint32_t& cPrimary, uint32_t& cSecondary, uint32_t& cDrop)
{
    const SceneView& S = P.S;
    const rt_features& F = P.F;
    const bool deferLeaves = (P.lightDraws == 0); // with point lights only, a ray at the recursion limit draws nothing
    const uint32_t N = F.numGlossySamples;
    const uint32_t nPixelRays = pixel_ray_count(F);
    const bool leader = (j == 0);
    // the pixel's context
    uint32_t smp = 0, pxy = 0, sampleIdx = 0;
    int depth = 0, loopDepth = -1;
    vec3 curW = splat(0.0f);
    if (pending) {
        smp = ctx[1 * CS], pxy = ctx[2 * CS];
        curW = make3(__uint_as_float(ctx[3 * CS]), __uint_as_float(ctx[4 * CS]), __uint_as_float(ctx[5 * CS]));
        const uint32_t pk = ctx[6 * CS];
        depth = (int)(pk & 0xffu), loopDepth = (int)((pk >> 8) & 0xffu) - 1, sampleIdx = pk >> 16;
    }
    const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
    const uint32_t pixel = (uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px; // Screen::indexAt (src/screen.cpp:119-122)
    int st = pending ? PS_HIT : PS_TRACE;
    do {
        bool wantShade = false, wantLeaves = false, openLoop = false;
        // the shade task of this pass's hit; the reflection of a loop that opens (frame or leaf tasks)
        uint32_t sMat = 0, sSmp = 0;
        vec3 sW = splat(0.0f), sP = splat(0.0f), sD = splat(0.0f), sN = splat(0.0f);
        vec3 lOrg = splat(0.0f), lDir = splat(0.0f), lKs = splat(0.0f);
        float lRadius = 0.0f;
        while (st > PS_TRACE && !wantShade) {
            switch (st) {
            case PS_HIT: {
                // renderRay after the intersection (src/recursive.cpp:29-36, 62-80)
                if (ttri == kNoHit) {
                    st = PS_POP;
                    break;
                }
                Hit h;
                fill_hit(S, F, ttri, ro, rd, tt, h);
                if (leader && depth == P.startDepth && sampleIdx == 0 && loopDepth < 0) {
                    if (P.primId) P.primId[py * P.W + px] = h.primId;
                    if (P.primT) P.primT[py * P.W + px] = tt;
                }
                const vec3 p = ro + tt * rd;
                sMat = h.mesh | (h.isSphere ? 0x80000000u : 0u);
                sSmp = smp;
                sW = curW, sP = p, sD = rd, sN = h.normal;
                wantShade = true; // the hit's direct light: deferred to the shade kernels
                for (uint32_t k = 0; k < P.lightDraws; k++) (void)sampler_next(smp); // the light loop's own draws
                st = PS_POP;
                if (depth < RT_MAX_RAY_DEPTH && F.enableReflections) {
                    const Mat m = load_material(S, h.mesh, h.isSphere);
                    const bool reflective = (m.ks.x != 0.0f) || (m.ks.y != 0.0f) || (m.ks.z != 0.0f);
                    if (reflective) {
                        // generateReflectionRay (src/recursive.cpp:91-103)
                        vec3 rorg = p;
                        const vec3 rdir = rd - 2.f * dot(rd, h.normal) * h.normal;
                        rorg = rorg + 0.00001f * rdir;
                        if (!F.enableGlossyReflection) { // renderRaySpecularComponent: weight * ks
                            ro = rorg, rd = rdir;
                            curW = curW * m.ks;
                            depth++;
                            cSecondary += leader ? 1u : 0u;
                            st = PS_TRACE;
                        } else if (N > 0) { // renderRayGlossyComponent (src/extra.cpp:464-508)
                            lOrg = rorg, lDir = rdir, lKs = m.ks;
                            lRadius = 0.5f / m.shininess;
                            // children at the recursion limit spawn nothing and (with point lights) draw nothing: all N leave
                            // the chain at once as leaf tasks; otherwise the loop is walked one child at a time
                            if (deferLeaves && depth + 1 >= RT_MAX_RAY_DEPTH) wantLeaves = true;
                            else openLoop = true;
                        }
                    }
                }
                break;
            }
            case PS_GLOSSY: {
                float* f = fr + loopDepth * kFrameDwords;
                const vec3 fW = make3(f[0], f[1], f[2]), rorg = make3(f[3], f[4], f[5]), rdir = make3(f[6], f[7], f[8]);
                const vec3 nrm = make3(f[9], f[10], f[11]), ks = make3(f[12], f[13], f[14]);
                const float radius = f[15];
                uint32_t i = __float_as_uint(f[16]);
                const int enclosing = (int)__float_as_uint(f[17]) - 1;
                vec3 w, u, v;
                ortho_basis(rdir, w, u, v);
                bool launched = false;
                while (i < N) {
                    vec3 pd;
                    const bool ok = glossy_sample(smp, radius, w, u, v, nrm, pd); // two draws, consumed even if rejected
                    i++;
                    if (!ok) continue;
                    // accumulatedColor += ks * L; hitColor += accumulatedColor / N  ->  weight * ks / N
                    curW = fW * ks / float(N);
                    ro = rorg + 0.0001f * pd, rd = pd;
                    depth = loopDepth + 1;
                    cSecondary += leader ? 1u : 0u;
                    launched = true;
                    st = PS_TRACE;
                    break;
                }
                f[16] = __uint_as_float(i);
                if (!launched) { // loop finished: resume the enclosing loop
                    loopDepth = enclosing;
                    st = PS_POP;
                }
                break;
            }
            case PS_POP: {
                // the ray in flight is finished: continue the innermost pending glossy loop, else the pixel
                if (loopDepth >= 0) {
                    st = PS_GLOSSY;
                    break;
                }
                if (sampleIdx + 1 < nPixelRays) {
                    sampleIdx++;
                    pixel_ray(P, pxy, sampleIdx, ro, rd);
                    curW = splat(1.0f / static_cast<float>(nPixelRays)); // renderRays: L / rays.size()
                    depth = P.startDepth;
                    cPrimary += leader ? 1u : 0u;
                    st = PS_TRACE;
                } else {
                    st = PS_DONE;
                }
                break;
            }
            default: st = PS_DONE; break;
            }
        }
        // ---- queue slots for what this pass produced ----
        const uint32_t shadeSlot = chunk_alloc(P.qCount + 0, P.shadeCap, K.shadeChunk, shadeC, wantShade && leader, 1u, lane);
        const uint32_t leafSlot = chunk_alloc(P.qCount + 1, P.leafCap, K.leafChunk, leafC, wantLeaves && leader, N, lane);
        if (wantShade) {
            if (leader) {
                if (shadeSlot != 0xffffffffu) store_shade_task(P.shadeQ + shadeSlot, pixel, sMat, sSmp, ttri, sW, sP, sD, sN);
                else cDrop++;
            }
            if (openLoop) { // push the loop's frame; its first sample is drawn in the next pass
                float* f = fr + depth * kFrameDwords;
                f[0] = sW.x, f[1] = sW.y, f[2] = sW.z;
                f[3] = lOrg.x, f[4] = lOrg.y, f[5] = lOrg.z;
                f[6] = lDir.x, f[7] = lDir.y, f[8] = lDir.z;
                f[9] = sN.x, f[10] = sN.y, f[11] = sN.z;
                f[12] = lKs.x, f[13] = lKs.y, f[14] = lKs.z;
                f[15] = lRadius;
                f[16] = __uint_as_float(0u), f[17] = __uint_as_float((uint32_t)(loopDepth + 1));
                loopDepth = depth;
                st = PS_GLOSSY;
            }
            if (wantLeaves) { // the whole loop at once: sample s to slot s, a hole where it is rejected
                vec3 w, u, v;
                ortho_basis(lDir, w, u, v);
                const vec3 lw = sW * lKs / float(N);
                if (leader && leafSlot == 0xffffffffu) cDrop++;
                for (uint32_t k = 0; k < N; k++) {
                    vec3 pd;
                    const bool ok = glossy_sample(smp, lRadius, w, u, v, sN, pd);
                    cSecondary += (ok && leader) ? 1u : 0u;
                    if (!leader || leafSlot == 0xffffffffu) continue;
                    if (ok) store_leaf_task(P.leafQ + leafSlot + k, pixel, lOrg + 0.0001f * pd, pd, FLT_MAX, lw);
                    else store_leaf_task(P.leafQ + leafSlot + k, kHolePixel, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, splat(0.0f));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    } while (__ballot(st > PS_TRACE) != 0ull);
    if (pending && st == PS_TRACE) { // every lane of the group writes the same words
        ctx[1 * CS] = smp;
        ctx[3 * CS] = __float_as_uint(curW.x), ctx[4 * CS] = __float_as_uint(curW.y), ctx[5 * CS] = __float_as_uint(curW.z);
        ctx[6 * CS] = (uint32_t)depth | ((uint32_t)(loopDepth + 1) << 8) | (sampleIdx << 16);
    }
    return (pending && st == PS_TRACE) ? PX_BEGIN : PX_NONE;
}

template <int G, bool QUERY>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void path_kernel(const WaveParams P, const PathKParams K)
{
    using Lay = GroupLayout<G, !QUERY>;
    constexpr int kGroups = kBlock / G;
    __shared__ uint32_t s_q[Lay::kWords];
    __shared__ uint32_t s_px[kGroups * kGroupStateDwords];
    const SceneView& TS = K.T;
    const int lane = threadIdx.x & 63;
    const int j = lane & (G - 1), leaderLane = lane - j, groupShift = leaderLane;
    uint32_t* const q = s_q + (threadIdx.x / G) * Lay::kGroupWords;
    uint32_t* const ctx = s_px + (threadIdx.x / G) * kGroupStateDwords;
    float* const fr = reinterpret_cast<float*>(ctx + kCtxDwords);
    const uint32_t total = *K.countPtr;
    TraceCounters tc { 0, 0, 0 };
    uint32_t cPrimary = 0, cSecondary = 0, cDrop = 0;
    // The first pixel of every group is fixed by the group's position in the grid; later pixels come from the shared
    // counter in chunks of kGroupChunk per wave (see trace_kernel).
    const uint32_t gridGroups = gridDim.x * kGroups;
    const uint32_t myGroup = blockIdx.x * kGroups + threadIdx.x / G;
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) / G >= total) return; // nothing for this wave, now or later
    bool exhausted = total <= gridGroups, first = true;
    uint32_t chunkNext = 0, chunkEnd = 0;
    QueueChunk shadeC { 0, 0 }, leafC { 0, 0 };
    GroupLane T;
    T.R.o = T.R.d = T.R.r = splat(0.0f);
    T.R.fast = T.R.approx = false;
    T.R.nor = splat(0.0f), T.R.eps = 0.0f;
    T.bestT = T.cullT = 0.0f, T.bestKey = 0, T.bestTri = kNoHit, T.nq = T.lq = 0;
    T.tieT = -1.0f;
    int pstate = PX_NONE; // same on every lane of a group
#ifdef RT_PATH_PROFILE
    long long tAdv = 0, tSteps = 0, tFetch = 0, tBegin = 0, t0 = clock64(), tA;
    uint32_t nIter = 0, nAdv = 0, nStepIt = 0;
    uint32_t dRaysPix = 0, dStepsRay = 0, dMaxRaysPix = 0, dMaxStepsRay = 0, dRays = 0, dSteps = 0;
#define RT_TICK() tA = clock64()
#define RT_TOCK(acc) acc += clock64() - tA
#else
#define RT_TICK()
#define RT_TOCK(acc)
#endif
    for (;;) {
#ifdef RT_PATH_PROFILE
        nIter++;
#endif
        RT_TICK();
        // ---- groups without a pixel take the next one ----
        const unsigned long long idle = __ballot(pstate == PX_NONE && j == 0);
        if (idle != 0ull && (first || !exhausted || chunkNext < chunkEnd)) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = myGroup;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(K.fetch, kGroupChunk);
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
            if (pstate == PX_NONE && next < total) {
                const uint32_t slot = K.list ? K.list[next] : next;
                const float* s = P.states + (size_t)slot * kStateDwords;
                const float4* hq = reinterpret_cast<const float4*>(s);
                const float4 a = hq[0], b = hq[1], c = hq[2], d = hq[3];
                ctx[0] = slot, ctx[1] = __float_as_uint(a.y), ctx[2] = __float_as_uint(a.w);
                ctx[3] = __float_as_uint(c.z), ctx[4] = __float_as_uint(c.w), ctx[5] = __float_as_uint(d.x);
                const uint32_t pk = __float_as_uint(d.y);
                ctx[6] = pk;
                T.R.o = make3(b.x, b.y, b.z), T.R.d = make3(b.w, c.x, c.y); // the ray to trace next
                const int frameWords = (int)((pk >> 8) & 0xffu) * kFrameDwords; // frames of the pending loops (a pixel taken over)
                for (int w = j; w < frameWords; w += G) fr[w] = s[16 + w];
                pstate = PX_BEGIN;
#ifdef RT_PATH_PROFILE
                dRaysPix = 0;
#endif
            }
        }
        RT_TOCK(tFetch);
        // ---- pixels whose ray has ended: on to their next ray ----
        if (__ballot(pstate == PX_HIT) != 0ull) {
            RT_TICK();
#ifdef RT_PATH_PROFILE
            nAdv++;
#endif
            const int ns = path_advance<G, 1>(P, K, pstate == PX_HIT, ctx, fr, T.R.o, T.R.d, T.bestT, T.bestTri, j, lane, shadeC, leafC, cPrimary,
                                           cSecondary, cDrop);
            if (pstate == PX_HIT) pstate = ns;
            RT_TOCK(tAdv);
        }
        RT_TICK();
        // ---- start the rays ----
        bool susp = false;
        if (pstate == PX_BEGIN) {
            const vec3 o = T.R.o, d = T.R.d;
            const bool has = group_begin<G, QUERY>(TS, T, q, j, o, d, FLT_MAX);
#ifdef RT_PATH_PROFILE
            dRaysPix++, dRays++, dStepsRay = 0;
            dMaxRaysPix = max(dMaxRaysPix, dRaysPix);
#endif
            if (!has) pstate = PX_HIT; // the ray misses the root box: T.bestTri = kNoHit
            else if (QUERY && !T.R.approx) susp = true; // the conservative tests need the approximate slabs
            else pstate = PX_TRACE;
        }
        RT_TOCK(tBegin);
        RT_TICK();
#pragma unroll 1
        for (int it = 0; it < 5; it++) {
            // it == 0 only settles the rays that could not start in the query tree
            if (it > 0) {
                const bool has = pstate == PX_TRACE;
#ifdef RT_PATH_PROFILE
                if (has) dStepsRay++, dSteps++, dMaxStepsRay = max(dMaxStepsRay, dStepsRay);
#endif
                const bool needLeaf = has && (T.lq >= G || (T.nq == 0 && T.lq > 0));
                const bool needNode = has && !needLeaf && T.nq > 0;
                if (__ballot(needNode) != 0ull)
                    group_node_step<true, false, G, QUERY>(TS, T, q, j, groupShift, K.reserve, has && T.nq > 0 && T.lq + 2 * G <= Lay::kLeafQ, tc);
                if (__ballot(has && (T.lq >= G || (T.nq == 0 && T.lq > 0))) != 0ull)
                    group_leaf_step<false, G, QUERY>(TS, T, q, j, has && T.lq > 0, tc);
                const bool done = has && T.nq == 0 && T.lq == 0;
                if (__ballot(done) != 0ull) {
                    bool proven = true;
                    if (QUERY) proven = group_finish_query<G>(TS, T, groupShift, done);
                    else group_finish<G>(TS, T, done);
                    if (done) {
                        if (proven) pstate = PX_HIT;
                        else susp = true;
                    }
                }
            }
            // a ray the query tree cannot answer with proof: the pixel goes on in the reference's tree (another launch)
            if (QUERY && __ballot(susp) != 0ull) {
                const uint32_t pos = wave_fetch(K.suspCount, susp && j == 0, lane);
                if (susp) {
                    float* s = P.states + (size_t)ctx[0] * kStateDwords;
                    const uint32_t pxy = ctx[2], pk = ctx[6];
                    if (j == 0) {
                        const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
                        float4* hq = reinterpret_cast<float4*>(s);
                        hq[0] = make_float4(__uint_as_float((uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px), __uint_as_float(ctx[1]),
                                            __uint_as_float(static_cast<uint32_t>(P.H * (int)px + (int)py)), __uint_as_float(pxy));
                        hq[1] = make_float4(T.R.o.x, T.R.o.y, T.R.o.z, T.R.d.x);
                        hq[2] = make_float4(T.R.d.y, T.R.d.z, __uint_as_float(ctx[3]), __uint_as_float(ctx[4]));
                        hq[3] = make_float4(__uint_as_float(ctx[5]), __uint_as_float(pk), __uint_as_float(1u), 0.0f);
                        K.suspList[pos] = ctx[0];
                    }
                    const int frameWords = (int)((pk >> 8) & 0xffu) * kFrameDwords;
                    for (int w = j; w < frameWords; w += G) s[16 + w] = fr[w];
                    pstate = PX_NONE;
                    susp = false;
                }
            }
            if (__ballot(pstate == PX_TRACE) == 0ull) break;
#ifdef RT_PATH_PROFILE
            nStepIt++;
#endif
        }
        RT_TOCK(tSteps);
        if (__ballot(pstate != PX_NONE) == 0ull && exhausted && chunkNext == chunkEnd) break;
    }
#ifdef RT_PATH_PROFILE
    if (j == 0) {
        atomicMax(P.stats + 24, (unsigned long long)dMaxRaysPix), atomicMax(P.stats + 25, (unsigned long long)dMaxStepsRay);
        atomicAdd(P.stats + 26, (unsigned long long)dRays), atomicAdd(P.stats + 27, (unsigned long long)dSteps);
    }
    if (lane == 0) {
        atomicAdd(P.stats + 28, (unsigned long long)(clock64() - t0)), atomicAdd(P.stats + 29, 1ull), atomicAdd(P.stats + 30, (unsigned long long)nStepIt);
    }
    if (lane == 0) { // slots 16..: the slowest wave's split of its time (cycles) and its iteration counts
        unsigned long long* dbg = P.stats + 16;
        const unsigned long long tot = (unsigned long long)(clock64() - t0);
        const unsigned long long prev = atomicMax(dbg + 0, tot);
        if (tot > prev) {
            dbg[1] = tFetch, dbg[2] = tAdv, dbg[3] = tBegin, dbg[4] = tSteps, dbg[5] = nIter, dbg[6] = nAdv, dbg[7] = nStepIt;
        }
    }
#endif
    // what is left of the wave's reserved queue slots: holes
    for (uint32_t k = shadeC.next + lane; k < shadeC.end && k < P.shadeCap; k += 64u)
        reinterpret_cast<uint32_t*>(P.shadeQ + k)[0] = kHolePixel;
    for (uint32_t k = leafC.next + lane; k < leafC.end && k < P.leafCap; k += 64u)
        store_leaf_task(P.leafQ + k, kHolePixel, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, splat(0.0f));
    flush_counters(P.stats, lane, cPrimary, 0, cSecondary, tc, 0, cDrop);
}

// The same stage with one pixel per lane: 64 pixels per wave instead of 64 / G, each ray walked by its own lane
// (trace_wave_step; with a query tree the 4-wide form).  A link of a pixel's chain then lasts as long as its own ray plus
// the wait for the wave's next pass over the recursion (path_advance), which serves every lane whose ray has ended at
// once.  Frames of pending glossy loops live in the pixel's state record in global memory.
#ifndef RT_LANE_PATH_WAVES
#define RT_LANE_PATH_WAVES 3
#endif
template <int STACK, bool QUERY>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_LANE_PATH_WAVES))) void lane_path_kernel(const WaveParams P, const PathKParams K)
{
    __shared__ uint32_t s_stack[2 * STACK * kBlock]; // per lane: STACK node references, then their entry distances
    __shared__ uint32_t s_ctx[kCtxDwords * kBlock];  // word w of thread t at [w * kBlock + t]
    uint32_t* const stack = s_stack + threadIdx.x;
    uint32_t* const ctx = s_ctx + threadIdx.x;
    const SceneView& TS = K.T;
    const int lane = threadIdx.x & 63;
    const uint32_t items = *K.countPtr;
    // Items follow the frame tile by tile and neighbouring pixels have chains of similar length: with `scatter` the lanes of a
    // wave take items a 64th of the list apart (work index i stands for item (i % 64) * rows + i / 64), so that every wave
    // holds the same mix of short and long chains.
    const uint32_t rows = (items + 63u) / 64u;
    const uint32_t total = K.scatter ? rows * 64u : items; // work indices
    TraceCounters tc { 0, 0, 0 };
    uint32_t cPrimary = 0, cSecondary = 0, cDrop = 0;
    const uint32_t gridLanes = gridDim.x * kBlock;
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) >= total) return; // nothing for this wave, now or later
    bool exhausted = total <= gridLanes, first = true;
    uint32_t chunkNext = 0, chunkEnd = 0, sinceAdvance = 0, links = 0;
    QueueChunk shadeC { 0, 0 }, leafC { 0, 0 };
    TraceLane T;
    trace_lane_clear(T);
    int pstate = PX_NONE;
#ifdef RT_PATH_PROFILE
    long long tAdv = 0, tSteps = 0, tFetch = 0, tBegin = 0, t0 = clock64(), tA;
    if (lane == 0 && K.dbg) K.dbg[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = wall_clock64();
    uint32_t nIter = 0, nAdv = 0, nStepIt = 0, nAdvLanes = 0, nStepLanes = 0;
    uint32_t dRaysPix = 0, dStepsRay = 0, dMaxRaysPix = 0, dMaxStepsRay = 0, dRays = 0, dSteps = 0;
#endif
    for (;;) {
#ifdef RT_PATH_PROFILE
        nIter++;
#endif
        RT_TICK();
        // ---- lanes without a pixel take the next one ----
        const unsigned long long idle = __ballot(pstate == PX_NONE);
        if (idle != 0ull && (first || !exhausted || chunkNext < chunkEnd)) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = blockIdx.x * kBlock + threadIdx.x;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(K.fetch, kLaneChunk);
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
            if (K.scatter && next < total) next = (next & 63u) * rows + (next >> 6);
            if (pstate == PX_NONE && next < items) {
                const uint32_t slot = K.list ? K.list[next] : next;
                const float4* hq = reinterpret_cast<const float4*>(P.states + (size_t)slot * kStateDwords);
                const float4 a = hq[0], b = hq[1], c = hq[2], d = hq[3];
                ctx[0] = slot, ctx[1 * kBlock] = __float_as_uint(a.y), ctx[2 * kBlock] = __float_as_uint(a.w);
                ctx[3 * kBlock] = __float_as_uint(c.z), ctx[4 * kBlock] = __float_as_uint(c.w), ctx[5 * kBlock] = __float_as_uint(d.x);
                ctx[6 * kBlock] = __float_as_uint(d.y);
                T.R.o = make3(b.x, b.y, b.z), T.R.d = make3(b.w, c.x, c.y); // the ray to trace next
                pstate = PX_BEGIN;
                links = 0;
#ifdef RT_PATH_PROFILE
                dRaysPix = 0;
#endif
            }
        }
        RT_TOCK(tFetch);
        // ---- pixels whose ray has ended: on to their next ray, once enough of them wait or nothing else is left to do ----
        {
            const uint32_t nHit = (uint32_t)__popcll(__ballot(pstate == PX_HIT));
            const bool tracing = __ballot(pstate == PX_TRACE || pstate == PX_BEGIN) != 0ull;
            if (nHit != 0u && (nHit >= K.advanceAt || !tracing || sinceAdvance >= K.advanceEvery)) {
                RT_TICK();
#ifdef RT_PATH_PROFILE
                nAdv++, nAdvLanes += nHit;
#endif
                float* const fr = P.states + (size_t)ctx[0] * kStateDwords + 16;
                const int ns = path_advance<1, kBlock>(P, K, pstate == PX_HIT, ctx, fr, T.R.o, T.R.d, T.t, T.tri, 0, lane, shadeC, leafC, cPrimary,
                                                       cSecondary, cDrop);
#ifdef RT_PATH_PROFILE
                if (pstate == PX_HIT && ns == PX_NONE) atomicAdd(P.stats + 32 * 1 + 9 + min(dRaysPix / 4u, 22u), 1ull); // copy 1, slots 9..31
#endif
                if (pstate == PX_HIT) pstate = ns;
                sinceAdvance = 0;
                RT_TOCK(tAdv);
            }
        }
        RT_TICK();
        // ---- start the rays ----
        bool susp = false; // set aside for the launch that walks the reference's tree
        bool cont = false; // set aside for the next slice
        if (K.contList) {
            const bool thin = exhausted && chunkNext == chunkEnd && (uint32_t)__popcll(__ballot(pstate != PX_NONE)) < K.minLanes;
            cont = pstate == PX_BEGIN && links != 0u && (thin || links >= K.linkBudget); // at least one ray per launch
        }
        if (pstate == PX_BEGIN && !cont) {
            const vec3 o = T.R.o, d = T.R.d;
            const bool has = trace_begin<false>(TS, P.F.enableAccelStructure != 0, T, o, d, FLT_MAX, tc);
            links++;
#ifdef RT_PATH_PROFILE
            dRaysPix++, dRays++, dStepsRay = 0;
            dMaxRaysPix = max(dMaxRaysPix, dRaysPix);
#endif
            if (!has) pstate = PX_HIT; // the ray misses the root box (or every triangle was tested: no tree)
            else if (QUERY && !T.R.approx) susp = true; // the conservative tests need the approximate slabs
            else pstate = PX_TRACE;
        }
        RT_TOCK(tBegin);
        RT_TICK();
#pragma unroll 1
        for (int it = 0; it <= kStepsPerCheck; it++) {
            // it == 0 only settles the rays that could not start in the query tree
            if (it > 0) {
#ifdef RT_PATH_PROFILE
                nStepIt++, nStepLanes += (uint32_t)__popcll(__ballot(pstate == PX_TRACE));
                if (pstate == PX_TRACE) dStepsRay++, dSteps++, dMaxStepsRay = max(dMaxStepsRay, dStepsRay);
#endif
                const bool ended = trace_wave_step<true, false, STACK, QUERY>(TS, pstate == PX_TRACE, T, stack, tc);
                if (ended) {
                    bool proven = true;
                    if (QUERY) {
                        proven = query_result_proven(TS, T.R, T.t, T.tri, T.tieT, T.lost);
                        for (uint32_t s = 0; s < TS.nSpheres; s++) // src/bvh.cpp:366-367
                            if (sphere_test(TS.spheres[s], T.R.o, T.R.d, T.t)) T.tri = kSphereBase + s;
                    }
                    if (proven) pstate = PX_HIT;
                    else susp = true;
                }
            }
            // a ray the query tree cannot answer with proof: the pixel goes on in the reference's tree (another launch);
            // a pixel set aside for the next slice: the same record, another list
            if (__ballot(susp || cont) != 0ull) {
                const uint32_t posS = QUERY ? wave_fetch(K.suspCount, susp, lane) : 0u;
                const uint32_t posC = wave_fetch(K.contCount, cont, lane);
                if (susp || cont) {
                    const uint32_t pxy = ctx[2 * kBlock];
                    const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
                    float4* hq = reinterpret_cast<float4*>(P.states + (size_t)ctx[0] * kStateDwords);
                    hq[0] = make_float4(__uint_as_float((uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px), __uint_as_float(ctx[1 * kBlock]),
                                        __uint_as_float(static_cast<uint32_t>(P.H * (int)px + (int)py)), __uint_as_float(pxy));
                    hq[1] = make_float4(T.R.o.x, T.R.o.y, T.R.o.z, T.R.d.x);
                    hq[2] = make_float4(T.R.d.y, T.R.d.z, __uint_as_float(ctx[3 * kBlock]), __uint_as_float(ctx[4 * kBlock]));
                    hq[3] = make_float4(__uint_as_float(ctx[5 * kBlock]), __uint_as_float(ctx[6 * kBlock]), __uint_as_float(1u), 0.0f);
                    if (susp) K.suspList[posS] = ctx[0];
                    else K.contList[posC] = ctx[0];
                    pstate = PX_NONE;
                    susp = cont = false;
                }
            }
            if (__ballot(pstate == PX_TRACE) == 0ull) break;
        }
        RT_TOCK(tSteps);
        sinceAdvance++;
        if (__ballot(pstate != PX_NONE) == 0ull && exhausted && chunkNext == chunkEnd) break;
    }
#ifdef RT_PATH_PROFILE
    if (lane == 0 && K.dbg) K.dbg[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = wall_clock64();
    atomicMax(P.stats + 24, (unsigned long long)dMaxRaysPix), atomicMax(P.stats + 25, (unsigned long long)dMaxStepsRay);
    atomicAdd(P.stats + 26, (unsigned long long)dRays), atomicAdd(P.stats + 27, (unsigned long long)dSteps);
    if (lane == 0) {
        atomicAdd(P.stats + 28, (unsigned long long)(clock64() - t0)), atomicAdd(P.stats + 29, 1ull), atomicAdd(P.stats + 30, (unsigned long long)nStepIt);
        atomicAdd(P.stats + 31, (unsigned long long)nStepLanes);
        atomicAdd(P.stats + 15, (unsigned long long)nAdv), atomicAdd(P.stats + 14, (unsigned long long)nAdvLanes);
        atomicAdd(P.stats + 13, (unsigned long long)tAdv), atomicAdd(P.stats + 12, (unsigned long long)tSteps);
        unsigned long long* dbg = P.stats + 16;
        const unsigned long long tot = (unsigned long long)(clock64() - t0);
        const unsigned long long prev = atomicMax(dbg + 0, tot);
        if (tot > prev) {
            dbg[1] = tFetch, dbg[2] = tAdv, dbg[3] = tBegin, dbg[4] = tSteps, dbg[5] = nIter, dbg[6] = nAdv, dbg[7] = nStepIt;
        }
    }
#endif
    // what is left of the wave's reserved queue slots: holes
    for (uint32_t k = shadeC.next + lane; k < shadeC.end && k < P.shadeCap; k += 64u)
        reinterpret_cast<uint32_t*>(P.shadeQ + k)[0] = kHolePixel;
    for (uint32_t k = leafC.next + lane; k < leafC.end && k < P.leafCap; k += 64u)
        store_leaf_task(P.leafQ + k, kHolePixel, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, splat(0.0f));
    flush_counters(P.stats, lane, cPrimary, 0, cSecondary, tc, 0, cDrop);
}

// ---------------------------------------------------------------------------------------------
// Four lanes per ray on the query tree's 4-wide form: lane k of a quad tests child k of the node (one box per lane
// instead of four, no sorting network: a lane's rank among the quad's entry distances comes from three cross-lane reads)
// and triangle k of the leaf (plane and edge record fetched together: one round trip per leaf).  A step is one memory
// round trip and a few dozen instructions, which is what bounds a chain of dependent rays.  Conservative box tests and the
// proof of equality with the reference's walk are those of the other query-tree kernels (node4_test, query_result_proven).
// ---------------------------------------------------------------------------------------------
constexpr int kQuadStack = 24;                                  // pending nodes per quad (a ray that needs more is traced in the reference's tree)
constexpr int kQuadWords = 2 * kQuadStack + kGroupStateDwords;  // stack references, their entry distances, pixel context, frames
struct QuadLane {
    RaySlabs R;
    float bestT, tieT;
    uint32_t bestTri, cur, lost;
    int sp;
};

template <int CTRL>
__device__ __forceinline__ uint32_t quad_read(const uint32_t v) // the value of another lane of the quad (CTRL: quad_perm)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ float quad_min(const float v)
{
    float m = fminf(v, __uint_as_float(quad_read<0xB1>(__float_as_uint(v)))); // lanes k ^ 1
    return fminf(m, __uint_as_float(quad_read<0x4E>(__float_as_uint(m))));    // lanes k ^ 2
}

// Next pending node of the quad into T.cur (kTraversalDone when none is left); entries beyond the closest hit are dropped.
__device__ __forceinline__ void quad_pop(QuadLane& T, const uint32_t* const stack)
{
    for (;;) {
        if (T.sp == 0) {
            T.cur = kTraversalDone;
            return;
        }
        T.sp--;
        if (T.sp >= kQuadStack) continue; // entries that found no room were dropped (T.lost)
        if (__uint_as_float(stack[kQuadStack + T.sp]) > T.bestT) continue;
        T.cur = stack[T.sp];
        return;
    }
}

// One step of every quad with `active`: the four children of an inner node, then - for quads that stand at a leaf - its
// triangles.  k: lane within the quad, quadShift: first lane of the quad within the wave.
__device__ __forceinline__ void quad_step(const SceneView& S, QuadLane& T, uint32_t* const stack, const int k, const int quadShift, const bool active)
{
    if (active && !(T.cur & kLeafBit)) {
        const float4* cq = reinterpret_cast<const float4*>(S.quadNodes + T.cur) + 2 * k;
        const float4 ca = cq[0], cb = cq[1];
        const float lx = ca.x, ly = ca.y, lz = ca.z, ux = ca.w, uy = cb.x, uz = cb.y;
        const uint32_t ref = __float_as_uint(cb.z);
        const float E = T.R.eps, E2 = E + E;
        const float ax = fmaf(lx, T.R.r.x, T.R.nor.x), bx = fmaf(ux, T.R.r.x, T.R.nor.x);
        const float ay = fmaf(ly, T.R.r.y, T.R.nor.y), by = fmaf(uy, T.R.r.y, T.R.nor.y);
        const float az = fmaf(lz, T.R.r.z, T.R.nor.z), bz = fmaf(uz, T.R.r.z, T.R.nor.z);
        const float tin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float tex = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const bool hit = ref != kEmptyRef && !(tin > tex + E2) && !(tex < -E) && !(tin - E > T.bestT);
        const uint32_t key = hit ? ((__float_as_uint(fmaxf(tin, 0.0f)) & ~3u) | (uint32_t)k) : kMissKey;
        const uint32_t k1 = quad_read<0xB1>(key), k2 = quad_read<0x4E>(key), k3 = quad_read<0x1B>(key);
        const int rank = (k1 < key ? 1 : 0) + (k2 < key ? 1 : 0) + (k3 < key ? 1 : 0); // keys differ in their low bits
        const int nHit = (key != kMissKey ? 1 : 0) + (k1 != kMissKey ? 1 : 0) + (k2 != kMissKey ? 1 : 0) + (k3 != kMissKey ? 1 : 0);
        // the children entered go onto the stack farthest first; the nearest is then taken from its top
        if (hit) {
            const int pos = T.sp + (nHit - 1 - rank);
            if (pos < kQuadStack) {
                stack[pos] = ref;
                stack[kQuadStack + pos] = __float_as_uint(__uint_as_float(key & ~3u) - E);
            } else {
                T.lost = 1;
            }
        }
        if (__builtin_amdgcn_mov_dpp((int)T.lost, 0xB1, 0xf, 0xf, true) | __builtin_amdgcn_mov_dpp((int)T.lost, 0x4E, 0xf, 0xf, true)
            | __builtin_amdgcn_mov_dpp((int)T.lost, 0x1B, 0xf, 0xf, true))
            T.lost = 1;
        T.sp += nHit;
        __builtin_amdgcn_wave_barrier();
        quad_pop(T, stack);
    }
    if (active && T.cur != kTraversalDone && (T.cur & kLeafBit)) {
        const float4* lf = reinterpret_cast<const float4*>(S.quadLeaves + (T.cur & kRefOffsetMask));
        const float4 pl = lf[k];
        EdgeRec E;
        E.e0 = lf[4 + k], E.e1 = lf[8 + k], E.e2 = lf[12 + k], E.e3 = lf[16 + k], E.e4 = lf[20 + k];
        const vec3 n = make3(pl.x, pl.y, pl.z);
        const float tt = (pl.w - dot(T.R.o, n)) / dot(T.R.d, n); // an unused slot: 0 / 0
        const bool ok = tt >= 0.0f && T.bestT >= tt && point_in_triangle(E, T.R.d * tt + T.R.o); // equal distance accepted
        const float mine = ok ? tt : FLT_MAX;
        const float tmin = quad_min(mine);
        const uint32_t holders = (uint32_t)(__ballot(ok && mine == tmin) >> quadShift) & 0xfu;
        const uint32_t firstTri = quad_read<0x00>(__float_as_uint(E.e4.z)); // slot 0 is always in use
        if (holders != 0u) {
            // two triangles at the closest distance: the reference breaks the tie by the order of its own walk
            if (__popc(holders) > 1 || (tmin == T.bestT && T.bestTri != kNoHit)) T.tieT = tmin;
            T.bestT = tmin;
            T.bestTri = firstTri + (uint32_t)(__ffs(holders) - 1); // a leaf's triangles follow one another
        }
        quad_pop(T, stack);
    }
}

#ifndef RT_QUAD_WAVES
#define RT_QUAD_WAVES 4
#endif
template <bool FRAMES_IN_LDS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_QUAD_WAVES))) void quad_path_kernel(const WaveParams P, const PathKParams K)
{
    constexpr int kQuads = kBlock / 4;
    constexpr int kWords = FRAMES_IN_LDS ? kQuadWords : 2 * kQuadStack + kCtxDwords;
    __shared__ uint32_t s_q[kQuads * kWords];
    const SceneView& TS = K.T;
    const int lane = threadIdx.x & 63;
    const int k = lane & 3, leaderLane = lane - k, quadShift = leaderLane;
    uint32_t* const stack = s_q + (threadIdx.x / 4) * kWords;
    uint32_t* const ctx = stack + 2 * kQuadStack;
    float* const fr = reinterpret_cast<float*>(ctx + kCtxDwords);
    const uint32_t total = *K.countPtr;
    TraceCounters tc { 0, 0, 0 };
    uint32_t cPrimary = 0, cSecondary = 0, cDrop = 0;
    const uint32_t gridQuads = gridDim.x * kQuads;
    const uint32_t myQuad = blockIdx.x * kQuads + threadIdx.x / 4;
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) / 4 >= total) return; // nothing for this wave, now or later
    bool exhausted = total <= gridQuads, first = true;
    uint32_t chunkNext = 0, chunkEnd = 0, sinceAdvance = 0;
    QueueChunk shadeC { 0, 0 }, leafC { 0, 0 };
    QuadLane T;
    T.R.o = T.R.d = T.R.r = splat(0.0f);
    T.R.fast = T.R.approx = false;
    T.R.nor = splat(0.0f), T.R.eps = 0.0f;
    T.bestT = 0.0f, T.tieT = -1.0f, T.bestTri = kNoHit, T.cur = kTraversalDone, T.lost = 0, T.sp = 0;
    int pstate = PX_NONE; // same on every lane of a quad
#ifdef RT_PATH_PROFILE
    long long tAdv = 0, tSteps = 0, tFetch = 0, tBegin = 0, t0 = clock64(), tA;
    if (lane == 0 && K.dbg) K.dbg[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = wall_clock64();
    uint32_t nIter = 0, nAdv = 0, nStepIt = 0, nAdvLanes = 0, nStepLanes = 0;
    uint32_t dRaysPix = 0, dStepsRay = 0, dMaxRaysPix = 0, dMaxStepsRay = 0, dRays = 0, dSteps = 0;
#endif
    for (;;) {
#ifdef RT_PATH_PROFILE
        nIter++;
#endif
        RT_TICK();
        // ---- quads without a pixel take the next one ----
        const unsigned long long idle = __ballot(pstate == PX_NONE && k == 0);
        if (idle != 0ull && (first || !exhausted || chunkNext < chunkEnd)) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = myQuad;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(K.fetch, 16u);
                    base = __shfl(base, 0) + gridQuads;
                    exhausted = base + 16u >= total;
                    chunkNext = min(base, total), chunkEnd = min(base + 16u, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << leaderLane) - 1ull));
                const uint32_t avail = chunkEnd - chunkNext, n = (uint32_t)__popcll(idle);
                if (rank < avail) next = chunkNext + rank;
                chunkNext += min(n, avail);
            }
            first = false;
            if (pstate == PX_NONE && next < total) {
                const uint32_t slot = K.list ? K.list[next] : next;
                const float* s = P.states + (size_t)slot * kStateDwords;
                const float4* hq = reinterpret_cast<const float4*>(s);
                const float4 a = hq[0], b = hq[1], c = hq[2], d = hq[3];
                ctx[0] = slot, ctx[1] = __float_as_uint(a.y), ctx[2] = __float_as_uint(a.w);
                ctx[3] = __float_as_uint(c.z), ctx[4] = __float_as_uint(c.w), ctx[5] = __float_as_uint(d.x);
                const uint32_t pk = __float_as_uint(d.y);
                ctx[6] = pk;
                T.R.o = make3(b.x, b.y, b.z), T.R.d = make3(b.w, c.x, c.y); // the ray to trace next
                if (FRAMES_IN_LDS) {
                    const int frameWords = (int)((pk >> 8) & 0xffu) * kFrameDwords; // frames of the pending loops (a pixel taken over)
                    for (int w = k; w < frameWords; w += 4) fr[w] = s[16 + w];
                }
                pstate = PX_BEGIN;
#ifdef RT_PATH_PROFILE
                dRaysPix = 0;
#endif
            }
        }
        RT_TOCK(tFetch);
        // ---- pixels whose ray has ended: on to their next ray, once enough of them wait or nothing else is left to do ----
        {
            const uint32_t nHit = (uint32_t)__popcll(__ballot(pstate == PX_HIT && k == 0));
            const bool tracing = __ballot(pstate == PX_TRACE || pstate == PX_BEGIN) != 0ull;
            if (nHit != 0u && (nHit >= K.advanceAt || !tracing || sinceAdvance >= K.advanceEvery)) {
                RT_TICK();
#ifdef RT_PATH_PROFILE
                nAdv++, nAdvLanes += nHit;
#endif
                float* const frames = FRAMES_IN_LDS ? fr : P.states + (size_t)ctx[0] * kStateDwords + 16;
                const int ns = path_advance<4, 1>(P, K, pstate == PX_HIT, ctx, frames, T.R.o, T.R.d, T.bestT, T.bestTri, k, lane, shadeC, leafC,
                                                  cPrimary, cSecondary, cDrop);
#ifdef RT_PATH_PROFILE
                if (pstate == PX_HIT && ns == PX_NONE && k == 0) atomicAdd(P.stats + 32 * 1 + 9 + min(dRaysPix / 4u, 22u), 1ull);
#endif
                if (pstate == PX_HIT) pstate = ns;
                sinceAdvance = 0;
                RT_TOCK(tAdv);
            }
        }
        RT_TICK();
        // ---- start the rays ----
        bool susp = false;
        if (pstate == PX_BEGIN) {
            const vec3 o = T.R.o, d = T.R.d;
            T.R = make_slabs(TS, o, d);
            T.bestT = FLT_MAX, T.bestTri = kNoHit, T.tieT = -1.0f, T.lost = 0, T.sp = 0, T.cur = 0; // root of the 4-wide form: node 0
            float tr;
            bool outside;
#ifdef RT_PATH_PROFILE
            dRaysPix++, dRays++, dStepsRay = 0;
            dMaxRaysPix = max(dMaxRaysPix, dRaysPix);
#endif
            // root box against the ray's own t; a miss ends the query before the sphere loop (src/bvh.cpp:264-268)
            if (!box_test_any(TS.rootLo[0], TS.rootLo[1], TS.rootLo[2], TS.rootHi[0], TS.rootHi[1], TS.rootHi[2], T.R, FLT_MAX, tr, outside)) pstate = PX_HIT;
            else if (!T.R.approx) susp = true; // the conservative tests need the approximate slabs
            else pstate = PX_TRACE;
        }
        RT_TOCK(tBegin);
        RT_TICK();
#pragma unroll 1
        for (int it = 0; it <= 4; it++) {
            // it == 0 only settles the rays that could not start in the query tree
            if (it > 0) {
#ifdef RT_PATH_PROFILE
                nStepIt++, nStepLanes += (uint32_t)__popcll(__ballot(pstate == PX_TRACE));
                if (pstate == PX_TRACE) dStepsRay++, dSteps++, dMaxStepsRay = max(dMaxStepsRay, dStepsRay);
#endif
                quad_step(TS, T, stack, k, quadShift, pstate == PX_TRACE);
                if (pstate == PX_TRACE && T.cur == kTraversalDone) {
                    const bool proven = query_result_proven(TS, T.R, T.bestT, T.bestTri, T.tieT, T.lost);
                    for (uint32_t s = 0; s < TS.nSpheres; s++) // src/bvh.cpp:366-367
                        if (sphere_test(TS.spheres[s], T.R.o, T.R.d, T.bestT)) T.bestTri = kSphereBase + s;
                    if (proven) pstate = PX_HIT;
                    else susp = true;
                }
            }
            // a ray the query tree cannot answer with proof: the pixel goes on in the reference's tree (another launch)
            if (__ballot(susp) != 0ull) {
                const uint32_t pos = wave_fetch(K.suspCount, susp && k == 0, lane);
                if (susp) {
                    float* s = P.states + (size_t)ctx[0] * kStateDwords;
                    const uint32_t pxy = ctx[2], pk = ctx[6];
                    if (k == 0) {
                        const uint32_t px = pxy & 0xffffu, py = pxy >> 16;
                        float4* hq = reinterpret_cast<float4*>(s);
                        hq[0] = make_float4(__uint_as_float((uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px), __uint_as_float(ctx[1]),
                                            __uint_as_float(static_cast<uint32_t>(P.H * (int)px + (int)py)), __uint_as_float(pxy));
                        hq[1] = make_float4(T.R.o.x, T.R.o.y, T.R.o.z, T.R.d.x);
                        hq[2] = make_float4(T.R.d.y, T.R.d.z, __uint_as_float(ctx[3]), __uint_as_float(ctx[4]));
                        hq[3] = make_float4(__uint_as_float(ctx[5]), __uint_as_float(pk), __uint_as_float(1u), 0.0f);
                        K.suspList[pos] = ctx[0];
                    }
                    if (FRAMES_IN_LDS) {
                        const int frameWords = (int)((pk >> 8) & 0xffu) * kFrameDwords;
                        for (int w = k; w < frameWords; w += 4) s[16 + w] = fr[w];
                    }
                    pstate = PX_NONE;
                    susp = false;
                }
            }
            if (__ballot(pstate == PX_TRACE) == 0ull) break;
        }
        RT_TOCK(tSteps);
        sinceAdvance++;
        if (__ballot(pstate != PX_NONE) == 0ull && exhausted && chunkNext == chunkEnd) break;
    }
#ifdef RT_PATH_PROFILE
    if (lane == 0 && K.dbg) K.dbg[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = wall_clock64();
    if (k == 0) {
        atomicMax(P.stats + 24, (unsigned long long)dMaxRaysPix), atomicMax(P.stats + 25, (unsigned long long)dMaxStepsRay);
        atomicAdd(P.stats + 26, (unsigned long long)dRays), atomicAdd(P.stats + 27, (unsigned long long)dSteps);
    }
    if (lane == 0) {
        atomicAdd(P.stats + 28, (unsigned long long)(clock64() - t0)), atomicAdd(P.stats + 29, 1ull), atomicAdd(P.stats + 30, (unsigned long long)nStepIt);
        atomicAdd(P.stats + 31, (unsigned long long)nStepLanes);
        atomicAdd(P.stats + 15, (unsigned long long)nAdv), atomicAdd(P.stats + 14, (unsigned long long)nAdvLanes);
        atomicAdd(P.stats + 13, (unsigned long long)tAdv), atomicAdd(P.stats + 12, (unsigned long long)tSteps);
        unsigned long long* dbg = P.stats + 16;
        const unsigned long long tot = (unsigned long long)(clock64() - t0);
        const unsigned long long prev = atomicMax(dbg + 0, tot);
        if (tot > prev) {
            dbg[1] = tFetch, dbg[2] = tAdv, dbg[3] = tBegin, dbg[4] = tSteps, dbg[5] = nIter, dbg[6] = nAdv, dbg[7] = nStepIt;
        }
    }
#endif
    // what is left of the wave's reserved queue slots: holes
    for (uint32_t i = shadeC.next + lane; i < shadeC.end && i < P.shadeCap; i += 64u)
        reinterpret_cast<uint32_t*>(P.shadeQ + i)[0] = kHolePixel;
    for (uint32_t i = leafC.next + lane; i < leafC.end && i < P.leafCap; i += 64u)
        store_leaf_task(P.leafQ + i, kHolePixel, splat(0.0f), make3(0.0f, 0.0f, 1.0f), -1.0f, splat(0.0f));
    flush_counters(P.stats, lane, cPrimary, 0, cSecondary, tc, 0, cDrop);
}

// trace_kernel with four lanes per ray on the query tree's quad form (quad_step): rays in, {t, triangle} out, rays whose
// result is not provably the reference's appended to the fall-back list.
__global__ __launch_bounds__(kBlock) void quad_trace_kernel(const TraceParams P)
{
    constexpr int kQuads = kBlock / 4;
    __shared__ uint32_t s_q[kQuads * 2 * kQuadStack];
    const SceneView& S = P.S;
    const int lane = threadIdx.x & 63;
    const int k = lane & 3, leaderLane = lane - k, quadShift = leaderLane;
    uint32_t* const stack = s_q + (threadIdx.x / 4) * 2 * kQuadStack;
    const uint32_t have = *P.countPtr;
    const uint32_t items = have > P.countSub ? min(have - P.countSub, P.countMaxItems) : 0u;
    const uint32_t total = items * P.countMul;
    if (total < P.runLo || total >= P.runHi) return;
    const uint32_t gridQuads = gridDim.x * kQuads;
    const uint32_t myQuad = blockIdx.x * kQuads + threadIdx.x / 4;
    if ((blockIdx.x * kBlock + (threadIdx.x & ~63u)) / 4 >= total) return; // nothing for this wave, now or later
    bool has = false, exhausted = total <= gridQuads, first = true; // has: same on every lane of a quad
    uint32_t idx = 0, chunkNext = 0, chunkEnd = 0;
    QuadLane T;
    T.R.o = T.R.d = T.R.r = splat(0.0f);
    T.R.fast = T.R.approx = false;
    T.R.nor = splat(0.0f), T.R.eps = 0.0f;
    T.bestT = 0.0f, T.tieT = -1.0f, T.bestTri = kNoHit, T.cur = kTraversalDone, T.lost = 0, T.sp = 0;
    for (;;) {
        const unsigned long long idle = __ballot(!has && k == 0);
        if (idle != 0ull && (first || !exhausted || chunkNext < chunkEnd)) {
            uint32_t next = 0xffffffffu;
            if (first) {
                next = myQuad;
            } else {
                if (chunkNext == chunkEnd) { // !exhausted here
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(P.fetch, kGroupChunk);
                    base = __shfl(base, 0) + gridQuads;
                    exhausted = base + kGroupChunk >= total;
                    chunkNext = min(base, total), chunkEnd = min(base + kGroupChunk, total);
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << leaderLane) - 1ull));
                const uint32_t avail = chunkEnd - chunkNext, n = (uint32_t)__popcll(idle);
                if (rank < avail) next = chunkNext + rank;
                chunkNext += min(n, avail);
            }
            first = false;
            bool unfit = false;
            if (!has && next < total) {
                idx = (P.indirect && total >= P.indirectFrom) ? P.indirect[next] : next;
                const float4* r = reinterpret_cast<const float4*>(P.rays + (size_t)idx * P.rayStride);
                const float4 a = r[0], b = r[1];
                T.R = make_slabs(S, make3(a.x, a.y, a.z), make3(b.x, b.y, b.z));
                T.bestT = a.w, T.bestTri = kNoHit, T.tieT = -1.0f, T.lost = 0, T.sp = 0, T.cur = 0; // root of the 4-wide form: node 0
                float tr;
                bool outside;
                // root box against the ray's own t; a miss ends the query before the sphere loop (src/bvh.cpp:264-268)
                if (!box_test_any(S.rootLo[0], S.rootLo[1], S.rootLo[2], S.rootHi[0], S.rootHi[1], S.rootHi[2], T.R, a.w, tr, outside)) {
                    if (k == 0) P.hits[idx] = HitRec { a.w, kNoHit };
                } else if (!T.R.approx) {
                    unfit = true; // the conservative tests need the approximate slabs
                } else {
                    has = true;
                }
            }
            if (__ballot(unfit) != 0ull) {
                const uint32_t pos = wave_fetch(P.fallCount, unfit && k == 0, lane);
                if (unfit && k == 0) P.fallList[pos] = idx;
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted && chunkNext == chunkEnd) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < 4; it++) {
            quad_step(S, T, stack, k, quadShift, has);
            const bool done = has && T.cur == kTraversalDone;
            bool unproven = false;
            if (done) {
                unproven = !query_result_proven(S, T.R, T.bestT, T.bestTri, T.tieT, T.lost);
                for (uint32_t s = 0; s < S.nSpheres; s++) // src/bvh.cpp:366-367
                    if (sphere_test(S.spheres[s], T.R.o, T.R.d, T.bestT)) T.bestTri = kSphereBase + s;
                if (!unproven && k == 0) P.hits[idx] = HitRec { T.bestT, T.bestTri };
                has = false;
            }
            if (__ballot(unproven) != 0ull) {
                const uint32_t pos = wave_fetch(P.fallCount, unproven && k == 0, lane);
                if (unproven && k == 0) P.fallList[pos] = idx;
            }
        }
    }
}
