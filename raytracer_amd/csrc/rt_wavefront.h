// Wavefront render pipeline (included by rt_device.hip inside its anonymous namespace).
//
// The reference evaluates each pixel as one depth-first recursion (src/recursive.cpp:25-83) whose
// glossy branch draws from the pixel's sampler in recursion order (src/extra.cpp:489-505).  On a
// GPU a pixel with a full glossy tree is a chain of several hundred dependent rays.  Only part of
// that chain is really sequential:
//   * shadow rays never draw from the sampler and never change control flow: they only scale a
//     term that is added to the pixel;
//   * rays spawned at rayDepth 6 cannot spawn anything (recursion stops at rayDepth < 6), so with
//     point lights they draw nothing either.
// So the path kernel walks only the sequential part - the radiance rays whose outcome decides later
// sampler offsets - in the reference's exact order, and emits everything else as independent tasks:
//   primary_kernel / round_kernel  per pixel DFS over radiance rays, one ray per active pixel per round;
//                 append ShadeTask / LeafTask records
//   leaf_kernel   traces deferred last-level rays; a hit appends a ShadeTask
//   shade_kernel  light loop (shadow rays + shading model) per ShadeTask; adds weight * radiance to the pixel
//   resolve_kernel converts the pixel accumulators to float RGB in Screen order
// Pixel sums are kept in 64-bit fixed point so that the result does not depend on the order in which
// tasks finish (integer adds commute): a frame is reproducible bit for bit, and equal to the
// reference's nested float sums to ~1e-7 relative.
#pragma once

struct ShadeTask { // 64 B
    uint32_t pixel;   // Screen index (H-1-y)*W + x
    uint32_t matInfo; // bit 31: sphere material, low bits: mesh / sphere index
    uint32_t smp;     // sampler state at the start of this hit's light loop (area lights draw from it)
    uint32_t pad;
    float W[3], px;   // weight of this hit's direct light in the pixel; hit point p
    float py, pz, dx, dy; // incoming ray direction d
    float dz, nx, ny, nz; // shading normal
};
static_assert(sizeof(ShadeTask) == 64, "ShadeTask layout");
struct LeafTask { // 48 B
    uint32_t pixel, pad;
    float W[3];
    float ox, oy, oz, dx, dy, dz;
    float pad2;
};
static_assert(sizeof(LeafTask) == 48, "LeafTask layout");

constexpr double kFixScale = 1099511627776.0; // 2^40
constexpr float kFixLimit = 4.0e6f;           // |value| beyond this is recorded as non-finite

struct WaveParams {
    SceneView S;
    rt_camera cam;
    rt_features F;
    int W, H, startDepth;
    int tilesX, myTileRows, stripeTileRows, rank, world;
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
    float* states;           // PathState pool: kStateDwords floats per state
    const uint32_t* activeIn; // state indices to process this round (round_kernel)
    uint32_t* activeOut;      // survivors
    const uint32_t* activeInCount;
    uint32_t* activeOutCount;
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

__device__ __forceinline__ void flush_counters(unsigned long long* stats, const int lane, const uint32_t cPrimary, const uint32_t cShadow,
                                               const uint32_t cSecondary, const TraceCounters& tc, const uint32_t cPixels, const uint32_t cDrop)
{
    const uint32_t vals[8] = { cPrimary, cShadow, cSecondary, tc.nodePairs, tc.triTests, cPixels, tc.overflow, cDrop };
#pragma unroll
    for (int i = 0; i < 8; i++) {
        unsigned long long v = vals[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && v) atomicAdd(stats + i, v);
    }
}

// ---------------------------------------------------------------------------------------------
// Path stage: the sequential part of each pixel, one radiance ray per active pixel per round.
//
// A pixel's depth-first state lives in HBM between rounds (PathState: a 64-byte header with the ray
// to trace next + one 80-byte frame per pending glossy loop).  primary_kernel traces the camera rays
// (pixels fetched tile by tile from a shared counter) and creates a state only for pixels that still
// have a ray to trace; round_kernel r takes the compact list of active states, traces one ray for
// each, advances its recursion to the next ray and appends it to the list of round r+1.  Every round
// therefore runs dense waves, however uneven the pixels' chains are.
// ---------------------------------------------------------------------------------------------
constexpr int kFrameDwords = 20;                                   // 80 B per pending glossy loop
constexpr int kStateDwords = 16 + RT_MAX_RAY_DEPTH * kFrameDwords; // 544 B
enum PathStep : int { PS_DONE = 0, PS_TRACE = 1, PS_HIT, PS_GLOSSY, PS_POP };

struct PathLane {
    // header
    uint32_t pixel, smp, seed, pxy; // pxy = px | py << 16
    vec3 to, td, curW;
    int depth, loopDepth; // rayDepth of the ray in flight; rayDepth of the innermost pending glossy loop (-1: none)
    uint32_t sampleIdx;
    // working
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
}
__device__ __forceinline__ void store_header(float* s, const PathLane& L)
{
    float4* q = reinterpret_cast<float4*>(s);
    const uint32_t pk = (uint32_t)L.depth | ((uint32_t)(L.loopDepth + 1) << 8) | (L.sampleIdx << 16);
    q[0] = make_float4(__uint_as_float(L.pixel), __uint_as_float(L.smp), __uint_as_float(L.seed), __uint_as_float(L.pxy));
    q[1] = make_float4(L.to.x, L.to.y, L.to.z, L.td.x);
    q[2] = make_float4(L.td.y, L.td.z, L.curW.x, L.curW.y);
    q[3] = make_float4(L.curW.z, __uint_as_float(pk), 0.0f, 0.0f);
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
    const uint32_t px = L.pxy & 0xffffu, py = L.pxy >> 16;
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
    camera_ray(P.cam, ndcX, ndcY, L.to, L.td);
    L.curW = splat(1.0f / static_cast<float>(nPixelRays)); // renderRays: L / rays.size()
    L.depth = P.startDepth;
    L.loopDepth = -1;
    L.sampleIdx = k;
}

// Advances every lane from "its ray has been traced" (PS_HIT) to its next ray (PS_TRACE) or the end
// of its pixel (PS_DONE), appending the shade / leaf tasks met on the way.  Wave-convergent.
// frames: this lane's frame storage in HBM (null while the lane has none: lanes allocate lazily).
template <typename AllocFn>
__device__ __forceinline__ void advance_lanes(const WaveParams& P, PathLane& L, float*& frames, AllocFn&& allocFrames, const int lane,
                                              uint32_t& cPrimary, uint32_t& cSecondary, uint32_t& cDrop)
{
    const SceneView& S = P.S;
    const rt_features& F = P.F;
    const bool deferLeaves = (P.lightDraws == 0);
    const uint32_t nPixelRays = pixel_ray_count(F);
    ShadeTask st;
    LeafTask lt;
    do {
        bool wantShade = false, wantLeaf = false, wantFrames = false;
        while (L.state > PS_TRACE && !wantShade && !wantLeaf && !wantFrames) {
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
                st.pad = 0;
                st.W[0] = L.curW.x, st.W[1] = L.curW.y, st.W[2] = L.curW.z;
                st.px = p.x, st.py = p.y, st.pz = p.z;
                st.dx = L.td.x, st.dy = L.td.y, st.dz = L.td.z;
                st.nx = h.normal.x, st.ny = h.normal.y, st.nz = h.normal.z;
                wantShade = true; // the hit's direct light: deferred to the shade kernel
                for (uint32_t k = 0; k < P.lightDraws; k++) (void)sampler_next(L.smp); // the light loop's own draws
                L.state = PS_POP;
                if (L.depth < RT_MAX_RAY_DEPTH && F.enableReflections) {
                    const Mat m = load_material(S, h.mesh, h.isSphere);
                    const bool reflective = (m.ks.x != 0.0f) || (m.ks.y != 0.0f) || (m.ks.z != 0.0f);
                    if (reflective) {
                        // generateReflectionRay (src/recursive.cpp:91-103)
                        vec3 rorg = p;
                        const vec3 rdir = L.td - 2.f * dot(L.td, h.normal) * h.normal;
                        rorg = rorg + 0.00001f * rdir;
                        if (!F.enableGlossyReflection) { // renderRaySpecularComponent: weight * ks
                            L.to = rorg, L.td = rdir;
                            L.curW = L.curW * m.ks;
                            L.depth++;
                            cSecondary++;
                            L.state = PS_TRACE;
                        } else if (F.numGlossySamples > 0) { // renderRayGlossyComponent (src/extra.cpp:464-508)
                            // the frame is parked in the task record's neighbours until storage exists
                            L.to = rorg, L.td = rdir; // reused as scratch: reflection ray of the new loop
                            L.tt = 0.5f / m.shininess;
                            st.pad = 1; // marks "push a glossy frame after this append"
                            // ks and normal are re-read from st / material when the frame is written
                        }
                    }
                }
                break;
            }
            case PS_GLOSSY: {
                float4* fq = reinterpret_cast<float4*>(frames + L.loopDepth * kFrameDwords);
                const float4 f0 = fq[0], f1 = fq[1], f2 = fq[2], f3 = fq[3], f4 = fq[4];
                const vec3 fW = make3(f0.x, f0.y, f0.z), rorg = make3(f0.w, f1.x, f1.y), rdir = make3(f1.z, f1.w, f2.x);
                const vec3 nrm = make3(f2.y, f2.z, f2.w), ks = make3(f3.x, f3.y, f3.z);
                const float radius = f3.w;
                uint32_t i = __float_as_uint(f4.x);
                const int enclosing = (int)__float_as_uint(f4.y) - 1;
                vec3 w, u, v;
                ortho_basis(rdir, w, u, v);
                bool launched = false;
                while (i < F.numGlossySamples) {
                    // sampleDisk (src/extra.cpp:443-450): two draws per sample, consumed even if rejected
                    const float r = radius * sqrtf(sampler_next(L.smp));
                    const float angle = sampler_next(L.smp) * 2 * 3.14159265358979323846264338327950288f;
                    const float dx = r * cosf(angle), dy = r * sinf(angle);
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
                        lt.pixel = L.pixel, lt.pad = 0;
                        lt.W[0] = cw.x, lt.W[1] = cw.y, lt.W[2] = cw.z;
                        lt.ox = corg.x, lt.oy = corg.y, lt.oz = corg.z;
                        lt.dx = pd.x, lt.dy = pd.y, lt.dz = pd.z;
                        lt.pad2 = 0.0f;
                        wantLeaf = true; // leave to append; come back to PS_GLOSSY
                        break;
                    }
                    L.to = corg, L.td = pd;
                    L.curW = cw;
                    L.depth = L.loopDepth + 1;
                    L.state = PS_TRACE;
                    break;
                }
                frames[L.loopDepth * kFrameDwords + 16] = __uint_as_float(i);
                if (!launched) { // loop finished: resume the enclosing loop
                    L.loopDepth = enclosing;
                    L.state = PS_POP;
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
            default: L.state = PS_DONE; break;
            }
        }
        // ---- append the tasks produced in this pass (wave-aggregated) ----
        if (__ballot(wantShade) != 0ull) {
            const bool pushFrame = wantShade && st.pad == 1;
            st.pad = 0;
            if (!queue_append(P.shadeQ, P.qCount + 0, P.shadeCap, wantShade, st, lane)) cDrop++;
            // lanes that open a glossy loop need frame storage; allocate for those that have none yet
            if (__ballot(pushFrame && frames == nullptr) != 0ull) allocFrames(pushFrame && frames == nullptr);
            if (pushFrame) {
                const Mat m = load_material(S, st.matInfo & 0x7fffffffu, (st.matInfo >> 31) != 0);
                float4* fq = reinterpret_cast<float4*>(frames + L.depth * kFrameDwords);
                fq[0] = make_float4(L.curW.x, L.curW.y, L.curW.z, L.to.x);
                fq[1] = make_float4(L.to.y, L.to.z, L.td.x, L.td.y);
                fq[2] = make_float4(L.td.z, st.nx, st.ny, st.nz);
                fq[3] = make_float4(m.ks.x, m.ks.y, m.ks.z, L.tt);
                fq[4] = make_float4(__uint_as_float(0u), __uint_as_float((uint32_t)(L.loopDepth + 1)), 0.0f, 0.0f);
                L.loopDepth = L.depth;
                L.state = PS_GLOSSY;
            }
        }
        if (__ballot(wantLeaf) != 0ull) {
            if (!queue_append(P.leafQ, P.qCount + 1, P.leafCap, wantLeaf, lt, lane)) cDrop++;
        }
    } while (__ballot(L.state > PS_TRACE) != 0ull);
}

// Round 0: camera rays.  Persistent lanes fetch pixels (8x8 tiles in fetch order) until none are left.
template <bool CULL, bool COUNT>
__global__ __launch_bounds__(kBlock) void primary_kernel(const WaveParams P)
{
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t* const stack = s_stack + threadIdx.x;
    const rt_features& F = P.F;
    const int lane = threadIdx.x & 63;
    const uint32_t nPixelRays = pixel_ray_count(F);
    uint32_t cPrimary = 0, cSecondary = 0, cPixels = 0, cDrop = 0;
    TraceCounters tc { 0, 0, 0 };
    for (;;) {
        const uint32_t w = wave_fetch(P.work + 0, true, lane);
        if (__ballot(w < P.nWork) == 0ull) break;
        PathLane L;
        L.state = PS_DONE;
        if (w < P.nWork) {
            const uint32_t tile = w >> 6, l = w & 63;
            const uint32_t m = tile / P.tilesX, tx = tile - m * P.tilesX;
            const uint32_t blk = m / P.stripeTileRows, r = m - blk * P.stripeTileRows;
            const uint32_t ty = (blk * P.world + P.rank) * P.stripeTileRows + r;
            const uint32_t px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
            if (px < (uint32_t)P.W && py < (uint32_t)P.H) {
                cPixels++;
                L.pixel = (uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px; // Screen::indexAt (src/screen.cpp:119-122)
                L.seed = static_cast<uint32_t>(P.H * (int)px + (int)py);      // src/render.cpp:38
                L.smp = L.seed;
                L.pxy = px | (py << 16);
                for (uint32_t k = 0; k < (F.numPixelSamples > 1 ? 2 * nPixelRays : 0u); k++) (void)sampler_next(L.smp);
                if (P.primId) P.primId[py * P.W + px] = kNoHit;
                if (P.primT) P.primT[py * P.W + px] = FLT_MAX;
                make_pixel_ray(P, L, 0, nPixelRays);
                cPrimary++;
                L.tt = FLT_MAX;
                trace<CULL, COUNT>(P.S, F.enableAccelStructure != 0, L.to, L.td, L.tt, L.ttri, stack, tc);
                L.state = PS_HIT;
            }
        }
        float* frames = nullptr;
        uint32_t slot = 0xffffffffu;
        auto alloc = [&](const bool need) {
            const uint32_t s = wave_fetch(P.qCount + 2, need, lane);
            if (need) {
                slot = s;
                frames = P.states + (size_t)s * kStateDwords + 16;
            }
        };
        advance_lanes(P, L, frames, alloc, lane, cPrimary, cSecondary, cDrop);
        // pixels that still have a ray to trace become active states of round 1
        const bool active = (L.state == PS_TRACE);
        if (__ballot(active && slot == 0xffffffffu) != 0ull) alloc(active && slot == 0xffffffffu);
        const uint32_t pos = wave_fetch(P.activeOutCount, active, lane);
        if (active) {
            store_header(P.states + (size_t)slot * kStateDwords, L);
            P.activeOut[pos] = slot;
        }
    }
    flush_counters(P.stats, lane, cPrimary, 0, cSecondary, tc, cPixels, cDrop);
}

// Round r >= 1: one ray for every active state of list `activeIn`; survivors go to `activeOut`.
template <bool CULL, bool COUNT>
__global__ __launch_bounds__(kBlock) void round_kernel(const WaveParams P)
{
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t* const stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t total = *P.activeInCount;
    uint32_t cPrimary = 0, cSecondary = 0, cDrop = 0;
    TraceCounters tc { 0, 0, 0 };
    for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        PathLane L;
        L.state = PS_DONE;
        float* frames = nullptr;
        uint32_t slot = 0;
        if (i < total) {
            slot = P.activeIn[i];
            float* s = P.states + (size_t)slot * kStateDwords;
            load_header(s, L);
            frames = s + 16;
            L.tt = FLT_MAX;
            trace<CULL, COUNT>(P.S, P.F.enableAccelStructure != 0, L.to, L.td, L.tt, L.ttri, stack, tc);
            L.state = PS_HIT;
        }
        auto alloc = [&](const bool) {}; // every lane of a round already owns its state's frame storage
        advance_lanes(P, L, frames, alloc, lane, cPrimary, cSecondary, cDrop);
        const bool active = (L.state == PS_TRACE);
        const uint32_t pos = wave_fetch(P.activeOutCount, active, lane);
        if (active) {
            store_header(P.states + (size_t)slot * kStateDwords, L);
            P.activeOut[pos] = slot;
        }
    }
    flush_counters(P.stats, lane, cPrimary, 0, cSecondary, tc, 0, cDrop);
}

// ---------------------------------------------------------------------------------------------
// Leaf kernel: deferred rays at the recursion limit.  A hit becomes a ShadeTask.
// ---------------------------------------------------------------------------------------------
template <bool CULL, bool COUNT>
__global__ __launch_bounds__(kBlock) void leaf_kernel(const WaveParams P)
{
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t* const stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t total = min(P.qCount[1], P.leafCap);
    TraceCounters tc { 0, 0, 0 };
    uint32_t cDrop = 0;
    for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool active = i < total;
        ShadeTask st;
        bool want = false;
        if (active) {
            const float4* q = reinterpret_cast<const float4*>(P.leafQ + i);
            const float4 a = q[0], b = q[1], c = q[2];
            const uint32_t pixel = __float_as_uint(a.x);
            const vec3 W = make3(a.z, a.w, b.x), o = make3(b.y, b.z, b.w), d = make3(c.x, c.y, c.z);
            float t = FLT_MAX;
            uint32_t tri;
            trace<CULL, COUNT>(P.S, P.F.enableAccelStructure != 0, o, d, t, tri, stack, tc);
            if (tri != kNoHit) {
                Hit h;
                fill_hit(P.S, P.F, tri, o, d, t, h);
                const vec3 p = o + t * d;
                st.pixel = pixel;
                st.matInfo = h.mesh | (h.isSphere ? 0x80000000u : 0u);
                st.smp = 0, st.pad = 0;
                st.W[0] = W.x, st.W[1] = W.y, st.W[2] = W.z;
                st.px = p.x, st.py = p.y, st.pz = p.z;
                st.dx = d.x, st.dy = d.y, st.dz = d.z;
                st.nx = h.normal.x, st.ny = h.normal.y, st.nz = h.normal.z;
                want = true;
            }
        }
        if (!queue_append(P.shadeQ, P.qCount + 0, P.shadeCap, want, st, lane)) cDrop++;
    }
    flush_counters(P.stats, lane, 0, 0, 0, tc, 0, cDrop);
}

// ---------------------------------------------------------------------------------------------
// Shade kernel: computeLightContribution (src/light.cpp:262-276) for one hit per lane.
// ---------------------------------------------------------------------------------------------
template <bool CULL, bool COUNT>
__global__ __launch_bounds__(kBlock) void shade_kernel(const WaveParams P)
{
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t* const stack = s_stack + threadIdx.x;
    const SceneView& S = P.S;
    const rt_features& F = P.F;
    const int lane = threadIdx.x & 63;
    const uint32_t total = min(P.qCount[0], P.shadeCap);
    TraceCounters tc { 0, 0, 0 };
    uint32_t cShadow = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
        const float4* q = reinterpret_cast<const float4*>(P.shadeQ + i);
        const float4 a = q[0], b = q[1], c = q[2], e = q[3];
        const uint32_t pixel = __float_as_uint(a.x), matInfo = __float_as_uint(a.y);
        uint32_t smp = __float_as_uint(a.z);
        const vec3 W = make3(b.x, b.y, b.z), p = make3(b.w, c.x, c.y), rd = make3(c.z, c.w, e.x), hn = make3(e.y, e.z, e.w);
        const Mat m = load_material(S, matInfo & 0x7fffffffu, (matInfo >> 31) != 0);
        vec3 Lo = splat(0.0f);
        for (uint32_t li = 0; li < S.nLights; li++) {
            const GpuLight& L = S.lights[li];
            const int type = (int)L.type;
            const uint32_t nSamples = (type == 0) ? 1u : F.numShadowSamples;
            vec3 areaAcc = splat(0.0f);
            for (uint32_t s = 0; s < nSamples; s++) {
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
                vec3 visible = lcol;
                if (F.enableShadows) { // visibilityOfLightSampleBinary (src/light.cpp:66-90)
                    const vec3 sd = normalize(p - lpos);
                    float t = FLT_MAX;
                    uint32_t tri;
                    cShadow++;
                    trace<CULL, COUNT>(S, F.enableAccelStructure != 0, lpos, sd, t, tri, stack, tc);
                    const bool expected = fabsf(t - length(p - lpos)) <= 5e-4f;
                    if (!(tri != kNoHit && expected)) visible = splat(0.0f);
                }
                if (type == 0) { // computeContributionPointLight (src/light.cpp:151-159)
                    Lo = Lo + shade(F, -rd, normalize(lpos - p), visible, hn, m);
                } else { // the sampled POSITION is handed over as the light direction (src/light.cpp:191, 228)
                    areaAcc = areaAcc + shade(F, -rd, lpos, visible, hn, m) / (1.0f * F.numShadowSamples);
                }
            }
            if (type != 0) Lo = Lo + areaAcc;
        }
        accumulate_pixel(P, pixel, W * Lo);
    }
    flush_counters(P.stats, lane, 0, cShadow, 0, tc, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Resolve: fixed-point sums -> float RGB (Screen order) for this rank's stripes; clears the sums.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void resolve_kernel(const WaveParams P)
{
    for (uint32_t w = blockIdx.x * kBlock + threadIdx.x; w < P.nWork; w += gridDim.x * kBlock) {
        const uint32_t tile = w >> 6, l = w & 63;
        const uint32_t m = tile / P.tilesX, tx = tile - m * P.tilesX;
        const uint32_t blk = m / P.stripeTileRows, r = m - blk * P.stripeTileRows;
        const uint32_t ty = (blk * P.world + P.rank) * P.stripeTileRows + r;
        const uint32_t px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        if (px >= (uint32_t)P.W || py >= (uint32_t)P.H) continue;
        const uint32_t pixel = (uint32_t)(P.H - 1 - (int)py) * (uint32_t)P.W + px;
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
