// Device side of librt_amd.so for gfx950 (MI355X): BVH traversal, ray/box and ray/triangle tests,
// the recursive radiance estimate (lights, shadow rays, specular and glossy reflections) and the
// bloom post-process, plus the C ABI around them.
//
// Arithmetic contract: IEEE fp32, no contraction (-ffp-contract=off), correctly rounded division
// and square root, denormals kept.  Expression order follows the reference expression cited at
// each function because float ties decide which triangle a ray reports.
#include "rt_types.h"
#include "rt_tri.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace rt;

// Everything one frame in flight needs besides the scene: counters, queues, path states.  A scene keeps one set per stream
// that frames are queued on, so frames on different streams may overlap on the GPU (the rounds of one frame are chains of
// short dependent launches that leave most of the chip idle; a second frame fills it).
struct Workspace {
    hipStream_t stream = nullptr;
    uint64_t lastUse = 0;
    unsigned long long* d_stats = nullptr; // kStatCopies copies of the S_COUNT counters
    uint32_t* d_work = nullptr;            // work-fetch counter
    rt_camera* d_cams = nullptr;           // cameras of a launch sequence that renders several frames (RT_MAX_FRAMES entries)
    // wavefront pipeline buffers (sized on demand by rt_render)
    long long* d_accum = nullptr;
    uint32_t* d_nonfinite = nullptr;
    size_t accumPixels = 0;
    void* d_shadeQ = nullptr;
    void* d_leafQ = nullptr;
    uint32_t shadeCap = 0, leafCap = 0;
    uint32_t lastLeafCap = 0;       // leaf tasks the last frame's kernels were given room for (leafCap, or the multiple of its loop size below)
    float* d_states = nullptr;      // path states, one slot per pixel of the region: headers alone (rounds) or whole records (persistent launches)
    size_t stateWords = 0;          // ... its size in floats
    float* d_frames = nullptr;      // rounds: pending glossy loops, by rayDepth then state (WaveParams::frameBase)
    float* d_passes = nullptr;      // rounds with transparency: pending pass-through rays, by rayDepth then state
    size_t frameStates = 0, passStates = 0; // states x levels the two hold
    uint32_t* d_active[2] = { nullptr, nullptr };
    uint32_t* d_activeRay[2] = { nullptr, nullptr }; // first ray of each active state
    size_t pathRayCap = 0;
    uint32_t* d_roundCounts = nullptr; // [0, kMaxRounds): active-list size per round; then one work counter per trace launch
    uint32_t stateCap = 0;
    void* d_pathRays[2] = { nullptr, nullptr }; // rays of the even / odd path rounds
    void* d_hits = nullptr;       // hit records of the stage being traced
    size_t hitCap = 0;
    void* d_shadowRays = nullptr; // shadow rays of one chunk of shade tasks
    uint32_t* d_longList = nullptr; // indices of the rays a one-ray-per-lane launch handed over (hitCap entries)
    uint32_t* d_longList2 = nullptr; // ... and of those its second pass handed over in turn
    uint32_t* d_orderKeys = nullptr; // ray order of a big stage (hitCap entries each): keys, ray indices in key order,
    uint32_t* d_order = nullptr;
    uint32_t* d_orderHist = nullptr; // ... and the buckets' counters (kOrderBuckets)
    uint32_t* d_fallList = nullptr; // indices of the rays a query-tree launch could not settle (hitCap entries)
    size_t shadowCap = 0;
    bool lastWasWavefront = false;
    uint32_t dbgWaves = 0;
    std::vector<uint32_t> sliceSlots; // d_roundCounts entries that count the pixels every slice of the path stage set aside
    std::vector<uint32_t> fallSlots; // d_roundCounts entries that count the last frame's fall-back queries
    std::vector<uint32_t> longSlots; // ... and the rays its one-ray-per-lane launches handed over
    // Launch plan carried from one frame to the next: rt_sync notes how many rays every round of the frame held; the next
    // frame of the same description (size, depth, flags, features, region) leaves out the launches that only serve
    // larger rounds.  The plan is a hint: a round that turns out larger is still traced completely, by the group kernel alone.
    uint64_t lastKey = 0; // what shaped the rounds of the frame last queued here
    uint32_t lastRounds = 0;
    uint32_t lastWork = 0; // pixels (of all its frames) of the last wavefront sequence
    unsigned long long* d_waveTimes = nullptr; // RT_AMD_WAVE_TIMES: start / end clock of every wave of one round's group launch
    uint32_t waveTimesWaves = 0;
    uint64_t lastCfg = 0;  // ... and the description its queue sizes are remembered under (rt_scene::seenQueues)
    // What rt_sync reads of a frame - the counters, the queue counters, the rounds' counters - is copied to pinned host memory
    // by the frame's own launch sequence: rt_sync then waits for the stream once instead of once per blocking copy.
    unsigned char* h_status = nullptr;
    bool statusQueued = false;     // the last sequence queued here ends with those copies
    uint32_t statusCountWords = 0; // d_roundCounts words among them
};

struct RetiredFrame {
    hipStream_t stream;
    rt_stats stats;
    int status;
    std::string message;
};

// =============================================================================================
// Host-side scene object
// =============================================================================================
struct rt_scene {
    int device = 0;
    SceneView view {};
    SceneView queryView {}; // the same scene with the query tree in place of the reference's (cert != null when usable)
    uint32_t queryDepth = 0;
    void* d_blob = nullptr; // one allocation holding every scene array
    size_t blobBytes = 0;
    // lazily sized device buffers for rt_render_image
    float* d_rgb = nullptr;
    float* d_scratch = nullptr;
    uint32_t* d_primId = nullptr;
    float* d_primT = nullptr;
    size_t fbPixels = 0;
    // rt_render_images: up to four launch sequences in flight, each on a stream of its own with its frames on the device, the
    // bloom's intermediate image and a pinned host buffer the finished frames are copied into (all sized lazily)
    struct ImageLane {
        hipStream_t stream = nullptr;
        float* d_rgb = nullptr;
        float* d_scratch = nullptr;
        float* h_stage = nullptr;
        size_t frames = 0, px = 0;
    };
    ImageLane imageLanes[4];
    bool hasTextures = false;
    GpuTexture* d_textures = nullptr;  // one entry per material (part of no blob: rt_scene_set_texture rewrites entries)
    std::vector<float*> texturePixels; // device copy of each material's texels (null: none)
    uint32_t treeDepth = 0;
    std::vector<RetiredFrame> retired; // what rt_sync reports for streams whose buffers were taken over
    // launch plan: rays per path round of the last frame rt_sync looked at, and what shaped that frame's rounds; frames of the same
    // shape on any stream size their launches by it
    std::vector<uint32_t> roundRays;
    uint64_t planKey = 0;
    // task queue sizes a frame of this scene has asked for (a frame that overflowed its queues is reported by rt_sync with
    // RT_ERR_AGAIN; the next rt_render allocates these)
    uint64_t wantShadeTasks = 0, wantLeafTasks = 0, wantPathRays = 0;
    // What completed frames of this scene held at most, per pixel: shade tasks, leaf tasks, rays of one path round.  Until a frame
    // has completed the buffers are sized for the worst case the pipeline accepts; afterwards for half as much again as was seen
    // (a frame that needs more reports RT_ERR_AGAIN and the buffers grow).
    // (kept per description of the frame - features and depth - since one scene may be rendered in very different ways)
    struct SeenQueues {
        double shade = -1.0, leaf = -1.0, rays = -1.0;
    };
    std::unordered_map<uint64_t, SeenQueues> seenQueues;
    std::vector<struct Workspace*> workspaces; // per-frame buffers, one set per stream that frames are queued on (see workspace_for)
    uint64_t useCounter = 0;
    int numCUs = 256;
    uint32_t laneBlocksPerCU = 8; // grid of the one-ray-per-lane trace kernel (RT_AMD_TRACE_BLOCKS overrides)
    // ... of a launch sequence of several frames, or of frames that share the GPU with others: as many workgroups as a CU holds at
    // a time (four: 34 KB of LDS each), so that a lane walks several rays in turn and the wave's lanes are refilled, instead of one
    // ray per lane and a second generation of workgroups that each wait for their slowest ray (RT_AMD_TRACE_BLOCKS_POOLED overrides)
    uint32_t laneBlocksPerCUPooled = 4;
    uint32_t logicBlocksPerCU = 8; // workgroups per CU of a path_logic_kernel launch (RT_AMD_LOGIC_BLOCKS; a CU holds two at a time, which fill its registers)
    uint32_t groupBlocksPerCU = 0; // grid of the group trace kernel; 0 = one group per possible ray (RT_AMD_GROUP_BLOCKS overrides)
    uint32_t groupLanes = 8; // lanes sharing one ray in the path rounds (RT_AMD_GROUP_LANES overrides; 1 = one ray per lane)
    uint32_t batchWindow = 5;       // children of a last-level glossy loop traced per round as candidates (RT_AMD_BATCH_WINDOW; below 2 and
                                    // RT_AMD_NO_BATCH: one child per round; at most 5)
    bool noPlan = false, scatter = false, verbose = false; // RT_AMD_NO_PLAN, RT_AMD_SCATTER, RT_AMD_VERBOSE (read once, at scene creation)
    int waveTimesRound = -1; // RT_AMD_WAVE_TIMES=<round>: record when the waves of that round's group launch start and end (printed by rt_sync with RT_AMD_VERBOSE)
    bool groupNode4 = false; // RT_AMD_GROUP_NODE4: group launches in the query tree walk its 4-wide form
    bool fullQueues = false; // RT_AMD_FULL_QUEUES: frame buffers stay sized for the worst case whatever earlier frames held
    bool noShadowBound = false;     // RT_AMD_NO_SHADOW_BOUND: shadow rays are traced to their closest hit like every other ray
    bool packets = true;            // the camera rays (the path stage's first round) are traced as packets of 64, one stack per wave
                                    // (RT_AMD_PACKETS=0: one by one like every other stage)
    uint32_t queryDepth4 = 0;       // depth of the query tree's 4-wide form
    uint32_t wideBelow = 32768;     // planned rounds expected to hold fewer rays run 16 lanes per ray (RT_AMD_WIDE_BELOW overrides, 0 = never)
    uint32_t laneModeAbove = 262144; // stages with at least this many rays run one ray per lane (RT_AMD_T1 overrides)
    uint32_t stepBudget = 96;        // ... and hand a ray over after this many steps (RT_AMD_BUDGET overrides, 0 = never)
    uint32_t queryStepBudget = 16;   // the same in the query tree, whose steps take two levels at a time (RT_AMD_QBUDGET)
    uint32_t orderRaysFrom = 0;      // the deferred stages (last-level rays, shadow rays) put their rays into a traversal order from this
                                     // many rays on (RT_AMD_ORDER_FROM; 0 = never, the default: on the 1080p glossy frame the ordered
                                     // launches are 8-13 % faster and the ordering costs several times that)
    uint32_t secondPassBudget = 0, secondPassBudgetOverlapped = 0; // steps of a handed-over ray's second one-ray-per-lane pass before it goes
                                     // on to the group launch (0: no second pass; RT_AMD_QBUDGET2, RT_AMD_QBUDGET2_OVERLAPPED)
    uint32_t laneModeAboveOverlapped = 65536, queryStepBudgetOverlapped = 20; // the same two for frames that overlap with others
                                     // (RT_RENDER_OVERLAPPED; RT_AMD_T1_OVERLAPPED, RT_AMD_QBUDGET_OVERLAPPED)
    // Launch sequences over many pixels (several frames rendered together, or one very large frame): every round holds so many rays
    // that its fixed costs no longer count, and what pays is fewer rays - candidate batches of two children (three candidates for
    // two rays instead of ten for four, in twice the rounds) - and fewer rays handed over (a longer walk before a lane gives up).
    uint32_t largeFromPixels = 4000000;   // RT_AMD_LARGE_FROM (pixels of the sequence's region; 0 = never)
    uint32_t batchWindowLarge = 2;        // RT_AMD_BATCH_WINDOW_LARGE
    uint32_t queryStepBudgetLarge = 32;   // RT_AMD_QBUDGET_LARGE
    bool quadTrace = false;          // group launches in a query tree run four lanes per ray (RT_AMD_QUAD_TRACE; default: 8 or 16)
    int pathQuads = 0;               // pixel-per-lane mode with a query tree: 1 = four lanes per pixel (frames in LDS), 2 = the same with frames in
                                     // global memory, 0 = one lane per pixel (RT_AMD_PATH_QUADS)
    int pathMode = 0;                // path stage: 0 = frame-wide rounds (and 2 for frames whose chains exceed the rounds' limit); 2 = one
                                     // persistent launch, a pixel per lane; 1 = the same, a group of lanes per pixel (RT_AMD_PATH_MODE overrides)
    uint32_t pathBlocksPerCU = 4;    // grid of the persistent launch (RT_AMD_PATH_BLOCKS overrides)
    uint32_t advanceAt = 16, advanceEvery = 2; // pixel per lane: see PathKParams (RT_AMD_ADVANCE_AT, RT_AMD_ADVANCE_EVERY)
    uint32_t linkBudget = 0, minLanes = 32;   // pixel per lane, slices: see PathKParams (RT_AMD_LINK_BUDGET, 0 = one launch; RT_AMD_MIN_LANES)
    std::vector<int> lightTypes;
};

static int collect_frame(rt_scene* scene, struct Workspace& w, rt_stats* stats);

#define RT_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            rt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* reported here: a later launch check must not find it again */   \
            return RT_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

namespace {

constexpr int kBlock = 256;     // 4 waves per workgroup
constexpr int kFrameFloats = 28; // per recursion level saved state
constexpr float kFltMin = 1.17549435e-38f;
constexpr uint32_t kNoHit = 0xffffffffu;
constexpr uint32_t kSphereBase = 0xf0000000u; // tri >= kSphereBase: sphere index = tri - kSphereBase
constexpr uint32_t kTraversalDone = 0xffffffffu;
constexpr uint64_t kMaxQueueTasks = 1ull << 29; // per queue
#ifndef RT_QSTACK
#define RT_QSTACK 16
#endif
constexpr int kQueryStack = RT_QSTACK; // per-lane stack entries of the one-ray-per-lane launches in the query tree (an entry that finds no room sends its ray to the caller's tree)
constexpr uint32_t kMaxRounds = 4096;            // sequential rays per pixel the round-based path stage accepts
constexpr uint32_t kTraceSlotsPerRound = 8; // counters one stage's launches take at most: hand-over list, fall-back list, up to six launches (second pass included)
constexpr uint32_t kMaxTraceLaunches = kTraceSlotsPerRound * kMaxRounds + 1024; // work counters for the trace launches of one frame
constexpr uint64_t kMaxShadowRays = 1ull << 26;  // shadow rays in flight per chunk of shade tasks
constexpr uint32_t kMaxShadowSegments = 64;      // surfaces a light sample is followed through (transparent shadows)
constexpr uint32_t kMaxBatchRays = 16;           // candidate rays of a batched glossy loop (numGlossySamples <= 5)

enum StatSlot { S_PRIMARY = 0, S_SHADOW, S_SECONDARY, S_NODEPAIRS, S_TRITESTS, S_PIXELS, S_OVERFLOW, S_DROPPED, S_BIGVALUE, S_NODE4, S_COUNT = 10 };
// Words of the FIRST copy beyond the counters: what the earlier frames of a launch sequence that is rendered frame by frame leave
// for rt_sync (carry_frame_counts_kernel) - the most shade / leaf tasks one of them appended, and their fall-back queries.
enum CarrySlot { S_QMAX_SHADE = 10, S_QMAX_LEAF = 11, S_FALL_SUM = 12 };
// The counters exist kStatCopies times, 256 B apart; a workgroup adds to the copy of its index and rt_sync sums them.
// (Additions to one address are served one at a time, about 9 ns each: 8,000 waves ending together would wait 70 us.)
constexpr uint32_t kStatCopies = 64, kStatStride = 32; // stride in counters
constexpr size_t kCountWords = 3 * kMaxRounds + kMaxTraceLaunches; // Workspace::d_roundCounts: active lists' fronts, rays per round, active lists' backs, trace launches
constexpr size_t kStatusStatsBytes = kStatCopies * kStatStride * sizeof(unsigned long long), kStatusWorkBytes = 256; // Workspace::h_status: d_stats, d_work, d_roundCounts

// Ends a frame's launch sequence with the copies of what rt_sync reads (Workspace::h_status).  Best effort: without them
// rt_sync copies as it reads.
static void queue_status(Workspace& w, hipStream_t st, uint32_t countWords)
{
    w.statusQueued = false;
    if (countWords > kCountWords) countWords = (uint32_t)kCountWords;
    if (!w.h_status && hipHostMalloc(reinterpret_cast<void**>(&w.h_status), kStatusStatsBytes + kStatusWorkBytes + kCountWords * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        w.h_status = nullptr;
        (void)hipGetLastError();
        return;
    }
    bool ok = hipMemcpyAsync(w.h_status, w.d_stats, kStatusStatsBytes, hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(w.h_status + kStatusStatsBytes, w.d_work, kStatusWorkBytes, hipMemcpyDeviceToHost, st) == hipSuccess;
    if (countWords)
        ok = ok && hipMemcpyAsync(w.h_status + kStatusStatsBytes + kStatusWorkBytes, w.d_roundCounts, countWords * sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess;
    w.statusQueued = ok, w.statusCountWords = countWords;
}

struct RenderParams {
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
    uint32_t* work;
};

// ---------------------------------------------------------------------------------------------
// Sampler (reference src/sampler.h:13-46)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sampler_next(uint32_t& state)
{
    state = state * 747796405u + 2891336453u;
    uint32_t v = state;
    v ^= v >> ((v >> 28u) + 4u);
    v *= 277803737u;
    v ^= v >> 22u;
    return static_cast<float>(v) / 4294967295.f;
}

// ---------------------------------------------------------------------------------------------
// sin / cos of a glossy sample's angle (sampleDisk, src/extra.cpp:443-450: glm::cos / glm::sin of a float = the host
// libm's cosf / sinf).  A sample is kept or rejected on the sign of dot(direction, normal) (src/extra.cpp:491-499), and
// from a rejection on the pixel's sampler sequence is another: the last bit of these two values decides ray COUNTS, so
// they are computed as the reference's libm (glibc 2.35, sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h,
// s_sincosf_data.c) computes them: in double, one multiply-subtract range reduction to [-pi/4, pi/4] with the quadrant
// from a scaled float-to-int conversion, then the sine or cosine polynomial of the quadrant, rounded to float once.
// Bit-identical to that libm's sinf / cosf for every float in [0, 2 pi] (all 1,086,918,685 of them were compared on
// the host, with and without fused multiply-adds: the double results never lie close enough to a rounding boundary
// for the contraction to matter); the angle of a sample lies in that interval.  Arguments of 120 and beyond, which the
// libm reduces with a table this function does not carry, are answered by the device's own sinf / cosf.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void libm_sincosf(const float y, float& sn, float& cs)
{
    constexpr double kHpiInv = 0x1.45F306DC9C883p+23; // 2 / pi * 2^24
    constexpr double kHpi = 0x1.921FB54442D18p0;      // pi / 2
    constexpr double kC0 = 0x1p0, kC1 = -0x1.ffffffd0c621cp-2, kC2 = 0x1.55553e1068f19p-5, kC3 = -0x1.6c087e89a359dp-10,
                     kC4 = 0x1.99343027bf8c3p-16;
    constexpr double kS1 = -0x1.555545995a603p-3, kS2 = 0x1.1107605230bc4p-7, kS3 = -0x1.994eb3774cf24p-13;
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu; // exponent and the first mantissa bits of |y|
    if (!(top < 0x42fu)) { // |y| >= 120, infinities, NaN: not an angle of sampleDisk
        sn = sinf(y), cs = cosf(y);
        return;
    }
    double x = (double)y;
    int n = 0;
    double sign = 1.0;
    bool flip = false; // quadrants 2 and 3: the cosine polynomial changes sign
    if (top < 0x3f4u) { // |y| < pi / 4 (as a float: 0x1.921FB6p-1f)
        if (top < 0x398u) { // |y| < 2^-12
            sn = y, cs = 1.0f;
            return;
        }
    } else {
        const double r = x * kHpiInv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * kHpi;
        sign = (n & 3) == 1 || (n & 3) == 2 ? -1.0 : 1.0; // sign of the sine in the quadrant: {1, -1, -1, 1}
        flip = (n & 2) != 0;
    }
    const double x2 = x * x;
    // the sine polynomial takes x * sign, the cosine polynomial x; both take x * x
    const double xs = x * sign;
    const double x3 = xs * x2, s1 = kS2 + x2 * kS3, x7 = x3 * x2, s = xs + x3 * kS1;
    const float sinPoly = (float)(s + x7 * s1);
    const double c0 = flip ? -kC0 : kC0, c1 = flip ? -kC1 : kC1, c2 = flip ? -kC2 : kC2, c3 = flip ? -kC3 : kC3, c4 = flip ? -kC4 : kC4;
    const double x4 = x2 * x2, cc2 = c3 + x2 * c4, cc1 = c0 + x2 * c1, x6 = x4 * x2, c = cc1 + x4 * c2;
    const float cosPoly = (float)(c + x6 * cc2);
    // sinf evaluates the polynomial of quadrant n, cosf that of quadrant n ^ 1: odd quadrants swap the two
    sn = (n & 1) ? cosPoly : sinPoly;
    cs = (n & 1) ? sinPoly : cosPoly;
}

// ---------------------------------------------------------------------------------------------
// Ray / box: the slab test of the reference's intersect archive (symbol
// intersectRayWithShape(AxisAlignedBox const&, Ray&), declared src/intersect.h:16).
// Returns the reported distance in tOut: the entry distance, or the EXIT distance when the origin is
// inside the box; `outside` tells which.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool box_test(const float lox, const float loy, const float loz, const float hix, const float hiy,
                                         const float hiz, const vec3 o, const vec3 d, const float tmax, float& tOut, bool& outside)
{
    const float tUx = (d.x == 0.0f) ? FLT_MAX : (hix - o.x) / d.x;
    const float tUy = (d.y == 0.0f) ? FLT_MAX : (hiy - o.y) / d.y;
    const float tUz = (d.z == 0.0f) ? FLT_MAX : (hiz - o.z) / d.z;
    const float tLx = (d.x == 0.0f) ? kFltMin : (lox - o.x) / d.x;
    const float tLy = (d.y == 0.0f) ? kFltMin : (loy - o.y) / d.y;
    const float tLz = (d.z == 0.0f) ? kFltMin : (loz - o.z) / d.z;
    float tin = sse_max(sse_min(tUz, tLz), sse_max(sse_min(tUy, tLy), sse_min(tUx, tLx)));
    float tout = sse_min(sse_max(tUz, tLz), sse_min(sse_max(tUy, tLy), sse_max(tUx, tLx)));
    outside = true;
    if (0.0f > tin) {
        if (!(tout > 0.0f)) return false;
        tin = tout;
        tout = FLT_MAX;
        outside = false;
    }
    if (tin > tout) return false;
    if (0.0f > tin) return false;
    if (tin > tmax) return false;
    tOut = tin;
    return true;
}

// Size of the leaf a reference names: the count field, or for the caller's leaves of 63 triangles and more the table entry.
__device__ __forceinline__ uint32_t leaf_count(const SceneView& S, const uint32_t ref)
{
    uint32_t cnt = (ref >> kRefCountShift) & kMaxLeafCount;
    if (cnt == kMaxLeafCount && S.leafCounts) cnt = S.leafCounts[ref & kRefOffsetMask];
    return cnt;
}

struct TraceCounters {
    uint32_t nodePairs, triTests, overflow;
    uint32_t node4; // 4-wide nodes of the library's query tree
};

// ---------------------------------------------------------------------------------------------
// Slab distances without a division per slab.
//
// A ray tests thousands of boxes with the same three divisors.  With r = RN(1/d) computed once
// (IEEE division), RN(a/d) is recovered from q0 = a*r by two residual corrections, each a pair of
// fused multiply-adds: e = fma(-q, d, a) is the exact remainder once q is within an ulp, and
// fma(e, r, q) then rounds the exact quotient correctly (Markstein's theorem for a correctly
// rounded reciprocal).  The result is bit-identical to `a / d` while no step over- or underflows;
// rays or scenes outside that range (see ray_is_fast) take the plain-division test instead.
// rt_selftest_division checks the identity on the GPU over random and edge-case operands.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float div_by(const float a, const float d, const float r)
{
    const float q0 = a * r;
    const float e0 = fmaf(-q0, d, a);
    const float q1 = fmaf(e0, r, q0);
    const float e1 = fmaf(-q1, d, a);
    return fmaf(e1, r, q1);
}
// |x| in {0} U [2^-60, 2^30]: differences of two such numbers are 0 or at least 2^-84 in magnitude
__device__ __forceinline__ bool coord_in_fast_range(const float x)
{
    const float ax = fabsf(x);
    return x == 0.0f || (ax >= 8.6736174e-19f && ax <= 1073741824.0f);
}
__device__ __forceinline__ bool dir_in_fast_range(const float x)
{
    const float ax = fabsf(x);
    return x == 0.0f || (ax >= 9.31322575e-10f && ax <= 1073741824.0f); // [2^-30, 2^30]
}
struct RaySlabs {
    vec3 o, d, r;
    bool fast;
    // first-pass filter of pair_test: slab distance ~ fma(bound, r, nor), off from the exact quotient by less than eps
    bool approx;
    vec3 nor;  // -(o * r)
    float eps;
};
__device__ __forceinline__ RaySlabs make_slabs(const SceneView& S, const vec3 o, const vec3 d)
{
    RaySlabs R;
    R.o = o, R.d = d;
    R.r = make3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    R.fast = S.fastBounds != 0 && coord_in_fast_range(o.x) && coord_in_fast_range(o.y) && coord_in_fast_range(o.z)
        && dir_in_fast_range(d.x) && dir_in_fast_range(d.y) && dir_in_fast_range(d.z);
    R.approx = R.fast && d.x != 0.0f && d.y != 0.0f && d.z != 0.0f;
    R.nor = make3(-(o.x * R.r.x), -(o.y * R.r.y), -(o.z * R.r.z));
    // every box lies inside the root box: |bound| <= extent.  (|bound| + |o|) * |r| * 2^-22 * 1.25 bounds the distance
    // between fma(bound, r, nor) and RN(RN(bound - o) / d); 2^-20 leaves a factor of three.
    const float ex = fmaxf(fabsf(S.rootLo[0]), fabsf(S.rootHi[0])), ey = fmaxf(fabsf(S.rootLo[1]), fabsf(S.rootHi[1])),
                ez = fmaxf(fabsf(S.rootLo[2]), fabsf(S.rootHi[2]));
    R.eps = 9.5367431640625e-7f
        * fmaxf((fabsf(o.x) + ex) * fabsf(R.r.x), fmaxf((fabsf(o.y) + ey) * fabsf(R.r.y), (fabsf(o.z) + ez) * fabsf(R.r.z)));
    return R;
}
// same decision sequence as box_test on precomputed slab distances
__device__ __forceinline__ bool box_decide(const float tUx, const float tUy, const float tUz, const float tLx, const float tLy,
                                           const float tLz, const float tmax, float& tOut, bool& outside)
{
    float tin = sse_max(sse_min(tUz, tLz), sse_max(sse_min(tUy, tLy), sse_min(tUx, tLx)));
    float tout = sse_min(sse_max(tUz, tLz), sse_min(sse_max(tUy, tLy), sse_max(tUx, tLx)));
    const bool inside = 0.0f > tin;
    const bool behind = inside && !(tout > 0.0f);
    tin = inside ? tout : tin;
    tout = inside ? FLT_MAX : tout;
    outside = !inside;
    tOut = tin;
    return !behind && !(tin > tout) && !(0.0f > tin) && !(tin > tmax);
}
__device__ __forceinline__ bool box_test_fast(const float lox, const float loy, const float loz, const float hix, const float hiy,
                                              const float hiz, const RaySlabs& R, const float tmax, float& tOut, bool& outside)
{
    const float qUx = div_by(hix - R.o.x, R.d.x, R.r.x), qUy = div_by(hiy - R.o.y, R.d.y, R.r.y), qUz = div_by(hiz - R.o.z, R.d.z, R.r.z);
    const float qLx = div_by(lox - R.o.x, R.d.x, R.r.x), qLy = div_by(loy - R.o.y, R.d.y, R.r.y), qLz = div_by(loz - R.o.z, R.d.z, R.r.z);
    const bool zx = R.d.x == 0.0f, zy = R.d.y == 0.0f, zz = R.d.z == 0.0f; // the reference substitutes FLT_MAX / FLT_MIN
    return box_decide(zx ? FLT_MAX : qUx, zy ? FLT_MAX : qUy, zz ? FLT_MAX : qUz, zx ? kFltMin : qLx, zy ? kFltMin : qLy,
                      zz ? kFltMin : qLz, tmax, tOut, outside);
}
__device__ __forceinline__ bool box_test_any(const float lox, const float loy, const float loz, const float hix, const float hiy,
                                             const float hiz, const RaySlabs& R, const float tmax, float& tOut, bool& outside)
{
    if (R.fast) return box_test_fast(lox, loy, loz, hix, hiy, hiz, R, tmax, tOut, outside);
    float t = 0.0f;
    const bool h = box_test(lox, loy, loz, hix, hiy, hiz, R.o, R.d, tmax, t, outside);
    tOut = t;
    return h;
}

// The two sibling boxes of one NodePair (a, b = first record; c, e = second).  Same results as two box_test_any
// calls against tmax = FLT_MAX.  The decisions (hit, origin inside, which child is nearer) are first taken on
// cheap approximate slab distances; they are final when every comparison they rest on is decided by more than
// the approximation's error bound, which is the case for all but grazing boxes.  Otherwise the exact slab
// distances are computed.  tl / tr are only ever compared again (culling), with the same bound as margin.
__device__ __forceinline__ void pair_test(const float4 a, const float4 b, const float4 c, const float4 e, const RaySlabs& R, bool& hl, bool& hr,
                                          float& tl, float& tr, bool& outL, bool& outR)
{
    if (R.approx) {
        const float uLx = fmaf(a.w, R.r.x, R.nor.x), uLy = fmaf(b.x, R.r.y, R.nor.y), uLz = fmaf(b.y, R.r.z, R.nor.z);
        const float lLx = fmaf(a.x, R.r.x, R.nor.x), lLy = fmaf(a.y, R.r.y, R.nor.y), lLz = fmaf(a.z, R.r.z, R.nor.z);
        const float uRx = fmaf(c.w, R.r.x, R.nor.x), uRy = fmaf(e.x, R.r.y, R.nor.y), uRz = fmaf(e.y, R.r.z, R.nor.z);
        const float lRx = fmaf(c.x, R.r.x, R.nor.x), lRy = fmaf(c.y, R.r.y, R.nor.y), lRz = fmaf(c.z, R.r.z, R.nor.z);
        const float inL = fmaxf(fmaxf(fminf(uLx, lLx), fminf(uLy, lLy)), fminf(uLz, lLz));
        const float exL = fminf(fminf(fmaxf(uLx, lLx), fmaxf(uLy, lLy)), fmaxf(uLz, lLz));
        const float inR = fmaxf(fmaxf(fminf(uRx, lRx), fminf(uRy, lRy)), fminf(uRz, lRz));
        const float exR = fminf(fminf(fmaxf(uRx, lRx), fmaxf(uRy, lRy)), fmaxf(uRz, lRz));
        const float E = R.eps, E2 = E + E;
        outL = !(0.0f > inL), outR = !(0.0f > inR);
        hl = outL ? !(inL > exL) : (exL > 0.0f);
        hr = outR ? !(inR > exR) : (exR > 0.0f);
        tl = outL ? inL : exL; // a box holding the origin reports its exit distance
        tr = outR ? inR : exR;
        const bool unsure = fabsf(inL) <= E || fabsf(exL) <= E || fabsf(inL - exL) <= E2 || fabsf(inR) <= E || fabsf(exR) <= E
            || fabsf(inR - exR) <= E2 || (hl && hr && fabsf(tl - tr) <= E2);
        if (!unsure) return;
    }
    hl = box_test_any(a.x, a.y, a.z, a.w, b.x, b.y, R, FLT_MAX, tl, outL);
    hr = box_test_any(c.x, c.y, c.z, c.w, e.x, e.y, R, FLT_MAX, tr, outR);
}

// The conservative test of the query tree: a box is reported hit unless the approximate slab distances put the
// ray outside it by more than their error bound.  tl / tr are the approximate entry distances (negative when the
// origin is inside); a caller may skip a box once tl - R.eps exceeds the closest hit.  Only for rays with R.approx.
__device__ __forceinline__ void pair_test_loose(const float4 a, const float4 b, const float4 c, const float4 e, const RaySlabs& R, bool& hl,
                                                bool& hr, float& tl, float& tr)
{
    const float uLx = fmaf(a.w, R.r.x, R.nor.x), uLy = fmaf(b.x, R.r.y, R.nor.y), uLz = fmaf(b.y, R.r.z, R.nor.z);
    const float lLx = fmaf(a.x, R.r.x, R.nor.x), lLy = fmaf(a.y, R.r.y, R.nor.y), lLz = fmaf(a.z, R.r.z, R.nor.z);
    const float uRx = fmaf(c.w, R.r.x, R.nor.x), uRy = fmaf(e.x, R.r.y, R.nor.y), uRz = fmaf(e.y, R.r.z, R.nor.z);
    const float lRx = fmaf(c.x, R.r.x, R.nor.x), lRy = fmaf(c.y, R.r.y, R.nor.y), lRz = fmaf(c.z, R.r.z, R.nor.z);
    tl = fmaxf(fmaxf(fminf(uLx, lLx), fminf(uLy, lLy)), fminf(uLz, lLz));
    tr = fmaxf(fmaxf(fminf(uRx, lRx), fminf(uRy, lRy)), fminf(uRz, lRz));
    const float exL = fminf(fminf(fmaxf(uLx, lLx), fmaxf(uLy, lLy)), fmaxf(uLz, lLz));
    const float exR = fminf(fminf(fmaxf(uRx, lRx), fmaxf(uRy, lRy)), fmaxf(uRz, lRz));
    const float E = R.eps, E2 = E + E;
    hl = !(tl > exL + E2) && !(exL < -E);
    hr = !(tr > exR + E2) && !(exR < -E);
}

// The conservative test of the query tree, for the four children of one Node4: a box is reported hit unless the
// approximate slab distances put the ray outside it by more than their error bound, or its entry lies beyond `cullT`
// by more than that bound.  Result: the children sorted by entry distance, nearest first, as keys = bits of
// max(entry, 0) with the two lowest bits replaced by the child's slot; kMissKey for the others (they sort last).
// Only for rays with R.approx.
constexpr uint32_t kMissKey = 0xffffffffu;
struct Node4Hits {
    uint32_t key[4]; // ascending
    uint32_t ref[4]; // by slot, not sorted: ref[key & 3]
};
__device__ __forceinline__ uint32_t node4_ref(const Node4Hits& h, const uint32_t key)
{
    const uint32_t k = key & 3u;
    return k == 0 ? h.ref[0] : (k == 1 ? h.ref[1] : (k == 2 ? h.ref[2] : h.ref[3]));
}
__device__ __forceinline__ float node4_entry(const uint32_t key) { return __uint_as_float(key & ~3u); } // rounded towards zero
__device__ __forceinline__ Node4Hits node4_test(const Node4* node, const RaySlabs& R, const float cullT)
{
    const float4* q = reinterpret_cast<const float4*>(node);
    const float4 lox = q[0], loy = q[1], loz = q[2], hix = q[3], hiy = q[4], hiz = q[5];
    const float4 rf = q[6];
    Node4Hits h;
    h.ref[0] = __float_as_uint(rf.x), h.ref[1] = __float_as_uint(rf.y), h.ref[2] = __float_as_uint(rf.z), h.ref[3] = __float_as_uint(rf.w);
    const float E = R.eps, E2 = E + E;
    const float lx[4] = { lox.x, lox.y, lox.z, lox.w }, ly[4] = { loy.x, loy.y, loy.z, loy.w }, lz[4] = { loz.x, loz.y, loz.z, loz.w };
    const float ux[4] = { hix.x, hix.y, hix.z, hix.w }, uy[4] = { hiy.x, hiy.y, hiy.z, hiy.w }, uz[4] = { hiz.x, hiz.y, hiz.z, hiz.w };
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float ax = fmaf(lx[k], R.r.x, R.nor.x), bx = fmaf(ux[k], R.r.x, R.nor.x);
        const float ay = fmaf(ly[k], R.r.y, R.nor.y), by = fmaf(uy[k], R.r.y, R.nor.y);
        const float az = fmaf(lz[k], R.r.z, R.nor.z), bz = fmaf(uz[k], R.r.z, R.nor.z);
        const float tin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float tex = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const bool hit = h.ref[k] != kEmptyRef && !(tin > tex + E2) && !(tex < -E) && !(tin - E > cullT);
        h.key[k] = hit ? ((__float_as_uint(fmaxf(tin, 0.0f)) & ~3u) | (uint32_t)k) : kMissKey;
    }
    // sorting network for four
#define RT_CSWAP(a, b)                                   \
    do {                                                 \
        const uint32_t lo_ = min(h.key[a], h.key[b]);    \
        h.key[b] = max(h.key[a], h.key[b]);              \
        h.key[a] = lo_;                                  \
    } while (0)
    RT_CSWAP(0, 1);
    RT_CSWAP(2, 3);
    RT_CSWAP(0, 2);
    RT_CSWAP(1, 3);
    RT_CSWAP(1, 2);
#undef RT_CSWAP
    return h;
}

// Does the reference's walk reach the leaf with this box?  True only when that is certain: the leaf's box and,
// because the reference's boxes are nested, every box above it then pass the reference's test (box_test) whatever
// the rounding of its slab distances.  The slab intervals of an enclosing box contain the leaf's; the test fails only
// if the entry distance of one axis exceeds the exit distance of ANOTHER axis (on one axis min <= max holds in any
// arithmetic) or if a box lies behind the origin, so it is enough that the leaf's approximate distances keep those
// comparisons clear by a margin above the approximation's and the division's error (bounded per axis as R.eps is
// for all three, see make_slabs).  Only for rays with R.approx.
__device__ __forceinline__ bool leaf_surely_visited(const SceneView& S, const TriCert& c, const RaySlabs& R)
{
    const float ux = fmaf(c.leafHi[0], R.r.x, R.nor.x), uy = fmaf(c.leafHi[1], R.r.y, R.nor.y), uz = fmaf(c.leafHi[2], R.r.z, R.nor.z);
    const float lx = fmaf(c.leafLo[0], R.r.x, R.nor.x), ly = fmaf(c.leafLo[1], R.r.y, R.nor.y), lz = fmaf(c.leafLo[2], R.r.z, R.nor.z);
    const float inx = fminf(ux, lx), iny = fminf(uy, ly), inz = fminf(uz, lz);
    const float exx = fmaxf(ux, lx), exy = fmaxf(uy, ly), exz = fmaxf(uz, lz);
    const float k = 2.0f * 9.5367431640625e-7f;
    const float mx = k * (fabsf(R.o.x) + fmaxf(fabsf(S.rootLo[0]), fabsf(S.rootHi[0]))) * fabsf(R.r.x);
    const float my = k * (fabsf(R.o.y) + fmaxf(fabsf(S.rootLo[1]), fabsf(S.rootHi[1]))) * fabsf(R.r.y);
    const float mz = k * (fabsf(R.o.z) + fmaxf(fabsf(S.rootLo[2]), fabsf(S.rootHi[2]))) * fabsf(R.r.z);
    return exx > mx && exy > my && exz > mz && inx + (mx + my) < exy && inx + (mx + mz) < exz && iny + (my + mx) < exx
        && iny + (my + mz) < exz && inz + (mz + mx) < exx && inz + (mz + my) < exy;
}

// The same question answered by the reference's own tests: every box on the way from the root's children down to
// the leaf (the root box itself is tested when a query starts).
__device__ __forceinline__ bool leaf_visited_exactly(const SceneView& S, const uint32_t route, const RaySlabs& R)
{
    uint32_t ref = S.refRootRef;
    for (int k = 30 - __clz(route); k >= 0; k--) {
        if (ref & kLeafBit) return false;
        const float4* p = reinterpret_cast<const float4*>(S.refPairs + ref);
        const bool second = (route >> k) & 1u;
        const float4 a = second ? p[2] : p[0], b = second ? p[3] : p[1];
        float t;
        bool outside;
        if (!box_test_any(a.x, a.y, a.z, a.w, b.x, b.y, R, FLT_MAX, t, outside)) return false;
        ref = __float_as_uint(b.z);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Ray / triangle with hoisted per-triangle constants (archive symbols trianglePlane,
// intersectRayWithTriangle, pointInTriangle; declared src/intersect.h:5-12).
// ---------------------------------------------------------------------------------------------
// TIES: tieT records the distance at which a second triangle was accepted at exactly the distance of the one held.
template <bool TIES = false>
__device__ __forceinline__ void tri_test_plane(const SceneView& S, const uint32_t i, const float4 pl, const vec3 o, const vec3 d, float& t,
                                               uint32_t& tri, float* tieT = nullptr)
{
    const vec3 n = make3(pl.x, pl.y, pl.z);
    const float tt = (pl.w - dot(o, n)) / dot(d, n);
    if (!(tt >= 0.0f)) return;
    if (!(t >= tt)) return; // equal distance accepted: the later candidate wins the tie
    const float4* e = reinterpret_cast<const float4*>(S.triEdges + i);
    const float4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4];
    const vec3 p = d * tt + o;
    const vec3 v0 = make3(e0.x, e0.y, e0.z), c0 = make3(e0.w, e1.x, e1.y);
    const vec3 v1 = make3(e1.z, e1.w, e2.x), c1 = make3(e2.y, e2.z, e2.w);
    const vec3 v2 = make3(e3.x, e3.y, e3.z), c2 = make3(e3.w, e4.x, e4.y);
    const bool in0 = dot(c0, p - v0) >= 0.0f, in1 = dot(c1, p - v1) >= 0.0f, in2 = dot(c2, p - v2) >= 0.0f;
    if (in0 && in1 && in2) {
        if (TIES && tt == t && tri != kNoHit) *tieT = tt;
        t = tt;
        tri = i;
    }
}
__device__ __forceinline__ void tri_test(const SceneView& S, const uint32_t i, const vec3 o, const vec3 d, float& t, uint32_t& tri)
{
    tri_test_plane(S, i, reinterpret_cast<const float4*>(S.triPlane)[i], o, d, t, tri);
}

// The triangles [first, first + cnt) of one leaf in storage order.  Memory latency, not arithmetic,
// bounds a traversal step, so the work is arranged as two round trips per four triangles: the four
// planes together, then the edge records of every triangle whose plane distance is still in range
// together.  Acceptance then runs in storage order with the shrinking t, as the reference's loop does.
struct EdgeRec {
    float4 e0, e1, e2, e3, e4;
};
__device__ __forceinline__ EdgeRec load_edges(const SceneView& S, const uint32_t i)
{
    const float4* e = reinterpret_cast<const float4*>(S.triEdges + i);
    return EdgeRec { e[0], e[1], e[2], e[3], e[4] };
}
__device__ __forceinline__ bool point_in_triangle(const EdgeRec& E, const vec3 p)
{
    const vec3 v0 = make3(E.e0.x, E.e0.y, E.e0.z), c0 = make3(E.e0.w, E.e1.x, E.e1.y);
    const vec3 v1 = make3(E.e1.z, E.e1.w, E.e2.x), c1 = make3(E.e2.y, E.e2.z, E.e2.w);
    const vec3 v2 = make3(E.e3.x, E.e3.y, E.e3.z), c2 = make3(E.e3.w, E.e4.x, E.e4.y);
    const bool in0 = dot(c0, p - v0) >= 0.0f, in1 = dot(c1, p - v1) >= 0.0f, in2 = dot(c2, p - v2) >= 0.0f;
    return in0 && in1 && in2;
}
template <bool COUNT, bool TIES = false>
__device__ __forceinline__ void leaf_test(const SceneView& S, const uint32_t first, const uint32_t cnt, const vec3 o, const vec3 d, float& t,
                                          uint32_t& tri, TraceCounters& tc, float* tieT = nullptr)
{
    const float4* planes = reinterpret_cast<const float4*>(S.triPlane) + first;
    for (uint32_t k = 0; k < cnt; k += 4) {
        const uint32_t m = cnt - k;
        const float4 p0 = planes[k];
        const float4 p1 = planes[k + (m > 1 ? 1 : 0)];
        const float4 p2 = planes[k + (m > 2 ? 2 : 0)];
        const float4 p3 = planes[k + (m > 3 ? 3 : 0)];
        if (COUNT) tc.triTests += (m > 4 ? 4 : m);
        const float t0 = (p0.w - dot(o, make3(p0.x, p0.y, p0.z))) / dot(d, make3(p0.x, p0.y, p0.z));
        const float t1 = (p1.w - dot(o, make3(p1.x, p1.y, p1.z))) / dot(d, make3(p1.x, p1.y, p1.z));
        const float t2 = (p2.w - dot(o, make3(p2.x, p2.y, p2.z))) / dot(d, make3(p2.x, p2.y, p2.z));
        const float t3 = (p3.w - dot(o, make3(p3.x, p3.y, p3.z))) / dot(d, make3(p3.x, p3.y, p3.z));
        // in range against the t of the batch's start: necessary for acceptance, since t only shrinks
        const bool c0 = t0 >= 0.0f && t >= t0;
        const bool c1 = m > 1 && t1 >= 0.0f && t >= t1;
        const bool c2 = m > 2 && t2 >= 0.0f && t >= t2;
        const bool c3 = m > 3 && t3 >= 0.0f && t >= t3;
        if (!(c0 || c1 || c2 || c3)) continue;
        const uint32_t b = first + k;
        // two edge records in flight at a time (a triangle out of range re-reads the pair's first record: no branch
        // between the loads, no new cache line); equal distance accepted: the later candidate wins the tie
        if (c0 || c1) {
            const EdgeRec E0 = load_edges(S, b);
            const EdgeRec E1 = load_edges(S, c1 ? b + 1 : b);
            const bool i0 = point_in_triangle(E0, d * t0 + o), i1 = point_in_triangle(E1, d * t1 + o);
            if (c0 && i0) {
                if (TIES && t0 == t && tri != kNoHit) *tieT = t0;
                t = t0, tri = b;
            }
            if (c1 && i1 && t >= t1) {
                if (TIES && t1 == t && tri != kNoHit) *tieT = t1;
                t = t1, tri = b + 1;
            }
        }
        if (c2 || c3) {
            const EdgeRec E2 = load_edges(S, c2 ? b + 2 : b + 3);
            const EdgeRec E3 = load_edges(S, c3 ? b + 3 : b + 2);
            const bool i2 = point_in_triangle(E2, d * t2 + o), i3 = point_in_triangle(E3, d * t3 + o);
            if (c2 && i2 && t >= t2) {
                if (TIES && t2 == t && tri != kNoHit) *tieT = t2;
                t = t2, tri = b + 2;
            }
            if (c3 && i3 && t >= t3) {
                if (TIES && t3 == t && tri != kNoHit) *tieT = t3;
                t = t3, tri = b + 3;
            }
        }
    }
}

// The same leaf with one triangle in flight at a time: fewer live registers, for kernels that rely on
// many resident waves rather than on short steps.
template <bool COUNT, bool TIES = false>
__device__ __forceinline__ void leaf_test_narrow(const SceneView& S, const uint32_t first, const uint32_t cnt, const vec3 o, const vec3 d, float& t,
                                                 uint32_t& tri, TraceCounters& tc, float* tieT = nullptr)
{
    const float4* planes = reinterpret_cast<const float4*>(S.triPlane) + first;
    for (uint32_t k = 0; k < cnt; k += 4) {
        const uint32_t m = cnt - k; // planes of up to four triangles in flight together
        const float4 p0 = planes[k];
        const float4 p1 = planes[k + (m > 1 ? 1 : 0)];
        const float4 p2 = planes[k + (m > 2 ? 2 : 0)];
        const float4 p3 = planes[k + (m > 3 ? 3 : 0)];
        if (COUNT) tc.triTests += (m > 4 ? 4 : m);
        tri_test_plane<TIES>(S, first + k, p0, o, d, t, tri, tieT);
        if (m > 1) tri_test_plane<TIES>(S, first + k + 1, p1, o, d, t, tri, tieT);
        if (m > 2) tri_test_plane<TIES>(S, first + k + 2, p2, o, d, t, tri, tieT);
        if (m > 3) tri_test_plane<TIES>(S, first + k + 3, p3, o, d, t, tri, tieT);
    }
}

// Ray / sphere (archive symbol intersectRayWithShape(Sphere const&, Ray&, HitInfo&), src/intersect.h:14)
__device__ __forceinline__ bool sphere_test(const GpuSphere& sp, const vec3 o, const vec3 d, float& t)
{
    const vec3 L = o - load3(sp.center);
    const float h = dot(d, L);
    const float b = h + h;
    const float cc = dot(L, L) - sp.radius * sp.radius;
    const float disc = b * b - 4.0f * cc;
    if (disc < 0.0f) return false;
    float t0, t1;
    if (disc == 0.0f) {
        t0 = t1 = 0.5f * (-b);
    } else {
        const float s = sqrtf(disc);
        t0 = (-b - s) * 0.5f;
        t1 = (s - b) * 0.5f;
    }
    float tt = FLT_MAX;
    if (t0 >= 0.0f && t0 < tt) tt = t0;
    if (t1 >= 0.0f && t1 < tt) tt = t1;
    if (tt < 0.0f || tt >= t) return false;
    t = tt;
    return true;
}


// ---------------------------------------------------------------------------------------------
// Closest hit (reference intersectRayWithBVH, src/bvh.cpp:235-370).  Near child first, ties to the
// right child, leaves tested in storage order, `t >= candidate` accepted.  With CULL a child box
// whose entry lies beyond the closest hit so far is skipped (the reference visits it and finds
// nothing closer); without CULL the visit order is the reference's exactly.
// Lanes descend through inner nodes together and test leaves together (two nested loops), which
// keeps a wave's lanes on the same code far more often than one loop alternating between both.
// `stack` points at this lane's column of the LDS stack (stride kBlock).
// ---------------------------------------------------------------------------------------------
template <bool CULL, bool COUNT>
__device__ __forceinline__ void trace(const SceneView& S, const bool useBvh, const vec3 o, const vec3 d, float& t, uint32_t& tri,
                                      uint32_t* stack, TraceCounters& tc)
{
    tri = kNoHit;
    if (useBvh) {
        const RaySlabs R = make_slabs(S, o, d);
        float tr;
        bool outside;
        // root box against the ray's own t; a miss returns before the sphere loop (src/bvh.cpp:264-268)
        if (!box_test_any(S.rootLo[0], S.rootLo[1], S.rootLo[2], S.rootHi[0], S.rootHi[1], S.rootHi[2], R, t, tr, outside)) return;
        int sp = 0;
        uint32_t cur = S.rootRef;
        for (;;) {
            while (!(cur & kLeafBit)) {
                const float4* q = reinterpret_cast<const float4*>(S.pairs + cur);
                const float4 a = q[0], b = q[1], c = q[2], e = q[3];
                if (COUNT) tc.nodePairs++;
                float tl, trr;
                bool outL, outR, hl, hr;
                pair_test(a, b, c, e, R, hl, hr, tl, trr, outL, outR);
                const uint32_t ref0 = __float_as_uint(b.z), ref1 = __float_as_uint(e.z);
                const bool leftNear = tl < trr; // only used when both boxes are hit; equal: right first (src/bvh.cpp:301-310)
                if (CULL) { // the distances may be the approximate ones: cull only beyond their error bound
                    hl = hl && !(outL && tl - R.eps > t);
                    hr = hr && !(outR && trr - R.eps > t);
                }
                if (hl && hr) {
                    const uint32_t nearRef = leftNear ? ref0 : ref1, farRef = leftNear ? ref1 : ref0;
                    if (sp < kStackDepth) stack[sp * kBlock] = farRef;
                    else tc.overflow++;
                    sp++;
                    cur = nearRef;
                } else if (hl) {
                    cur = ref0;
                } else if (hr) {
                    cur = ref1;
                } else if (sp == 0) {
                    cur = kTraversalDone;
                } else {
                    sp--;
                    cur = (sp < kStackDepth) ? stack[sp * kBlock] : kLeafBit; // overflowed entries are dropped (counted)
                }
            }
            if (cur == kTraversalDone) break;
            const uint32_t first = cur & kRefOffsetMask, cnt = leaf_count(S, cur);
            leaf_test_narrow<COUNT>(S, first, cnt, o, d, t, tri, tc);
            if (sp == 0) break;
            sp--;
            cur = (sp < kStackDepth) ? stack[sp * kBlock] : kLeafBit;
        }
    } else {
        // enableAccelStructure off (src/bvh.cpp:351-363): every stored triangle, array order
        for (uint32_t k = 0; k < S.nTris; k++) {
            if (COUNT) tc.triTests++;
            tri_test(S, S.bruteOrder[k], o, d, t, tri);
        }
    }
    for (uint32_t s = 0; s < S.nSpheres; s++) // src/bvh.cpp:366-367
        if (sphere_test(S.spheres[s], o, d, t)) tri = kSphereBase + s;
}

// ---------------------------------------------------------------------------------------------
// Hit record (updateHitInfo src/bvh.cpp:18-44, computeBarycentricCoord / interpolateNormal
// src/interpolate.cpp:13-51).
// ---------------------------------------------------------------------------------------------
struct Hit {
    vec3 normal;
    vec3 bary;
    vec2 uv;
    uint32_t mesh;   // material index (meshID), or sphere index with isSphere
    uint32_t primId; // canonical id
    bool isSphere;
};

__device__ __forceinline__ void fill_hit(const SceneView& S, const rt_features& F, const uint32_t tri, const vec3 o, const vec3 d,
                                         const float t, Hit& h)
{
    if (tri >= kSphereBase) {
        const GpuSphere& sp = S.spheres[tri - kSphereBase];
        h.normal = normalize((d * t + o) - load3(sp.center));
        h.bary = splat(0.0f);
        h.uv = vec2 { 0.0f, 0.0f };
        h.mesh = tri - kSphereBase;
        h.primId = 0xfffffffeu;
        h.isSphere = true;
        return;
    }
    h.isSphere = false;
    const float4 pl = reinterpret_cast<const float4*>(S.triPlane)[tri];
    const vec3 n = make3(pl.x, pl.y, pl.z);
    const float4* e = reinterpret_cast<const float4*>(S.triEdges + tri);
    const float4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
    const vec3 v0 = make3(e0.x, e0.y, e0.z), v1 = make3(e1.z, e1.w, e2.x), v2 = make3(e3.x, e3.y, e3.z);
    const float4* aq = reinterpret_cast<const float4*>(S.triAttr + tri);
    const float4 a0 = aq[0], a1 = aq[1], a2 = aq[2], a3 = aq[3], a4 = aq[4];
    h.primId = __float_as_uint(a0.x);
    h.mesh = __float_as_uint(a0.y);
    const float d00 = a0.z, d01 = a0.w, d11 = a1.x, denom = a1.y;
    const vec3 p = o + t * d;
    const vec3 first = v1 - v0, second = v2 - v0, point = p - v0;
    const float d0p = dot(first, point), d1p = dot(second, point);
    const float beta = (d11 * d0p - d01 * d1p) / denom;
    const float gamma = (d00 * d1p - d01 * d0p) / denom;
    h.bary = make3(1.f - beta - gamma, beta, gamma);
    h.normal = n;
    if (F.enableNormalInterp) {
        const vec3 n0 = make3(a1.z, a1.w, a2.x), n1 = make3(a2.y, a2.z, a2.w), n2 = make3(a3.x, a3.y, a3.z);
        h.normal = h.bary.x * n0 + h.bary.y * n1 + h.bary.z * n2;
    }
    if (F.enableTextureMapping) {
        const float4 a5 = aq[5];
        const vec2 t0 { a3.w, a4.x }, t1 { a4.y, a4.z }, t2 { a4.w, a5.x };
        h.uv = h.bary.x * t0 + h.bary.y * t1 + h.bary.z * t2;
    } else {
        h.uv = vec2 { 0.0f, 0.0f };
    }
    if (dot(d, n) > 0.0f) h.normal = -h.normal;
}

struct Mat {
    vec3 kd, ks;
    float shininess, transparency;
};
__device__ __forceinline__ Mat load_material(const SceneView& S, const uint32_t mesh, const bool isSphere)
{
    Mat m;
    if (isSphere) {
        const GpuSphere& sp = S.spheres[mesh];
        m.kd = load3(sp.kd);
        m.ks = load3(sp.ks);
        m.shininess = sp.shininess;
        m.transparency = sp.transparency;
    } else {
        const float4* q = reinterpret_cast<const float4*>(S.materials + mesh);
        const float4 a = q[0], b = q[1];
        m.kd = make3(a.x, a.y, a.z);
        m.shininess = a.w;
        m.ks = make3(b.x, b.y, b.z);
        m.transparency = b.w;
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// Textures (src/texture.cpp:5-114) and sampleMaterialKd (src/shading.cpp:13-24)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ vec3 texel(const GpuTexture& T, int x, int y) // getPixelFromCoordinates: clamped, rows flipped
{
    x = x < 0 ? 0 : (x > T.width - 1 ? T.width - 1 : x);
    y = y < 0 ? 0 : (y > T.height - 1 ? T.height - 1 : y);
    const float* q = T.pixels + 3 * ((size_t)(T.height - y - 1) * T.width + x);
    return make3(q[0], q[1], q[2]);
}
__device__ __forceinline__ vec3 texture_nearest(const GpuTexture& T, const vec2 uv)
{
    return texel(T, (int)floorf(uv.x * (float)T.width), (int)floorf(uv.y * (float)T.height));
}
__device__ __forceinline__ vec3 texture_bilinear(const GpuTexture& T, const vec2 uv)
{
    int xTL = (int)floorf(uv.x * (float)T.width);
    if (xTL >= T.width) xTL = T.width - 1;
    else if (uv.x * (float)T.width - (float)xTL < 0.5f) --xTL;
    if (xTL < 0) xTL = 0;
    int xTR = xTL + 1;
    if (xTR >= T.width) xTR = xTL;
    if (xTR < 0) xTR = 0;
    int yTL = (int)floorf(uv.y * (float)T.height);
    if (yTL >= T.height) yTL = T.height - 1;
    else if (uv.y * (float)T.height - (float)yTL < 0.5f) --yTL;
    if (yTL < 0) yTL = 0;
    int yBL = yTL + 1;
    if (yBL >= T.height) yBL = T.height - 1;
    float xc = uv.x * (float)T.width;
    if (xc >= (float)T.width) xc = (float)(T.width - 1);
    float yc = uv.y * (float)T.height;
    if (yc >= (float)T.height) yc = (float)(T.height - 1);
    const float alpha = xc - (float)xTL - 0.5f, beta = yc - (float)yTL - 0.5f;
    const float cTL = (1.f - alpha) * (1.f - beta), cTR = alpha * (1.f - beta), cBL = (1.f - alpha) * beta, cBR = alpha * beta;
    vec3 r = cTL * texel(T, xTL, yTL);
    r = r + cTR * texel(T, xTR, yTL);
    r = r + cBL * texel(T, xTL, yBL);
    r = r + cBR * texel(T, xTR, yBL);
    return r;
}
// kd of a triangle hit of mesh `mesh` with interpolated texture coordinates uv
__device__ __forceinline__ vec3 sample_material_kd(const SceneView& S, const rt_features& F, const uint32_t mesh, const vec2 uv, const vec3 kd)
{
    if (!F.enableTextureMapping) return kd;
    const GpuTexture T = S.textures[mesh];
    if (!T.pixels) return kd;
    return F.enableBilinearTextureFiltering ? texture_bilinear(T, uv) : texture_nearest(T, uv);
}
// interpolateTexCoord at point p of triangle `tri` (updateHitInfo, src/bvh.cpp:18-38): the same operations as fill_hit
__device__ __forceinline__ vec2 tri_uv_at(const SceneView& S, const uint32_t tri, const vec3 p)
{
    const float4* e = reinterpret_cast<const float4*>(S.triEdges + tri);
    const float4 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
    const vec3 v0 = make3(e0.x, e0.y, e0.z), v1 = make3(e1.z, e1.w, e2.x), v2 = make3(e3.x, e3.y, e3.z);
    const float4* aq = reinterpret_cast<const float4*>(S.triAttr + tri);
    const float4 a0 = aq[0], a1 = aq[1], a3 = aq[3], a4 = aq[4], a5 = aq[5];
    const float d00 = a0.z, d01 = a0.w, d11 = a1.x, denom = a1.y;
    const vec3 first = v1 - v0, second = v2 - v0, point = p - v0;
    const float d0p = dot(first, point), d1p = dot(second, point);
    const float beta = (d11 * d0p - d01 * d1p) / denom;
    const float gamma = (d00 * d1p - d01 * d0p) / denom;
    const vec3 bary = make3(1.f - beta - gamma, beta, gamma);
    const vec2 t0 { a3.w, a4.x }, t1 { a4.y, a4.z }, t2 { a4.w, a5.x };
    return bary.x * t0 + bary.y * t1 + bary.z * t2;
}

// ---------------------------------------------------------------------------------------------
// Shading models (src/shading.cpp:29-233); m.kd is what sampleMaterialKd returns for the hit
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ vec3 lambert(const rt_features& F, const vec3 lightDirection, const vec3 lightColor, const vec3 normal, const Mat& m)
{
    const vec3 l = normalize(lightDirection), n = normalize(normal);
    float dt = dot(l, n);
    if (F.enableTransparency && m.transparency < 1.0f && dt < 0) dt = -dt; // lit from behind through the surface (src/shading.cpp:67-69)
    if (dt <= 0) return splat(0.0f);
    return m.kd * lightColor * dt;
}
__device__ vec3 gradient_sample(const float ti) // LinearGradient::sample over the hard-coded gradient (src/shading.cpp:32-40,185-211)
{
    const float ts[5] = { 0.1f, 0.22f, 0.5f, 0.78f, 0.9f };
    const vec3 cs[5] = { make3(215.f / 256.f, 210.f / 256.f, 203.f / 256.f), make3(250.f / 256.f, 250.f / 256.f, 240.f / 256.f),
                         make3(145.f / 256.f, 170.f / 256.f, 175.f / 256.f), make3(255.f / 256.f, 250.f / 256.f, 205.f / 256.f),
                         make3(170.f / 256.f, 170.f / 256.f, 170.f / 256.f) };
    if (ts[0] > ti) return cs[0];
    for (int i = 0; i < 4; i++) {
        if (fabsf(ts[i] - ti) <= 1e-6f) return cs[i];
        if (ts[i] < ti && ts[i + 1] > ti) {
            const float alpha = (ti - ts[i]) / (ts[i + 1] - ts[i]);
            return (1.0f - alpha) * cs[i] + alpha * cs[i + 1];
        }
    }
    return cs[4];
}
__device__ vec3 shade(const rt_features& F, const vec3 cameraDirection, const vec3 lightDirection, const vec3 lightColor,
                      const vec3 normal, const Mat& m)
{
    if (!F.enableShading) return lightColor * m.kd;
    switch (F.shadingModel) {
    case 0: return lambert(F, lightDirection, lightColor, normal, m);
    case 1: { // computePhongModel, src/shading.cpp:92-123
        const vec3 diffuse = lambert(F, lightDirection, lightColor, normal, m);
        const vec3 d = normalize(-lightDirection), n = normalize(normal);
        float dt = dot(-d, n);
        if (F.enableTransparency && m.transparency < 1.0f && dt < 0) dt = -dt; // src/shading.cpp:103-105
        if (dt <= 0) return diffuse;
        const vec3 v = normalize(cameraDirection);
        const vec3 r = normalize(d - 2.0f * dot(d, n) * n);
        const float VdotR = dot(v, r);
        if (VdotR <= 0) return diffuse;
        const float pw = powf(VdotR, m.shininess);
        return diffuse + pw * m.ks * lightColor;
    }
    case 2: { // computeBlinnPhongModel, src/shading.cpp:140-169
        const vec3 diffuse = lambert(F, lightDirection, lightColor, normal, m);
        const vec3 d = normalize(lightDirection), n = normalize(normal);
        const float dt = dot(n, d);
        if (dt <= 0) return diffuse;
        const vec3 hv = normalize(d + cameraDirection);
        const float NdotH = dot(n, hv);
        if (NdotH <= 0) return diffuse;
        const float pw = powf(NdotH, m.shininess);
        return diffuse + pw * m.ks * lightColor;
    }
    default: { // computeLinearGradientModel, src/shading.cpp:228-233
        const float c = dot(normalize(lightDirection), normalize(normal));
        return lightColor * gradient_sample(c);
    }
    }
}

// constructOrthonormalBasis (src/extra.cpp:418-441)
__device__ __forceinline__ void ortho_basis(const vec3 r, vec3& w, vec3& u, vec3& v)
{
    w = normalize(r);
    const float ax = fabsf(w.x), ay = fabsf(w.y), az = fabsf(w.z);
    vec3 t;
    if (ax < ay && ax < az) t = make3(1.0f, w.y, w.z);
    else if (ay < ax && ay < az) t = make3(w.x, 1.0f, w.z);
    else t = make3(w.x, w.y, 1.0f);
    u = normalize(cross(t, w));
    v = cross(w, u);
}

// Trackball::generateRay (framework/src/trackball.cpp:101-110) with the quaternion precomputed
__device__ __forceinline__ void camera_ray(const rt_camera& cam, const float ndcX, const float ndcY, vec3& o, vec3& d)
{
    const vec3 cs = normalize(make3(-ndcX * cam.halfWidth, ndcY * cam.halfHeight, 1.0f));
    o = load3(cam.position);
    d = quat_rotate(cam.quat, cs);
}

// ---------------------------------------------------------------------------------------------
// The render kernel: one lane = one pixel at a time, pixels fetched from a shared counter.
//
// renderRay's recursion (src/recursive.cpp:25-83) is run as an explicit depth-first state machine
// so that each pixel's sampler is consumed in exactly the reference's order (glossy samples draw
// two numbers each, in recursion order, src/extra.cpp:489-505).  Every lane funnels all of its
// closest-hit queries - primary, shadow, reflection - through the single trace() call in the loop
// body, so a wave keeps traversing together whatever mix of ray kinds its lanes hold.
// ---------------------------------------------------------------------------------------------
enum LaneState : int { ST_NEW = 0, ST_EXIT, ST_TRACE, ST_HIT_RAD, ST_HIT_SHD, ST_LIGHTS, ST_SPAWN, ST_GLOSSY_NEXT, ST_PASS, ST_RETURN };
enum ChildKind : int { CK_SPEC = 0, CK_GLOSSY = 1, CK_PASS = 2 };

template <bool CULL, bool COUNT, bool PERSISTENT>
__global__ __launch_bounds__(kBlock) void render_kernel(const RenderParams P)
{
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    uint32_t* const stack = s_stack + threadIdx.x;
    const SceneView& S = P.S;
    const rt_features& F = P.F;
    const int lane = threadIdx.x & 63;

    // saved state of the callers of the ray being evaluated, indexed by their rayDepth
    float frames[RT_MAX_RAY_DEPTH * kFrameFloats];

    // pixel
    uint32_t px = 0, py = 0, smp = 0, seed = 0;
    uint32_t sampleIdx = 0, nPixelRays = 1;
    vec3 Lsum = splat(0.0f);
    // ray being traced
    vec3 to = splat(0.0f), td = splat(0.0f);
    float tt = 0.0f;
    uint32_t ttri = kNoHit;
    bool traceIsShadow = false;
    // radiance ray + hit under evaluation
    int depth = 0;
    vec3 ro = splat(0.0f), rd = splat(0.0f);
    float rt = 0.0f;
    vec3 hn = splat(0.0f);
    vec2 huv { 0.0f, 0.0f };
    uint32_t hmesh = 0;
    bool hsphere = false;
    vec3 Lo = splat(0.0f);
    // light loop
    uint32_t lightIdx = 0, lightSample = 0;
    vec3 areaAcc = splat(0.0f), lpos = splat(0.0f), lcol = splat(0.0f);
    vec3 shIncoming = splat(0.0f), shPrev = splat(0.0f); // light through transparent occluders (src/light.cpp:107-136)
    uint32_t shSegments = 0;
    vec3 ret = splat(0.0f);
    // counters
    uint32_t cPrimary = 0, cShadow = 0, cSecondary = 0, cPixels = 0;
    TraceCounters tc { 0, 0, 0 };

    int state = ST_NEW;
    uint32_t staticNext = blockIdx.x * kBlock + threadIdx.x;

    for (;;) {
        // ---- pixel fetch (wave-convergent) ----
        if (__ballot(state == ST_NEW) != 0ull) {
            uint32_t w = 0xffffffffu;
            if (PERSISTENT) {
                const bool need = (state == ST_NEW);
                const unsigned long long mask = __ballot(need);
                const int leader = __ffsll((long long)mask) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(P.work, (uint32_t)__popcll(mask));
                base = __shfl(base, leader);
                if (need) w = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            } else if (state == ST_NEW) {
                w = staticNext;
                staticNext += gridDim.x * kBlock;
            }
            if (state == ST_NEW) {
                if (w >= P.nWork) {
                    state = ST_EXIT;
                } else {
                    const uint32_t tile = w >> 6, l = w & 63;
                    const uint32_t m = tile / P.tilesX, tx = tile - m * P.tilesX;
                    const uint32_t blk = m / P.stripeTileRows, r = m - blk * P.stripeTileRows;
                    const uint32_t ty = (blk * P.world + P.rank) * P.stripeTileRows + r;
                    px = tx * 8 + (l & 7);
                    py = ty * 8 + (l >> 3);
                    if (px < (uint32_t)P.W && py < (uint32_t)P.H) {
                        cPixels++;
                        seed = static_cast<uint32_t>(P.H * (int)px + (int)py); // src/render.cpp:38
                        smp = seed;
                        sampleIdx = 0;
                        Lsum = splat(0.0f);
                        nPixelRays = 1;
                        if (F.numPixelSamples > 1) { // generatePixelRays* draw every offset before tracing (src/render.cpp:82-123)
                            if (F.enableJitteredSampling) {
                                const uint32_t ns = (uint32_t)roundf(sqrtf((float)F.numPixelSamples));
                                nPixelRays = ns * ns;
                            } else {
                                nPixelRays = F.numPixelSamples;
                            }
                            for (uint32_t k = 0; k < 2 * nPixelRays; k++) (void)sampler_next(smp);
                        }
                        state = ST_RETURN; // enters the per-pixel ray loop with nothing accumulated yet
                        ret = splat(0.0f);
                        depth = P.startDepth - 1; // sentinel: "return to pixel level"
                        sampleIdx = 0xffffffffu;  // becomes 0 on the first pass through ST_RETURN
                    }
                }
            }
        }

        // ---- advance every lane to its next query (or to NEW / EXIT) ----
        while (state > ST_TRACE) {
            switch (state) {
            case ST_HIT_RAD: {
                // renderRay after state.bvh.intersect (src/recursive.cpp:29-36)
                if (ttri == kNoHit) {
                    ret = splat(0.0f); // sampleEnvironmentMap with the map disabled (src/extra.cpp:577-579)
                    state = ST_RETURN;
                    break;
                }
                ro = to, rd = td, rt = tt;
                Hit h;
                fill_hit(S, F, ttri, ro, rd, rt, h);
                hn = h.normal;
                hmesh = h.mesh;
                huv = h.uv;
                hsphere = h.isSphere;
                if (depth == P.startDepth && sampleIdx == 0 && P.primId) P.primId[py * P.W + px] = h.primId;
                if (depth == P.startDepth && sampleIdx == 0 && P.primT) P.primT[py * P.W + px] = rt;
                Lo = splat(0.0f);
                lightIdx = 0;
                lightSample = 0;
                areaAcc = splat(0.0f);
                state = ST_LIGHTS;
                break;
            }
            case ST_LIGHTS: {
                // computeLightContribution (src/light.cpp:262-276), resumable at (lightIdx, lightSample)
                if (lightIdx >= S.nLights) {
                    state = ST_SPAWN;
                    break;
                }
                const GpuLight& L = S.lights[lightIdx];
                const int type = (int)L.type;
                if (type != 0 && lightSample >= F.numShadowSamples) { // area light finished (src/light.cpp:185-194, 222-231)
                    Lo = Lo + areaAcc;
                    areaAcc = splat(0.0f);
                    lightSample = 0;
                    lightIdx++;
                    break;
                }
                if (type == 0) {
                    lpos = load3(L.f);
                    lcol = load3(L.f + 3);
                } else if (type == 1) { // sampleSegmentLight (src/light.cpp:25-29)
                    const float s = sampler_next(smp);
                    const vec3 e0 = load3(L.f), e1 = load3(L.f + 3), c0 = load3(L.f + 6), c1 = load3(L.f + 9);
                    lpos = e0 + s * (e1 - e0);
                    lcol = c0 + s * (c1 - c0);
                } else { // sampleParallelogramLight (src/light.cpp:40-53)
                    const float sx = sampler_next(smp);
                    const float sy = sampler_next(smp);
                    const vec3 v0 = load3(L.f), e01 = load3(L.f + 3), e02 = load3(L.f + 6);
                    lpos = v0 + sx * e01 + sy * e02;
                    const float a0 = (1.0f - sx) * (1.0f - sy), a1 = (1.0f - sx) * sy, a2 = sx * (1.0f - sy), a3 = sx * sy;
                    lcol = a0 * load3(L.f + 9) + a1 * load3(L.f + 12) + a2 * load3(L.f + 15) + a3 * load3(L.f + 18);
                }
                if (F.enableShadows) {
                    // visibilityOfLightSampleBinary (src/light.cpp:66-90): closest hit FROM the light
                    const vec3 ip = ro + rt * rd;
                    to = lpos;
                    td = normalize(ip - lpos);
                    tt = FLT_MAX;
                    traceIsShadow = true;
                    shIncoming = shPrev = lcol;
                    shSegments = 1;
                    cShadow++;
                    state = ST_TRACE;
                } else {
                    ttri = 0; // visible
                    tt = 0.0f;
                    state = ST_HIT_SHD;
                }
                break;
            }
            case ST_HIT_SHD: {
                vec3 visible = lcol;
                if (F.enableShadows && F.enableTransparency) {
                    // visibilityOfLightSampleTransparency (src/light.cpp:107-136): walk from the light through every
                    // surface on the way; each one tints and dims what gets past it; the light that reaches the LAST
                    // surface met counts
                    const vec3 ip = ro + rt * rd;
                    const bool intersects = (ttri != kNoHit);
                    const bool finalRayT = fabsf(tt - length(ip - to)) <= 1e-5f;
                    if (intersects) {
                        const bool sph = ttri >= kSphereBase;
                        const Mat hm = load_material(S, sph ? ttri - kSphereBase : S.triAttr[ttri].meshID, sph);
                        shPrev = shIncoming;
                        shIncoming = shIncoming * hm.kd * (1.0f - hm.transparency);
                        to = to + (tt + 1e-6f) * td;
                    }
                    // The reference's walk ends only when a surface is met within 1e-5 of the shading point or nothing
                    // is hit; when the shaded surface itself is met a little further off, it oscillates around it for
                    // ever.  A kernel must end: the walk stops after kMaxShadowSegments surfaces.
                    if (intersects && !finalRayT && shSegments < kMaxShadowSegments) {
                        shSegments++;
                        td = normalize(ip - to);
                        tt = FLT_MAX;
                        cShadow++;
                        state = ST_TRACE;
                        break;
                    }
                    visible = shPrev;
                } else if (F.enableShadows) {
                    const vec3 ip = ro + rt * rd;
                    const bool intersects = (ttri != kNoHit);
                    const bool expected = fabsf(tt - length(ip - lpos)) <= 5e-4f;
                    if (!(intersects && expected)) visible = splat(0.0f);
                }
                Mat m = load_material(S, hmesh, hsphere);
                if (!hsphere) m.kd = sample_material_kd(S, F, hmesh, huv, m.kd);
                const GpuLight& L = S.lights[lightIdx];
                if ((int)L.type == 0) { // computeContributionPointLight (src/light.cpp:151-159)
                    const vec3 p = ro + rt * rd;
                    const vec3 l = normalize(lpos - p);
                    Lo = Lo + shade(F, -rd, l, visible, hn, m);
                    lightIdx++;
                } else { // the sampled POSITION is handed over as the light direction (src/light.cpp:191, 228)
                    areaAcc = areaAcc + shade(F, -rd, lpos, visible, hn, m) / (1.0f * F.numShadowSamples);
                    lightSample++;
                }
                state = ST_LIGHTS;
                break;
            }
            case ST_SPAWN: {
                // recursive components (src/recursive.cpp:62-80): reflection first, then the pass-through ray
                bool spawned = false;
                float* fr = frames + (depth < RT_MAX_RAY_DEPTH ? depth : 0) * kFrameFloats; // a ray at the depth limit has no frame
                if (depth < RT_MAX_RAY_DEPTH) fr[27] = 0.0f;
                if (depth < RT_MAX_RAY_DEPTH && (F.enableReflections || F.enableTransparency)) {
                    const Mat m = load_material(S, hmesh, hsphere);
                    if (F.enableTransparency && m.transparency != 1.f) {
                        // generatePassthroughRay (src/recursive.cpp:112-123), kept until the reflection has returned
                        vec3 porg = ro + rt * rd;
                        porg = porg + 0.00001f * rd;
                        fr[20] = porg.x, fr[21] = porg.y, fr[22] = porg.z;
                        fr[23] = rd.x, fr[24] = rd.y, fr[25] = rd.z;
                        fr[26] = m.transparency;
                        fr[27] = 1.0f;
                    }
                    const bool reflective = (m.ks.x != 0.0f) || (m.ks.y != 0.0f) || (m.ks.z != 0.0f);
                    if (F.enableReflections && reflective) {
                        // generateReflectionRay (src/recursive.cpp:91-103)
                        vec3 rorg = ro + rt * rd;
                        const vec3 rdir = rd - 2.f * dot(rd, hn) * hn;
                        rorg = rorg + 0.00001f * rdir;
                        fr[0] = Lo.x, fr[1] = Lo.y, fr[2] = Lo.z;
                        fr[3] = m.ks.x, fr[4] = m.ks.y, fr[5] = m.ks.z;
                        if (!F.enableGlossyReflection) { // renderRaySpecularComponent (src/recursive.cpp:135-143)
                            fr[6] = __int_as_float(CK_SPEC);
                            to = rorg, td = rdir, tt = FLT_MAX;
                            traceIsShadow = false;
                            depth++;
                            cSecondary++;
                            state = ST_TRACE;
                            spawned = true;
                        } else if (F.numGlossySamples > 0) { // renderRayGlossyComponent (src/extra.cpp:464-508)
                            fr[6] = __int_as_float(CK_GLOSSY);
                            fr[7] = 0.5f / m.shininess; // disk radius
                            fr[8] = rorg.x, fr[9] = rorg.y, fr[10] = rorg.z;
                            fr[11] = rdir.x, fr[12] = rdir.y, fr[13] = rdir.z;
                            fr[14] = hn.x, fr[15] = hn.y, fr[16] = hn.z;
                            fr[17] = 0.0f, fr[18] = 0.0f, fr[19] = 0.0f; // accumulatedColor
                            lightSample = 0;                              // reused as the glossy sample index i
                            state = ST_GLOSSY_NEXT;
                            spawned = true;
                        }
                    }
                }
                if (!spawned) state = ST_PASS; // Lo = the hit's colour so far
                break;
            }
            case ST_PASS: {
                // renderRayTransparentComponent (src/recursive.cpp:155-163) of the hit at `depth`, whose colour so far is Lo
                float* fr = frames + (depth < RT_MAX_RAY_DEPTH ? depth : 0) * kFrameFloats;
                if (depth < RT_MAX_RAY_DEPTH && fr[27] != 0.0f) {
                    fr[0] = Lo.x, fr[1] = Lo.y, fr[2] = Lo.z;
                    fr[6] = __int_as_float(CK_PASS);
                    to = make3(fr[20], fr[21], fr[22]), td = make3(fr[23], fr[24], fr[25]), tt = FLT_MAX;
                    traceIsShadow = false;
                    depth++;
                    cSecondary++;
                    state = ST_TRACE;
                } else {
                    ret = Lo;
                    state = ST_RETURN;
                }
                break;
            }
            case ST_GLOSSY_NEXT: {
                // next accepted sample of the frame at `depth`, or the end of its loop
                float* fr = frames + depth * kFrameFloats;
                const vec3 rorg = make3(fr[8], fr[9], fr[10]), rdir = make3(fr[11], fr[12], fr[13]), nrm = make3(fr[14], fr[15], fr[16]);
                const float radius = fr[7];
                vec3 w, u, v;
                ortho_basis(rdir, w, u, v);
                bool launched = false;
                while (lightSample < F.numGlossySamples) {
                    // sampleDisk (src/extra.cpp:443-450): two draws per sample, consumed even if rejected
                    const float r = radius * sqrtf(sampler_next(smp));
                    const float angle = sampler_next(smp) * 2 * 3.14159265358979323846264338327950288f;
                    float sn, cs;
                    libm_sincosf(angle, sn, cs);
                    const float dx = r * cs, dy = r * sn;
                    const vec3 pd = normalize(w + dx * u + dy * v);
                    lightSample++;
                    if (dot(pd, nrm) <= 0) continue;
                    to = rorg + 0.0001f * pd;
                    td = pd;
                    tt = FLT_MAX;
                    traceIsShadow = false;
                    fr[6] = __int_as_float(CK_GLOSSY | (int)(lightSample << 8)); // resume index
                    depth++;
                    cSecondary++;
                    state = ST_TRACE;
                    launched = true;
                    break;
                }
                if (!launched) { // hitColor += accumulatedColor / float(numSamples) (src/extra.cpp:507)
                    const vec3 acc = make3(fr[17], fr[18], fr[19]);
                    Lo = make3(fr[0], fr[1], fr[2]) + acc / float(F.numGlossySamples);
                    state = ST_PASS;
                }
                break;
            }
            case ST_RETURN: {
                // `ret` is the radiance of the ray evaluated at `depth`; hand it to its caller
                if (depth <= P.startDepth) {
                    // pixel level: renderRays (src/recursive.cpp:10-17)
                    if (sampleIdx != 0xffffffffu) Lsum = Lsum + ret;
                    sampleIdx++;
                    if (sampleIdx < nPixelRays) {
                        float ndcX, ndcY;
                        if (F.numPixelSamples > 1) {
                            uint32_t s2 = seed;
                            for (uint32_t k = 0; k < 2 * sampleIdx; k++) (void)sampler_next(s2);
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
                        camera_ray(P.cam, ndcX, ndcY, to, td);
                        tt = FLT_MAX;
                        traceIsShadow = false;
                        depth = P.startDepth;
                        cPrimary++;
                        if (sampleIdx == 0) {
                            if (P.primId) P.primId[py * P.W + px] = kNoHit;
                            if (P.primT) P.primT[py * P.W + px] = FLT_MAX;
                        }
                        state = ST_TRACE;
                    } else {
                        const vec3 Lp = Lsum / static_cast<float>(nPixelRays);
                        float* out = P.rgb + 3 * ((size_t)(P.H - 1 - (int)py) * P.W + px); // Screen::setPixel (src/screen.cpp:36-42)
                        out[0] = Lp.x, out[1] = Lp.y, out[2] = Lp.z;
                        state = ST_NEW;
                    }
                    break;
                }
                depth--;
                float* fr = frames + depth * kFrameFloats;
                const int kind = __float_as_int(fr[6]);
                const vec3 ks = make3(fr[3], fr[4], fr[5]);
                if ((kind & 0xff) == CK_SPEC) {
                    Lo = make3(fr[0], fr[1], fr[2]) + ret * ks; // hitColor += specularColor * ks
                    state = ST_PASS;
                } else if ((kind & 0xff) == CK_PASS) { // hitColor = hitColor * (1 - transparency) + passColor * transparency
                    ret = make3(fr[0], fr[1], fr[2]) * (1.f - fr[26]) + ret * fr[26];
                    state = ST_RETURN;
                } else {
                    const vec3 acc = make3(fr[17], fr[18], fr[19]) + ks * ret; // accumulatedColor += ks * renderRay(...)
                    fr[17] = acc.x, fr[18] = acc.y, fr[19] = acc.z;
                    lightSample = (uint32_t)kind >> 8;
                    state = ST_GLOSSY_NEXT;
                }
                break;
            }
            default: state = ST_EXIT; break;
            }
        }

        // ---- one closest-hit query per lane that has one ----
        const unsigned long long tracing = __ballot(state == ST_TRACE);
        if (tracing == 0ull) {
            if (__ballot(state == ST_NEW) != 0ull) continue;
            break;
        }
        if (state == ST_TRACE) {
            trace<CULL, COUNT>(S, F.enableAccelStructure != 0, to, td, tt, ttri, stack, tc);
            state = traceIsShadow ? ST_HIT_SHD : ST_HIT_RAD;
        }
    }

    // ---- ray counters: wave reduction, one atomic per counter per wave ----
    uint32_t vals[7] = { cPrimary, cShadow, cSecondary, tc.nodePairs, tc.triTests, cPixels, tc.overflow };
#pragma unroll
    for (int i = 0; i < 7; i++) {
        unsigned long long v = vals[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && v) atomicAdd(P.stats + (blockIdx.x % kStatCopies) * kStatStride + i, v);
    }
}

#include "rt_wavefront.h"
#include "rt_path.h"

// ---------------------------------------------------------------------------------------------
// Batch closest-hit kernel (BVHInterface::intersect for n rays)
// ---------------------------------------------------------------------------------------------
struct IntersectParams {
    SceneView S;
    rt_features F;
    float* rays; // n x 7
    uint32_t* hits; // n x 12
    uint32_t n;
};
__global__ __launch_bounds__(kBlock) void intersect_kernel(const IntersectParams P)
{
    __shared__ uint32_t s_stack[kStackDepth * kBlock];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P.n) return;
    float* r = P.rays + 7 * (size_t)i;
    const vec3 o = load3(r), d = load3(r + 3);
    float t = r[6];
    uint32_t tri;
    TraceCounters tc { 0, 0, 0 };
    trace<false, false>(P.S, P.F.enableAccelStructure != 0, o, d, t, tri, s_stack + threadIdx.x, tc);
    uint32_t* h = P.hits + 12 * (size_t)i;
    if (tri == kNoHit) {
        for (int k = 0; k < 12; k++) h[k] = 0;
        h[8] = kNoHit;
        return;
    }
    Hit hit;
    fill_hit(P.S, P.F, tri, o, d, t, hit);
    r[6] = t;
    h[0] = __float_as_uint(hit.normal.x), h[1] = __float_as_uint(hit.normal.y), h[2] = __float_as_uint(hit.normal.z);
    h[3] = __float_as_uint(hit.bary.x), h[4] = __float_as_uint(hit.bary.y), h[5] = __float_as_uint(hit.bary.z);
    h[6] = __float_as_uint(hit.uv.x), h[7] = __float_as_uint(hit.uv.y);
    h[8] = hit.primId, h[9] = hit.mesh, h[10] = 1, h[11] = 0;
}

// ---------------------------------------------------------------------------------------------
// Bloom (postprocessImageWithBloom, src/extra.cpp:330-416) as two LDS-tiled passes.
// Memory rows are Screen rows (row m holds pixel row y = H-1-m).  Taps outside the image read the
// zero border the reference pads with (padBorders, src/extra.cpp:236-256); sums run in the
// reference's tap order with separate multiply and add.
// ---------------------------------------------------------------------------------------------
constexpr int kBloomMaxF = 30;
struct BloomParams {
    const float* src; // pre-bloom image (pass 1) / unused (pass 2)
    float* rgb;       // image updated in place by pass 2
    float* tmp;       // row-filtered thresholded image
    int W, H, F;
    float threshold, intensity;
    int stripeRows, rank, world; // rows written by pass 2
    float taps[2 * kBloomMaxF + 1];
};

// Scratch of rt_bloom_static: the row-filtered image of the largest frame that entry point accepts.  One copy per GPU
// that loads the library, held until the process ends; kernels receive its device address as an argument.
static __device__ __align__(256) float g_bloomScratch[3ull * RT_BLOOM_STATIC_MAX_PIXELS];
constexpr int kMaxDevices = 64;
float* g_bloomScratchPtr[kMaxDevices] = {}; // device address of g_bloomScratch per device (hipGetSymbolAddress)
std::mutex g_bloomScratchMutex;
// The address of the current device's copy of the static bloom array, looked up on the first call per device (with that
// device current) and kept.  rt_set_device and rt_scene_create call it too, so that a caller who has set up a scene on a GPU
// finds the address resolved when it later calls rt_bloom_static from a place that must not take a lock or query the runtime.
int bloom_static_scratch(const int device, float** out)
{
    if (device < 0 || device >= kMaxDevices) {
        rt::set_error("rt_bloom_static: device %d outside [0,%d)", device, kMaxDevices);
        return RT_ERR_LIMIT;
    }
    std::lock_guard<std::mutex> lock(g_bloomScratchMutex);
    if (!g_bloomScratchPtr[device]) {
        void* p = nullptr;
        RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_bloomScratch)));
        g_bloomScratchPtr[device] = static_cast<float*>(p);
    }
    *out = g_bloomScratchPtr[device];
    return RT_OK;
}


constexpr int kBloomRowTile = 256;
// pass 1: threshold (applyThreshold, :186-196) + filter along x (applyFilter1dToPixel toRow, :307-314)
__global__ __launch_bounds__(kBloomRowTile) void bloom_rows_kernel(const BloomParams P)
{
    __shared__ float s_px[3][kBloomRowTile + 2 * kBloomMaxF];
    const int m = blockIdx.y, x0 = blockIdx.x * kBloomRowTile, F = P.F;
    const int span = kBloomRowTile + 2 * F;
    for (int i = threadIdx.x; i < span; i += kBloomRowTile) {
        const int x = x0 - F + i;
        float r = 0.0f, g = 0.0f, b = 0.0f;
        if (x >= 0 && x < P.W) {
            const float* p = P.src + 3 * ((size_t)m * P.W + x);
            r = p[0], g = p[1], b = p[2];
            const float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b; // perceivedLuminance (:177-184)
            if (lum < P.threshold) r = g = b = 0.0f;
        }
        s_px[0][i] = r, s_px[1][i] = g, s_px[2][i] = b;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= P.W) return;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    for (int k = 0; k <= 2 * F; k++) {
        const float f = P.taps[k];
        ar = ar + s_px[0][threadIdx.x + k] * f;
        ag = ag + s_px[1][threadIdx.x + k] * f;
        ab = ab + s_px[2][threadIdx.x + k] * f;
    }
    float* o = P.tmp + 3 * ((size_t)m * P.W + x);
    o[0] = ar, o[1] = ag, o[2] = ab;
}

constexpr int kBloomColX = 64, kBloomColY = 16; // output tile of pass 2: 64 columns x 16 memory rows, 256 threads
// pass 2: filter along y (toRow = false, :317-321) + combine (:403-413).  Tap k of pixel row y reads pixel
// row y - F + k, i.e. memory row m + F - k.
__global__ __launch_bounds__(kBloomColX * 4) void bloom_cols_kernel(const BloomParams P)
{
    __shared__ float s_px[3][kBloomColY + 2 * kBloomMaxF][kBloomColX];
    const int F = P.F, x0 = blockIdx.x * kBloomColX, m0 = blockIdx.y * kBloomColY;
    const int tx = threadIdx.x & (kBloomColX - 1), tyBase = threadIdx.x / kBloomColX; // 4 row groups
    const int rows = kBloomColY + 2 * F;
    const int x = x0 + tx;
    for (int i = tyBase; i < rows; i += 4) {
        const int m = m0 - F + i;
        float r = 0.0f, g = 0.0f, b = 0.0f;
        if (m >= 0 && m < P.H && x < P.W) {
            const float* p = P.tmp + 3 * ((size_t)m * P.W + x);
            r = p[0], g = p[1], b = p[2];
        }
        s_px[0][i][tx] = r, s_px[1][i][tx] = g, s_px[2][i][tx] = b;
    }
    __syncthreads();
    if (x >= P.W) return;
    for (int j = tyBase; j < kBloomColY; j += 4) {
        const int m = m0 + j;
        if (m >= P.H) break;
        if (P.world > 1) { // only this rank's stripes (stripes are cut on pixel rows y = H-1-m)
            const int y = P.H - 1 - m;
            if ((y / P.stripeRows) % P.world != P.rank) continue;
        }
        float ar = 0.0f, ag = 0.0f, ab = 0.0f;
        for (int k = 0; k <= 2 * F; k++) {
            const float f = P.taps[k];
            const int i = j + 2 * F - k; // local row of memory row m + F - k
            ar = ar + s_px[0][i][tx] * f;
            ag = ag + s_px[1][i][tx] * f;
            ab = ab + s_px[2][i][tx] * f;
        }
        float* o = P.rgb + 3 * ((size_t)m * P.W + x);
        o[0] = o[0] + ar * P.intensity;
        o[1] = o[1] + ag * P.intensity;
        o[2] = o[2] + ab * P.intensity;
    }
}

// Self-test of div_by against IEEE division on random and edge-case operands of the range in which
// the traversal uses it.  counts[0] += mismatching results, counts[1] += cases checked; counts[4..] keep the operands
// and results of the first 16 mismatches.
__global__ __launch_bounds__(kBlock) void division_selftest_kernel(const uint32_t seed, const uint32_t perThread, unsigned long long* counts)
{
    uint32_t s = seed ^ (blockIdx.x * kBlock + threadIdx.x) * 2654435761u;
    auto next = [&]() {
        s = s * 747796405u + 2891336453u;
        uint32_t v = s;
        v ^= v >> ((v >> 28u) + 4u);
        v *= 277803737u;
        v ^= v >> 22u;
        return v;
    };
    unsigned long long bad = 0, n = 0;
    for (uint32_t i = 0; i < perThread; i++) {
        const uint32_t mode = next() & 7u;
        // divisor: any mantissa, exponent in [-30, 30], either sign
        const uint32_t db = (next() & 0x807fffffu) | ((97u + next() % 61u) << 23);
        const float d = __uint_as_float(db);
        float a;
        if (mode < 4) { // numerator bound - origin with both in the fast coordinate range (differences can cancel heavily)
            const float x = __uint_as_float((next() & 0x807fffffu) | ((67u + next() % 91u) << 23));
            float y = __uint_as_float((next() & 0x807fffffu) | ((67u + next() % 91u) << 23));
            if (mode == 1) y = __uint_as_float(__float_as_uint(x) + (next() % 5u) - 2u); // neighbours of x
            if (mode == 2) y = 0.0f;
            a = x - y;
        } else if (mode < 6) { // scene-scale numerators
            a = ((float)(int)(next() % 2000001u) - 1000000.0f) * 1e-6f * (float)(1u << (next() % 8u));
        } else if (mode == 6) { // exact multiples of the divisor (remainder zero) and their neighbours
            a = d * (float)(int)(1u + next() % 4096u);
            a = __uint_as_float(__float_as_uint(a) + (next() % 3u) - 1u);
            if (fabsf(a) < 5.169878828e-26f) a = 0.0f; // keep to {0} U [2^-84, ...): the range ray_is_fast guarantees
        } else {
            a = __uint_as_float((next() & 0x807fffffu) | ((43u + next() % 115u) << 23)); // |a| in [2^-84, 2^31)
        }
        const float want = a / d;
        const float got = div_by(a, d, 1.0f / d);
        n++;
        if (__float_as_uint(want) != __float_as_uint(got)) {
            bad++;
            const unsigned long long slot = atomicAdd(counts + 2, 1ull);
            if (slot < 16) {
                counts[4 + 2 * slot] = ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(d);
                counts[5 + 2 * slot] = ((unsigned long long)__float_as_uint(want) << 32) | __float_as_uint(got);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_xor(bad, off);
        n += __shfl_xor(n, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + 0, bad);
        atomicAdd(counts + 1, n);
    }
}

// binomial / generateGaussianFilter (src/extra.cpp:258-292)
uint64_t binomial(uint32_t n, uint32_t k)
{
    const uint32_t k_min = k > n - k ? n - k : k;
    uint64_t coef = 1;
    for (uint32_t i = 1; i <= k_min; i++) coef = (coef * (n - i + 1)) / i;
    return coef;
}
// Between two frames of a launch sequence that is rendered frame by frame (keepStats): the next frame zeroes the queue
// counters and the launch counters, so what rt_sync needs of the frame just rendered moves into the statistics block.
__global__ void carry_frame_counts_kernel(const uint32_t* qCount, const uint32_t* fallCount, unsigned long long* stats)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if ((unsigned long long)qCount[0] > stats[S_QMAX_SHADE]) stats[S_QMAX_SHADE] = qCount[0];
    if ((unsigned long long)qCount[1] > stats[S_QMAX_LEAF]) stats[S_QMAX_LEAF] = qCount[1];
    if (fallCount) stats[S_FALL_SUM] += *fallCount;
}

// rt_scene_create's triangle work, from the caller's primitives() as uploaded: compact triangle i is primitive srcPos[i].
// The boxes feed the device build of the query tree; the records are the scene's own (precompute_triangle: the host's function).
__global__ __launch_bounds__(kBlock) void triangle_boxes_kernel(const RefPrimitive* prims, const uint32_t* srcPos, const uint32_t n, float4* lo4, float4* hi4)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const RefPrimitive& p = prims[srcPos[i]];
    const vec3 v0 = load3(p.v0.position), v1 = load3(p.v1.position), v2 = load3(p.v2.position);
    const vec3 lo = gmin(gmin(v0, v1), v2), hi = gmax(gmax(v0, v1), v2);
    lo4[i] = make_float4(lo.x, lo.y, lo.z, 0.0f), hi4[i] = make_float4(hi.x, hi.y, hi.z, 0.0f);
}
__global__ __launch_bounds__(kBlock) void triangle_records_kernel(const RefPrimitive* prims, const uint32_t* srcPos, const uint32_t* srcPrimId, const uint32_t n,
                                                                  TriPlane* plane, TriEdges* edges, TriAttr* attr)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    TriPlane pl;
    TriEdges ed;
    TriAttr at;
    precompute_triangle(prims[srcPos[i]], srcPrimId[i], pl, ed, at);
    plane[i] = pl, edges[i] = ed, attr[i] = at;
}

// The triangle records of the query tree: place i of its leaf order holds triangle order[i] of the caller's tree - the plane and
// edge records copied, and the certificate (the triangle's index, the box of its reference leaf and that leaf's route from
// the root) put together.  Runs once per scene, after the upload.
__global__ __launch_bounds__(kBlock) void gather_query_triangles_kernel(const uint32_t* order, const TriPlane* plane, const TriEdges* edges,
                                                                        const float* leafBox, const uint32_t* leafRoute, const uint32_t n,
                                                                        TriPlane* qPlane, TriEdges* qEdges, TriCert* qCert)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = order[i];
    reinterpret_cast<float4*>(qPlane)[i] = reinterpret_cast<const float4*>(plane)[t];
    const float4* src = reinterpret_cast<const float4*>(edges + t);
    float4* dst = reinterpret_cast<float4*>(qEdges + i);
#pragma unroll
    for (int k = 0; k < 5; k++) dst[k] = src[k];
    const float* b = leafBox + 6 * (size_t)t;
    float4* c = reinterpret_cast<float4*>(qCert + i);
    c[0] = make_float4(b[0], b[1], b[2], __uint_as_float(t));
    c[1] = make_float4(b[3], b[4], b[5], __uint_as_float(leafRoute[t]));
}

void gaussian_taps(uint32_t F, float* taps)
{
    const uint32_t radius = F * 2 + 1;
    float sum = 0.0f;
    for (uint32_t i = 0; i < radius; i++) {
        taps[i] = float(binomial(radius - 1, i));
        sum += taps[i];
    }
    for (uint32_t i = 0; i < radius; i++) taps[i] /= sum;
}

int check_features(const rt_features* f)
{
    if (f->shadingModel < 0 || f->shadingModel > 3) {
        rt::set_error("shadingModel %d out of range", f->shadingModel);
        return RT_ERR_INVALID;
    }
    if (f->numPixelSamples == 0 || f->numPixelSamples > 4096) {
        rt::set_error("numPixelSamples %u out of range", f->numPixelSamples);
        return RT_ERR_INVALID;
    }
    return RT_OK;
}

template <typename T>
T* carve(char*& p, size_t n)
{
    T* r = reinterpret_cast<T*>(p);
    p += (n * sizeof(T) + 255) & ~size_t(255);
    return r;
}
size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }

void free_workspace(Workspace* w)
{
    if (!w) return;
    (void)hipFree(w->d_stats), (void)hipFree(w->d_work), (void)hipFree(w->d_cams), (void)hipFree(w->d_accum), (void)hipFree(w->d_nonfinite);
    if (w->h_status) (void)hipHostFree(w->h_status);
    (void)hipFree(w->d_shadeQ), (void)hipFree(w->d_leafQ), (void)hipFree(w->d_states), (void)hipFree(w->d_frames), (void)hipFree(w->d_passes);
    (void)hipFree(w->d_active[0]), (void)hipFree(w->d_active[1]), (void)hipFree(w->d_activeRay[0]), (void)hipFree(w->d_activeRay[1]);
    (void)hipFree(w->d_roundCounts), (void)hipFree(w->d_pathRays[0]), (void)hipFree(w->d_pathRays[1]);
    (void)hipFree(w->d_hits), (void)hipFree(w->d_shadowRays), (void)hipFree(w->d_longList), (void)hipFree(w->d_longList2), (void)hipFree(w->d_fallList);
    (void)hipFree(w->d_orderKeys), (void)hipFree(w->d_order), (void)hipFree(w->d_orderHist), (void)hipFree(w->d_waveTimes);
    delete w;
}

constexpr size_t kMaxWorkspaces = 8;
// The buffers of the frames queued on `stream`: frames on one stream share a set (they run one after the other anyway), a
// stream not seen before gets its own, and beyond kMaxWorkspaces streams the set unused for longest is taken over once the
// frames of its stream have finished.  Null (with the error set) when nothing can be allocated.
Workspace* workspace_for(rt_scene* scene, hipStream_t stream, const bool create)
{
    scene->useCounter++;
    for (Workspace* w : scene->workspaces)
        if (w->stream == stream) {
            w->lastUse = scene->useCounter;
            return w;
        }
    if (!create) return nullptr;
    Workspace* w = nullptr;
    if (scene->workspaces.size() < kMaxWorkspaces) {
        w = new (std::nothrow) Workspace();
        if (!w) {
            rt::set_error("out of memory");
            return nullptr;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&w->d_stats), kStatCopies * kStatStride * sizeof(unsigned long long));
        // (cleared by every rt_render on the frame's own stream: a memset on the null stream is not ordered with the
        // non-blocking streams frames are queued on, and may land after their first kernels)
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&w->d_work), 256);
        if (e != hipSuccess) {
            rt::set_error("device allocation of a frame's counters failed: %s", hipGetErrorString(e));
            free_workspace(w);
            return nullptr;
        }
        scene->workspaces.push_back(w);
    } else {
        w = scene->workspaces[0];
        for (Workspace* c : scene->workspaces)
            if (c->lastUse < w->lastUse) w = c;
        if (hipStreamSynchronize(w->stream) != hipSuccess) {
            rt::set_error("hipStreamSynchronize failed while taking over a frame's buffers");
            return nullptr;
        }
        RetiredFrame r;
        r.stream = w->stream;
        r.status = collect_frame(scene, *w, &r.stats);
        r.message = r.status != RT_OK ? rt_last_error() : "";
        for (size_t i = 0; i < scene->retired.size(); i++)
            if (scene->retired[i].stream == r.stream) scene->retired.erase(scene->retired.begin() + i--);
        if (scene->retired.size() >= 64) scene->retired.erase(scene->retired.begin());
        scene->retired.push_back(r);
    }
    for (size_t i = 0; i < scene->retired.size(); i++) // this stream's frames report through `w` from now on
        if (scene->retired[i].stream == stream) scene->retired.erase(scene->retired.begin() + i--);
    w->stream = stream, w->lastUse = scene->useCounter;
    return w;
}

} // namespace

// =============================================================================================
// C ABI, device side
// =============================================================================================
extern "C" {

int rt_device_count(int* count)
{
    if (!count) {
        rt::set_error("rt_device_count: null argument");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipGetDeviceCount(count));
    return RT_OK;
}
int rt_set_device(int device)
{
    RT_HIP(hipSetDevice(device));
    float* unused = nullptr;
    return bloom_static_scratch(device, &unused); // (see there: resolved at set-up, for rt_bloom_static)
}

int rt_scene_create(const void* nodes, uint32_t nNodes, const void* prims, uint32_t nPrims, const uint32_t* primIds,
                    const float* materials8, uint32_t nMaterials, const float* lights22, uint32_t nLights,
                    const float* spheres12, uint32_t nSpheres, rt_scene** out)
{
    if (!out || !nodes || nNodes == 0 || (nPrims && !prims) || (nMaterials && !materials8) || (nLights && !lights22)
        || (nSpheres && !spheres12)) {
        rt::set_error("rt_scene_create: null or empty argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    for (uint32_t i = 0; i < nLights; i++) {
        const float ty = lights22[22 * (size_t)i];
        if (!(ty == 0.0f || ty == 1.0f || ty == 2.0f)) {
            rt::set_error("rt_scene_create: light %u has unknown type %g", i, (double)ty);
            return RT_ERR_INVALID;
        }
    }
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t0) { return std::chrono::duration<float, std::milli>(clock::now() - t0).count(); };
    const auto tStart = clock::now();
    FlatScene flat;
    const char* err = "";
    // (the triangles' records are computed on the device, from the primitives as they are: RT_AMD_HOST_RECORDS keeps them on the host)
    const bool deviceRecords = getenv("RT_AMD_HOST_RECORDS") == nullptr && nPrims != 0;
    int rc = flatten(static_cast<const RefNode*>(nodes), nNodes, static_cast<const RefPrimitive*>(prims), nPrims, primIds, nMaterials,
                     flat, &err, !deviceRecords);
    if (rc != RT_OK) {
        rt::set_error("%s", err);
        return rc;
    }
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) {
        rt::set_error("rt_scene_create: out of memory");
        return RT_ERR_NOMEM;
    }
    if (hipGetDevice(&s->device) != hipSuccess) {
        delete s;
        rt::set_error("rt_scene_create: no HIP device available");
        return RT_ERR_HIP;
    }
    {
        float* unused = nullptr;
        (void)bloom_static_scratch(s->device, &unused); // set-up on this device: rt_bloom_static finds its array's address resolved
    }
    const float flattenMs = ms_since(tStart);
    const auto tQuery = clock::now();
    QueryTree qt;
    // the library's own tree: hierarchy on the device unless RT_AMD_HOST_QUERY_BUILD asks for the host builder
    // (its triangle records are gathered on the device, after the upload)
    const size_t nPairs = flat.pairs.size(), nTris = flat.srcPos.size();
    // the primitives, the compact triangles' places in them and their boxes on the device (released when the scene is up)
    char* d_temp = nullptr;
    const RefPrimitive* dPrims = nullptr;
    const uint32_t *dSrcPos = nullptr, *dSrcPrimId = nullptr;
    float4 *dLo4 = nullptr, *dHi4 = nullptr;
    if (deviceRecords && nTris) {
        const size_t bytes = padded((size_t)nPrims * sizeof(RefPrimitive)) + 2 * padded(nTris * 4) + 2 * padded(nTris * 16);
        hipError_t te = hipMalloc(reinterpret_cast<void**>(&d_temp), bytes);
        if (te == hipSuccess) {
            char* tp = d_temp;
            dPrims = carve<RefPrimitive>(tp, nPrims), dSrcPos = carve<uint32_t>(tp, nTris), dSrcPrimId = carve<uint32_t>(tp, nTris);
            dLo4 = const_cast<float4*>(carve<float4>(tp, nTris)), dHi4 = const_cast<float4*>(carve<float4>(tp, nTris));
            te = hipMemcpy(const_cast<RefPrimitive*>(dPrims), prims, (size_t)nPrims * sizeof(RefPrimitive), hipMemcpyHostToDevice);
            if (te == hipSuccess) te = hipMemcpy(const_cast<uint32_t*>(dSrcPos), flat.srcPos.data(), nTris * 4, hipMemcpyHostToDevice);
            if (te == hipSuccess) te = hipMemcpy(const_cast<uint32_t*>(dSrcPrimId), flat.srcPrimId.data(), nTris * 4, hipMemcpyHostToDevice);
            if (te == hipSuccess) {
                hipLaunchKernelGGL(triangle_boxes_kernel, dim3((unsigned)((nTris + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, dPrims, dSrcPos, (uint32_t)nTris, dLo4, dHi4);
                te = hipGetLastError();
            }
        }
        if (te != hipSuccess) {
            rt::set_error("rt_scene_create: device allocation/upload of the primitives failed: %s", hipGetErrorString(te));
            (void)hipFree(d_temp);
            delete s;
            return RT_ERR_HIP;
        }
    }
    if (!getenv("RT_AMD_NO_QUERY_TREE"))
        build_query_tree(flat, qt, getenv("RT_AMD_HOST_QUERY_BUILD") == nullptr, false, static_cast<const RefPrimitive*>(prims),
                         reinterpret_cast<const float*>(dLo4), reinterpret_cast<const float*>(dHi4));
    const float queryMs = ms_since(tQuery);
    const auto tUpload = clock::now();
    const size_t nQNodes = qt.usable ? qt.nodes.size() : 0, nQPairs = qt.usable ? qt.pairs.size() : 0, nQTris = qt.usable ? nTris : 0;
    const size_t nQuadNodes = qt.usable ? qt.quadNodes.size() : 0, nQuadLeaves = qt.usable ? qt.quadLeaves.size() : 0;
    // (the last three: what the query-order copies are gathered from on the device - place -> triangle, and each triangle's
    // reference leaf box and route; kept in the allocation, read by no kernel afterwards)
    const size_t total = padded(nPairs * sizeof(NodePair)) + padded(nTris * sizeof(TriPlane)) + padded(nTris * sizeof(TriEdges))
        + padded(nTris * sizeof(TriAttr)) + padded(nTris * 4) + padded(flat.leafCounts.size() * 4) + padded((nTris / 32 + 1) * 4) + padded(nMaterials * sizeof(GpuMaterial))
        + padded(nLights * sizeof(GpuLight)) + padded(nSpheres * sizeof(GpuSphere)) + padded(nQNodes * sizeof(Node4)) + padded(nQPairs * sizeof(NodePair))
        + padded(nQTris * sizeof(TriPlane)) + padded(nQTris * sizeof(TriEdges)) + padded(nQTris * sizeof(TriCert))
        + padded(nQuadNodes * sizeof(QuadNode)) + padded(nQuadLeaves * sizeof(QuadLeaf)) + padded(nQTris * 4) + padded(nQTris * 24) + padded(nQTris * 4) + 256;
    // the small tables, as the kernels read them
    std::vector<uint32_t> hReflects(nTris / 32 + 1, 0u);
    for (size_t i = 0; i < nTris; i++) { // the material test of renderRay (src/recursive.cpp:62-80), once per triangle
        const float* m = materials8 + 8 * (size_t)static_cast<const RefPrimitive*>(prims)[flat.srcPos[i]].meshID;
        if (m[3] != 0.0f || m[4] != 0.0f || m[5] != 0.0f) hReflects[i >> 5] |= 1u << (i & 31);
    }
    std::vector<GpuMaterial> hMat(nMaterials);
    for (uint32_t i = 0; i < nMaterials; i++) {
        const float* m = materials8 + 8 * (size_t)i;
        hMat[i] = GpuMaterial { { m[0], m[1], m[2] }, m[6], { m[3], m[4], m[5] }, m[7] };
    }
    std::vector<GpuLight> hLight(nLights);
    for (uint32_t i = 0; i < nLights; i++) {
        const float* l = lights22 + 22 * (size_t)i;
        hLight[i].type = l[0];
        memcpy(hLight[i].f, l + 1, 21 * 4);
        hLight[i].pad[0] = hLight[i].pad[1] = 0.0f;
    }
    std::vector<GpuSphere> hSph(nSpheres);
    for (uint32_t i = 0; i < nSpheres; i++) {
        const float* p = spheres12 + 12 * (size_t)i;
        hSph[i] = GpuSphere { { p[0], p[1], p[2] }, p[3], { p[4], p[5], p[6] }, p[10], { p[7], p[8], p[9] }, p[11] };
    }
    // One allocation, every array copied into its place straight from where it was built (no second copy on the host); the
    // triangle records in query-tree order and their certificates are gathered from the uploaded ones by a kernel.
    hipError_t e = hipMalloc(&s->d_blob, total);
    char* up = static_cast<char*>(s->d_blob);
    auto put = [&](auto*& d, const void* src, size_t count) { // carves `count` records for d and fills them from src (null: left as they are)
        using T = std::remove_const_t<std::remove_pointer_t<std::remove_reference_t<decltype(d)>>>;
        d = carve<T>(up, count);
        if (e == hipSuccess && src && count) e = hipMemcpy(const_cast<T*>(d), src, count * sizeof(T), hipMemcpyHostToDevice);
    };
    const NodePair *uPairs = nullptr, *uQPairs = nullptr;
    const TriPlane *uPlane = nullptr, *uQPlane = nullptr;
    const TriEdges *uEdges = nullptr, *uQEdges = nullptr;
    const TriAttr* uAttr = nullptr;
    const uint32_t *uBrute = nullptr, *uLeafCounts = nullptr, *uReflects = nullptr, *uOrder = nullptr, *uRoute = nullptr;
    const GpuMaterial* uMat = nullptr;
    const GpuLight* uLight = nullptr;
    const GpuSphere* uSph = nullptr;
    const Node4* uQNodes = nullptr;
    const TriCert* uQCert = nullptr;
    const QuadNode* uQuadNodes = nullptr;
    const QuadLeaf* uQuadLeaves = nullptr;
    struct Box6 {
        float v[6];
    };
    const Box6* uLeafBox = nullptr;
    if (e == hipSuccess) {
        put(uPairs, flat.pairs.data(), nPairs);
        put(uPlane, deviceRecords ? nullptr : flat.triPlane.data(), nTris);
        put(uEdges, deviceRecords ? nullptr : flat.triEdges.data(), nTris);
        put(uAttr, deviceRecords ? nullptr : flat.triAttr.data(), nTris);
        if (e == hipSuccess && deviceRecords && nTris) {
            hipLaunchKernelGGL(triangle_records_kernel, dim3((unsigned)((nTris + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, dPrims, dSrcPos, dSrcPrimId,
                               (uint32_t)nTris, const_cast<TriPlane*>(uPlane), const_cast<TriEdges*>(uEdges), const_cast<TriAttr*>(uAttr));
            e = hipGetLastError();
        }
        put(uBrute, flat.bruteOrder.data(), nTris);
        put(uLeafCounts, flat.leafCounts.data(), flat.leafCounts.size());
        put(uReflects, hReflects.data(), nTris / 32 + 1);
        put(uMat, hMat.data(), nMaterials);
        put(uLight, hLight.data(), nLights);
        put(uSph, hSph.data(), nSpheres);
        put(uQNodes, qt.nodes.data(), nQNodes);
        put(uQPairs, qt.pairs.data(), nQPairs);
        put(uQPlane, qt.triPlane.empty() ? nullptr : qt.triPlane.data(), nQTris);
        put(uQEdges, qt.triEdges.empty() ? nullptr : qt.triEdges.data(), nQTris);
        put(uQCert, qt.cert.empty() ? nullptr : qt.cert.data(), nQTris);
        put(uQuadNodes, qt.quadNodes.data(), nQuadNodes);
        put(uQuadLeaves, qt.quadLeaves.data(), nQuadLeaves);
        put(uOrder, qt.order.data(), nQTris);
        put(uLeafBox, flat.leafBox.data(), nQTris);
        put(uRoute, flat.leafRoute.data(), nQTris);
        if (e == hipSuccess && nQTris && qt.triPlane.empty()) { // (the host did not gather them)
            hipLaunchKernelGGL(gather_query_triangles_kernel, dim3((unsigned)((nQTris + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, uOrder, uPlane,
                               uEdges, reinterpret_cast<const float*>(uLeafBox), uRoute, (uint32_t)nQTris, const_cast<TriPlane*>(uQPlane),
                               const_cast<TriEdges*>(uQEdges), const_cast<TriCert*>(uQCert));
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->d_textures), (nMaterials ? nMaterials : 1) * sizeof(GpuTexture));
    if (e == hipSuccess) e = hipMemset(s->d_textures, 0, (nMaterials ? nMaterials : 1) * sizeof(GpuTexture));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // frames run on streams the null stream's work is not ordered with
    (void)hipFree(d_temp);
    s->texturePixels.assign(nMaterials, nullptr);
    if (e != hipSuccess) {
        rt::set_error("rt_scene_create: device allocation/upload of %zu bytes failed: %s", total, hipGetErrorString(e));
        rt_scene_destroy(s);
        return RT_ERR_HIP;
    }
    s->blobBytes = total;
    char* dp = static_cast<char*>(s->d_blob);
    SceneView& V = s->view;
    V.pairs = carve<NodePair>(dp, nPairs);
    V.triPlane = carve<TriPlane>(dp, nTris);
    V.triEdges = carve<TriEdges>(dp, nTris);
    V.triAttr = carve<TriAttr>(dp, nTris);
    V.bruteOrder = carve<uint32_t>(dp, nTris);
    V.leafCounts = carve<uint32_t>(dp, flat.leafCounts.size());
    if (flat.leafCounts.empty()) V.leafCounts = nullptr;
    V.triReflects = carve<uint32_t>(dp, nTris / 32 + 1);
    V.materials = carve<GpuMaterial>(dp, nMaterials);
    V.lights = carve<GpuLight>(dp, nLights);
    V.spheres = carve<GpuSphere>(dp, nSpheres);
    V.textures = s->d_textures;
    const Node4* dQNodes = carve<Node4>(dp, nQNodes);
    const NodePair* dQPairs = carve<NodePair>(dp, nQPairs);
    const TriPlane* dQPlane = carve<TriPlane>(dp, nQTris);
    const TriEdges* dQEdges = carve<TriEdges>(dp, nQTris);
    const TriCert* dQCert = carve<TriCert>(dp, nQTris);
    const QuadNode* dQuadNodes = carve<QuadNode>(dp, nQuadNodes);
    const QuadLeaf* dQuadLeaves = carve<QuadLeaf>(dp, nQuadLeaves);
    V.cert = nullptr, V.nodes4 = nullptr, V.quadNodes = nullptr, V.quadLeaves = nullptr;
    V.refPairs = V.pairs, V.refRootRef = flat.rootRef;
    memcpy(V.rootLo, flat.rootLo, 12);
    memcpy(V.rootHi, flat.rootHi, 12);
    V.rootRef = flat.rootRef;
    V.nPairs = uint32_t(nPairs), V.nTris = uint32_t(nTris), V.nMaterials = nMaterials, V.nLights = nLights, V.nSpheres = nSpheres;
    V.fastBounds = flat.fastBounds ? 1u : 0u;
    s->treeDepth = flat.maxDepth;
    s->queryView = V;
    if (qt.usable) { // same root box: a query starts with the reference's own root test
        s->queryView.pairs = dQPairs, s->queryView.nodes4 = dQNodes;
        if (nQuadNodes) s->queryView.quadNodes = dQuadNodes, s->queryView.quadLeaves = dQuadLeaves;
        s->queryView.triPlane = dQPlane, s->queryView.triEdges = dQEdges, s->queryView.cert = dQCert;
        s->queryView.rootRef = qt.pairRootRef, s->queryView.nPairs = uint32_t(nQPairs); // root of the 4-wide form: node 0
        s->queryDepth = qt.pairDepth;
        s->queryDepth4 = qt.maxDepth;
    }
    if (getenv("RT_AMD_VERBOSE"))
        fprintf(stderr, "rt_scene_create: %.2f ms: flattening the caller's tree %.2f, query tree %.2f, assembling and uploading %zu bytes %.2f\n",
                ms_since(tStart), flattenMs, queryMs, total, ms_since(tUpload));
    if (getenv("RT_AMD_VERBOSE"))
        fprintf(stderr, "rt_scene_create: %zu triangles, reference tree %zu pairs depth %u, query tree %zu pairs depth %u = %zu 4-wide nodes depth %u%s; "
                        "hierarchy built on the %s in %.2f ms (launches %.2f ms), layouts written in %.2f ms\n",
                nTris, nPairs, flat.maxDepth, nQPairs, qt.pairDepth, nQNodes, qt.maxDepth, qt.usable ? "" : " (not used)",
                qt.builtOnDevice ? "device" : "host", qt.hierarchyMs, qt.deviceKernelMs, qt.layoutMs);
    if (const char* e = getenv("RT_AMD_T1")) s->laneModeAbove = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_BATCH_WINDOW")) s->batchWindow = (uint32_t)std::min(5, std::max(0, atoi(e)));
    if (getenv("RT_AMD_NO_BATCH")) s->batchWindow = 0;
    s->noPlan = getenv("RT_AMD_NO_PLAN") != nullptr, s->scatter = getenv("RT_AMD_SCATTER") != nullptr;
    s->fullQueues = getenv("RT_AMD_FULL_QUEUES") != nullptr;
    if (const char* e = getenv("RT_AMD_GROUP_NODE4")) s->groupNode4 = atoi(e) != 0;
    if (const char* e = getenv("RT_AMD_WAVE_TIMES")) s->waveTimesRound = atoi(e);
    s->verbose = getenv("RT_AMD_VERBOSE") != nullptr;
    if (const char* e = getenv("RT_AMD_WIDE_BELOW")) s->wideBelow = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_BUDGET")) s->stepBudget = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_QBUDGET")) s->queryStepBudget = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_T1_OVERLAPPED")) s->laneModeAboveOverlapped = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_ORDER_FROM")) s->orderRaysFrom = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_QBUDGET2")) s->secondPassBudget = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_QBUDGET2_OVERLAPPED")) s->secondPassBudgetOverlapped = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_QBUDGET_OVERLAPPED")) s->queryStepBudgetOverlapped = (uint32_t)atoi(e);
    if (getenv("RT_AMD_NO_SHADOW_BOUND")) s->noShadowBound = true;
    if (const char* e = getenv("RT_AMD_PACKETS")) s->packets = atoi(e) != 0;
    if (const char* e = getenv("RT_AMD_LARGE_FROM")) s->largeFromPixels = (uint32_t)atoll(e);
    if (const char* e = getenv("RT_AMD_BATCH_WINDOW_LARGE")) s->batchWindowLarge = (uint32_t)std::min(5, std::max(0, atoi(e)));
    if (const char* e = getenv("RT_AMD_QBUDGET_LARGE")) s->queryStepBudgetLarge = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_GROUP_LANES")) {
        const int g = atoi(e);
        if (g == 1 || g == 4 || g == 8 || g == 16 || g == 32) s->groupLanes = (uint32_t)g;
    }
    if (const char* e = getenv("RT_AMD_TRACE_BLOCKS")) s->laneBlocksPerCU = s->laneBlocksPerCUPooled = (uint32_t)(atoi(e) > 0 ? atoi(e) : 1);
    if (const char* e = getenv("RT_AMD_TRACE_BLOCKS_POOLED")) s->laneBlocksPerCUPooled = (uint32_t)(atoi(e) > 0 ? atoi(e) : 1);
    if (const char* e = getenv("RT_AMD_LOGIC_BLOCKS")) s->logicBlocksPerCU = (uint32_t)(atoi(e) > 0 ? atoi(e) : 1);
    if (const char* e = getenv("RT_AMD_GROUP_BLOCKS")) s->groupBlocksPerCU = (uint32_t)(atoi(e) > 0 ? atoi(e) : 0);
    if (const char* e = getenv("RT_AMD_PATH_MODE")) s->pathMode = atoi(e);
    if (const char* e = getenv("RT_AMD_PATH_QUADS")) s->pathQuads = atoi(e);
    if (getenv("RT_AMD_QUAD_TRACE")) s->quadTrace = true;
    if (const char* e = getenv("RT_AMD_LINK_BUDGET")) s->linkBudget = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_MIN_LANES")) s->minLanes = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_ADVANCE_AT")) s->advanceAt = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_ADVANCE_EVERY")) s->advanceEvery = (uint32_t)atoi(e);
    if (const char* e = getenv("RT_AMD_PATH_BLOCKS")) s->pathBlocksPerCU = (uint32_t)(atoi(e) > 0 ? atoi(e) : 1);
    for (uint32_t i = 0; i < nLights; i++) s->lightTypes.push_back((int)lights22[22 * (size_t)i]);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device) == hipSuccess && cus > 0) s->numCUs = cus;
    *out = s;
    return RT_OK;
}

int rt_scene_set_texture(rt_scene* scene, uint32_t material, const float* rgb, int width, int height)
{
    if (!scene || material >= scene->view.nMaterials || (rgb && (width <= 0 || height <= 0 || (int64_t)width * height > (1 << 28)))) {
        rt::set_error("rt_scene_set_texture: null scene, material index out of range or empty image");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(scene->device));
    RT_HIP(hipDeviceSynchronize()); // frames in flight may still read the old texels
    float* d_px = nullptr;
    if (rgb) {
        const size_t bytes = (size_t)width * height * 3 * sizeof(float);
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&d_px), bytes));
        if (hipMemcpy(d_px, rgb, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_px);
            rt::set_error("rt_scene_set_texture: upload of %zu bytes failed", bytes);
            return RT_ERR_HIP;
        }
    }
    const GpuTexture t { d_px, rgb ? width : 0, rgb ? height : 0 };
    if (hipMemcpy(scene->d_textures + material, &t, sizeof(t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d_px);
        rt::set_error("rt_scene_set_texture: table update failed");
        return RT_ERR_HIP;
    }
    (void)hipFree(scene->texturePixels[material]);
    scene->texturePixels[material] = d_px;
    return RT_OK;
}

int rt_scene_device(const rt_scene* scene, int* device)
{
    if (!scene || !device) {
        rt::set_error("rt_scene_device: null argument");
        return RT_ERR_INVALID;
    }
    *device = scene->device;
    return RT_OK;
}

int rt_scene_stream(rt_scene* scene, int index, void** stream)
{
    if (!scene || !stream || index < 0 || index >= 4) {
        rt::set_error("rt_scene_stream: null argument or index outside [0,4)");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(scene->device));
    rt_scene::ImageLane& l = scene->imageLanes[index];
    if (!l.stream) {
        // equal sequences on equal streams run in step, all asking for the same resource at a time: two of the four
        // streams get the higher of the device's stream priorities and the sequences drift apart
        int least = 0, greatest = 0;
        RT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        RT_HIP(hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, index < 2 ? greatest : least));
    }
    *stream = l.stream;
    return RT_OK;
}

// The scene once more, on another GPU (or the same): the uploaded arrays are copied device to device and every pointer into
// them moved by the distance between the two copies; textures likewise.  Render state (working buffers, streams, launch
// plan) is not shared: the copy starts without any.
int rt_scene_clone(const rt_scene* src, int device, rt_scene** out)
{
    if (!src || !out) {
        rt::set_error("rt_scene_clone: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    int count = 0;
    RT_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        rt::set_error("rt_scene_clone: device %d outside [0,%d)", device, count);
        return RT_ERR_INVALID;
    }
    int previous = 0;
    RT_HIP(hipGetDevice(&previous));
    rt_scene* s = new (std::nothrow) rt_scene(*src); // the knobs, tree depths, light types ... (and pointers that are replaced below)
    if (!s) {
        rt::set_error("rt_scene_clone: out of memory");
        return RT_ERR_NOMEM;
    }
    s->device = device;
    s->d_blob = nullptr, s->d_rgb = s->d_scratch = nullptr, s->d_primId = nullptr, s->d_primT = nullptr, s->fbPixels = 0;
    for (rt_scene::ImageLane& l : s->imageLanes) l = rt_scene::ImageLane();
    s->d_textures = nullptr;
    s->texturePixels.assign(src->texturePixels.size(), nullptr);
    s->retired.clear(), s->roundRays.clear(), s->planKey = 0;
    s->workspaces.clear(), s->useCounter = 0;
    auto fail = [&](const char* what, hipError_t e) {
        rt::set_error("rt_scene_clone: %s failed: %s", what, hipGetErrorString(e));
        (void)hipGetLastError();
        rt_scene_destroy(s);
        (void)hipSetDevice(previous);
        return RT_ERR_HIP;
    };
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail("hipSetDevice", e);
    {
        float* unused = nullptr;
        (void)bloom_static_scratch(device, &unused);
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->numCUs = cus;
    }
    e = hipMalloc(&s->d_blob, src->blobBytes);
    if (e == hipSuccess) e = hipMemcpyPeer(s->d_blob, device, src->d_blob, src->device, src->blobBytes);
    if (e != hipSuccess) return fail("copying the scene's arrays", e);
    const ptrdiff_t moved = static_cast<char*>(s->d_blob) - static_cast<char*>(src->d_blob);
    auto move_view = [&](SceneView& V) {
        auto mv = [&](auto*& p) {
            using T = std::remove_reference_t<decltype(p)>;
            if (p) p = reinterpret_cast<T>(reinterpret_cast<char*>(const_cast<void*>(static_cast<const void*>(p))) + moved);
        };
        mv(V.pairs), mv(V.triPlane), mv(V.triEdges), mv(V.triAttr), mv(V.bruteOrder), mv(V.leafCounts), mv(V.triReflects);
        mv(V.materials), mv(V.lights), mv(V.spheres), mv(V.cert), mv(V.nodes4), mv(V.quadNodes), mv(V.quadLeaves), mv(V.refPairs);
    };
    move_view(s->view), move_view(s->queryView);
    // textures: the table's entries point at allocations of their own
    const size_t nMat = src->texturePixels.size();
    std::vector<GpuTexture> table(nMat ? nMat : 1);
    memset(table.data(), 0, table.size() * sizeof(GpuTexture));
    e = hipMalloc(reinterpret_cast<void**>(&s->d_textures), table.size() * sizeof(GpuTexture));
    if (e != hipSuccess) return fail("hipMalloc of the texture table", e);
    if (nMat) {
        e = hipSetDevice(src->device);
        if (e == hipSuccess) e = hipMemcpy(table.data(), src->d_textures, nMat * sizeof(GpuTexture), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipSetDevice(device);
        if (e != hipSuccess) return fail("reading the texture table", e);
        for (size_t m = 0; m < nMat; m++) {
            if (!src->texturePixels[m]) {
                memset(&table[m], 0, sizeof(GpuTexture));
                continue;
            }
            const size_t bytes = (size_t)table[m].width * (size_t)table[m].height * 12;
            e = hipMalloc(reinterpret_cast<void**>(&s->texturePixels[m]), bytes);
            if (e == hipSuccess) e = hipMemcpyPeer(s->texturePixels[m], device, src->texturePixels[m], src->device, bytes);
            if (e != hipSuccess) return fail("copying a texture", e);
            table[m].pixels = s->texturePixels[m];
        }
    }
    e = hipMemcpy(s->d_textures, table.data(), table.size() * sizeof(GpuTexture), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail("writing the texture table", e);
    s->view.textures = s->d_textures, s->queryView.textures = s->d_textures;
    (void)hipSetDevice(previous);
    *out = s;
    return RT_OK;
}

int rt_scene_trim(rt_scene* s)
{
    if (!s) {
        rt::set_error("rt_scene_trim: null scene");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipDeviceSynchronize()); // nothing of this scene may still be running
    for (Workspace* w : s->workspaces) free_workspace(w);
    s->workspaces.clear();
    for (rt_scene::ImageLane& l : s->imageLanes) {
        (void)hipFree(l.d_rgb), (void)hipFree(l.d_scratch);
        if (l.h_stage) (void)hipHostFree(l.h_stage);
        l.d_rgb = l.d_scratch = l.h_stage = nullptr, l.frames = l.px = 0;
    }
    (void)hipFree(s->d_rgb), (void)hipFree(s->d_scratch), (void)hipFree(s->d_primId), (void)hipFree(s->d_primT);
    s->d_rgb = s->d_scratch = nullptr, s->d_primId = nullptr, s->d_primT = nullptr, s->fbPixels = 0;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* s)
{
    if (!s) return;
    for (float* p : s->texturePixels) (void)hipFree(p);
    (void)hipFree(s->d_textures);
    (void)hipFree(s->d_blob);
    (void)hipFree(s->d_rgb);
    (void)hipFree(s->d_scratch);
    (void)hipFree(s->d_primId);
    (void)hipFree(s->d_primT);
    for (rt_scene::ImageLane& l : s->imageLanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        (void)hipFree(l.d_rgb), (void)hipFree(l.d_scratch);
        if (l.h_stage) (void)hipHostFree(l.h_stage);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (Workspace* w : s->workspaces) free_workspace(w);
    delete s;
}

// rt_render_frames for frames that can share one launch sequence.  kNotPooled (nothing has been launched then): this
// description of a frame is rendered by a schedule that takes one frame at a time.  keepStats: the ray counters go on from
// the frame queued before on this stream.
static constexpr int kNotPooled = -1000;
static int render_frames_impl(rt_scene* scene, const rt_camera* camera, const int nFrames, const rt_features* features, int width, int height,
                              int startDepth, const rt_region* region, uint32_t flags, float* d_rgb, uint32_t* d_primId, float* d_primT, void* stream,
                              const bool keepStats)
{
    if (!scene || !camera || !features || !d_rgb || width <= 0 || height <= 0) {
        rt::set_error("rt_render: null argument or empty frame");
        return RT_ERR_INVALID;
    }
    if (nFrames < 1 || nFrames > RT_MAX_FRAMES) {
        rt::set_error("rt_render_frames: %d frames outside [1,%d]", nFrames, (int)RT_MAX_FRAMES);
        return RT_ERR_INVALID;
    }
    if (startDepth < 0 || startDepth > RT_MAX_RAY_DEPTH) {
        rt::set_error("rt_render: startDepth %d outside [0,%d]", startDepth, (int)RT_MAX_RAY_DEPTH);
        return RT_ERR_INVALID;
    }
    if ((long long)width * height * nFrames > (1ll << 30)) {
        rt::set_error("rt_render: %d frame(s) of %dx%d pixels exceed the 2^30 pixel limit", nFrames, width, height);
        return RT_ERR_LIMIT;
    }
    if (nFrames > 1 && (long long)height * nFrames > 65535) { // a pixel's packed position keeps frame * height + y in 16 bits
        rt::set_error("rt_render_frames: %d frames of %d rows exceed 65535 rows per launch sequence", nFrames, height);
        return RT_ERR_LIMIT;
    }
    if (!(flags & RT_RENDER_SERIAL) && (width > 65535 || height > 65535)) { // ... and x in the other 16 (the depth-first kernel packs nothing)
        rt::set_error("rt_render: a frame of %dx%d pixels exceeds 65535 columns or rows - use RT_RENDER_SERIAL", width, height);
        return RT_ERR_LIMIT;
    }
    int rc = check_features(features);
    if (rc != RT_OK) return rc;
    rt_region reg { 8, 0, 1 };
    if (region && region->world > 1) reg = *region;
    if (reg.stripeRows <= 0 || (reg.stripeRows % 8) != 0 || reg.rank < 0 || reg.rank >= reg.world) {
        rt::set_error("rt_render: region {stripeRows=%d (must be a positive multiple of 8), rank=%d, world=%d} is invalid",
                      reg.stripeRows, reg.rank, reg.world);
        return RT_ERR_INVALID;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    RT_HIP(hipSetDevice(scene->device));
    Workspace* const wsp = workspace_for(scene, st, true);
    if (!wsp) return RT_ERR_HIP;
    Workspace& w = *wsp;
    const int tilesX = (width + 7) / 8;
    const int tileRows = (height + 7) / 8;
    const int stripeTileRows = reg.stripeRows / 8;
    // stripes rank, rank+world, ... belong to this rank; the kernels enumerate full stripes and reject
    // rows past the frame per pixel
    const int stripes = (tileRows + stripeTileRows - 1) / stripeTileRows;
    const int myStripes = (stripes > reg.rank) ? (stripes - reg.rank + reg.world - 1) / reg.world : 0;
    const int myTileRows = myStripes * stripeTileRows;
    const uint32_t nWork = (uint32_t)nFrames * (uint32_t)myTileRows * (uint32_t)tilesX * 64u;
    if (!keepStats) RT_HIP(hipMemsetAsync(w.d_stats, 0, kStatCopies * kStatStride * sizeof(unsigned long long), st));
    else if (w.lastWasWavefront) { // the frame before this one of the sequence: its queue and fall-back counters are about to be zeroed
        // (schedules that render frame by frame leave at most one fall-back counter: the persistent launch's)
        const uint32_t* fall = (w.fallSlots.size() == 1 && w.d_roundCounts) ? w.d_roundCounts + w.fallSlots[0] : nullptr;
        hipLaunchKernelGGL(carry_frame_counts_kernel, dim3(1), dim3(64), 0, st, w.d_work + 8, fall, w.d_stats);
    }
    RT_HIP(hipMemsetAsync(w.d_work, 0, 256, st));
    w.lastWasWavefront = false;
    w.statusQueued = false;
    if (nWork == 0) return RT_OK;
    // countRef: the counters of the reference's walk (its rays, the caller's tree); countSchedule: the same counters, and
    // the query tree's, for the frame exactly as it is scheduled by default (query tree, candidate batches, handed-over rays)
    const bool countRef = (flags & RT_RENDER_COUNT_WORK) != 0, countSchedule = !countRef && (flags & RT_RENDER_COUNT_SCHEDULE) != 0;
    const bool cull = !(flags & RT_RENDER_NO_CULL), count = countRef || countSchedule;
    // a frame that shares the GPU with others: one ray per lane (cheaper per ray, slower for the stage's longest ray) from fewer rays on
    const bool overlapped = (flags & RT_RENDER_OVERLAPPED) != 0;
    const uint32_t laneModeAbove = overlapped ? scene->laneModeAboveOverlapped : scene->laneModeAbove;
    const bool large = scene->largeFromPixels != 0 && nWork >= scene->largeFromPixels;
    const uint32_t queryStepBudget = large ? scene->queryStepBudgetLarge : (overlapped ? scene->queryStepBudgetOverlapped : scene->queryStepBudget);
    const uint32_t sceneBatchWindow = large ? std::min(scene->batchWindow, scene->batchWindowLarge) : scene->batchWindow;
    const uint32_t fullGrid = (nWork + kBlock - 1) / kBlock;
    uint32_t grid = (uint32_t)scene->numCUs * 4u; // persistent: 4 workgroups per CU
    if (grid > fullGrid) grid = fullGrid;

    // With transparency a hit can continue twice (reflection, then the pass-through ray): the sequential rays of a full
    // recursion tree, one per round.  Depth-6 children of a glossy loop are deferred when no light draws from the sampler.
    uint64_t transparentChain = 0;
    if (features->enableTransparency) {
        bool area = false;
        for (int ty : scene->lightTypes) area = area || ty != 0;
        const bool glossy = features->enableReflections && features->enableGlossyReflection;
        uint64_t t = 1; // a ray at the depth limit
        for (int d = RT_MAX_RAY_DEPTH - 1; d >= startDepth; d--) {
            uint64_t refl = 0;
            if (features->enableReflections) refl = !glossy ? t : ((!area && d == RT_MAX_RAY_DEPTH - 1) ? 0 : features->numGlossySamples * t);
            t = std::min<uint64_t>(1 + refl + t, 1u << 24);
        }
        uint64_t nRays = features->numPixelSamples > 1 ? features->numPixelSamples : 1;
        if (features->numPixelSamples > 1 && features->enableJitteredSampling) {
            const uint64_t ns = (uint64_t)roundf(sqrtf((float)features->numPixelSamples));
            nRays = ns * ns;
        }
        transparentChain = t * nRays;
    }
    // the depth-first kernel: on request, and for transparent frames whose recursion tree is too long for rounds
    if ((flags & RT_RENDER_SERIAL) || transparentChain > kMaxRounds - 1) {
        if (nFrames > 1) return kNotPooled;
        RenderParams P;
        P.S = scene->view;
        P.cam = *camera;
        P.F = *features;
        P.W = width, P.H = height, P.startDepth = startDepth;
        P.tilesX = tilesX, P.myTileRows = myTileRows, P.stripeTileRows = stripeTileRows;
        P.rank = reg.rank, P.world = reg.world;
        P.nWork = nWork;
        P.rgb = d_rgb, P.primId = d_primId, P.primT = d_primT;
        P.stats = w.d_stats, P.work = w.d_work;
        if (cull && !count) hipLaunchKernelGGL((render_kernel<true, false, true>), dim3(grid), dim3(kBlock), 0, st, P);
        else if (cull && count) hipLaunchKernelGGL((render_kernel<true, true, true>), dim3(grid), dim3(kBlock), 0, st, P);
        else if (!cull && !count) hipLaunchKernelGGL((render_kernel<false, false, true>), dim3(grid), dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL((render_kernel<false, true, true>), dim3(grid), dim3(kBlock), 0, st, P);
        RT_HIP(hipGetLastError());
        queue_status(w, st, 0);
        return RT_OK;
    }

    // ---- wavefront pipeline ----
    const bool groupOk = features->enableAccelStructure && scene->groupLanes > 1 && (int)scene->treeDepth <= (scene->groupLanes == 4 ? GroupLayout<4>::kMaxTreeDepth : 30);
    const bool useQuery = groupOk && cull && !countRef && scene->queryView.cert != nullptr && (scene->groupLanes == 8 || scene->groupLanes == 16)
        && scene->queryDepth <= 48 && !(flags & RT_RENDER_REFERENCE_TREE);
    // The path stage - the rays whose outcome the pixel's sampler depends on - runs as one persistent launch in which a group
    // of lanes owns a pixel from its first ray to its last (path_kernel), or, without it, in frame-wide rounds.
    const bool usePathGroup = scene->pathMode == 1 && groupOk && cull && !count && (scene->groupLanes == 8 || scene->groupLanes == 16)
        && !features->enableTransparency; // the persistent schedules know no pass-through rays
    // Rounds a full recursion tree needs when its loops are walked one child per round (batched loops need fewer): beyond what
    // the round-based stage accepts, the pixel-per-lane launch takes the frame, whose chains may be as long as they like.
    bool longChains = false;
    {
        const uint64_t fan = (features->enableReflections && features->enableGlossyReflection) ? std::max(1u, features->numGlossySamples) : 1;
        uint64_t links = 0, term = 1;
        for (int l = startDepth; l <= RT_MAX_RAY_DEPTH && links < (1u << 24); l++) {
            links += term, term *= fan;
            if (!features->enableReflections) break;
        }
        const uint64_t ns = features->enableJitteredSampling ? (uint64_t)roundf(sqrtf((float)features->numPixelSamples)) : 0;
        links *= std::max<uint64_t>(features->numPixelSamples, ns * ns);
        longChains = links > kMaxRounds / 2;
    }
    const bool usePathLane = (scene->pathMode == 2 || (longChains && scene->pathMode == 0)) && cull && !count && !features->enableTransparency;
    const bool usePath = usePathGroup || usePathLane;
    if (usePath && nFrames > 1) return kNotPooled; // the persistent schedules take one frame at a time
    // (buffers are sized only now: a sequence that is rendered frame by frame after all asks for one frame's worth)
    const size_t px = (size_t)width * height * (size_t)nFrames;
    if (px > w.accumPixels) {
        (void)hipFree(w.d_accum), (void)hipFree(w.d_nonfinite);
        w.d_accum = nullptr, w.d_nonfinite = nullptr, w.accumPixels = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_accum), px * 3 * sizeof(long long)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_nonfinite), px * sizeof(uint32_t)));
        RT_HIP(hipMemsetAsync(w.d_accum, 0, px * 3 * sizeof(long long), st)); // on the frame's stream: ordered before its kernels
        RT_HIP(hipMemsetAsync(w.d_nonfinite, 0, px * sizeof(uint32_t), st));
        w.accumPixels = px;
    }
    uint64_t want = (uint64_t)nWork * 8u;
    if (want < (1u << 20)) want = 1u << 20;
    uint64_t cfgKey = 1469598103934665603ull; // FNV-1a over what decides how many tasks and rays a pixel gives rise to
    {
        auto mix = [&](const void* p, size_t n) {
            const unsigned char* b = static_cast<const unsigned char*>(p);
            for (size_t i = 0; i < n; i++) cfgKey = (cfgKey ^ b[i]) * 1099511628211ull;
        };
        const int32_t depth = startDepth; // (the flags change how rays are traced and counted, not which rays and tasks there are)
        mix(&depth, sizeof(depth)), mix(features, sizeof(*features));
    }
    w.lastCfg = cfgKey;
    rt_scene::SeenQueues held;
    if (!scene->fullQueues) {
        const auto it = scene->seenQueues.find(cfgKey);
        if (it != scene->seenQueues.end()) held = it->second;
    }
    const bool seen = held.shade >= 0.0;
    auto sized_for = [&](const double perPixel) { return (uint64_t)(perPixel * 1.5 * (double)nWork) + 65536u; };
    const uint64_t wantShade = std::min<uint64_t>(std::max(seen ? sized_for(held.shade) : want, scene->wantShadeTasks), kMaxQueueTasks);
    const uint64_t wantLeaf = std::min<uint64_t>(std::max(seen ? sized_for(held.leaf) : want, scene->wantLeafTasks), kMaxQueueTasks);
    if (w.shadeCap < wantShade) {
        (void)hipFree(w.d_shadeQ);
        w.d_shadeQ = nullptr, w.shadeCap = 0;
        RT_HIP(hipMalloc(&w.d_shadeQ, wantShade * sizeof(ShadeTask)));
        w.shadeCap = (uint32_t)wantShade;
    }
    if (w.leafCap < wantLeaf) {
        (void)hipFree(w.d_leafQ);
        w.d_leafQ = nullptr, w.leafCap = 0;
        RT_HIP(hipMalloc(&w.d_leafQ, wantLeaf * sizeof(LeafTask)));
        w.leafCap = (uint32_t)wantLeaf;
    }
    // (rounds only) a glossy loop whose children are the last rays that can spawn anything is traced as one batch of candidates
    uint32_t batchRays = 0, batchWindow = 0;
    if (!usePath) {
        bool area = false;
        for (int ty : scene->lightTypes) area = area || ty != 0;
        const uint32_t N = features->numGlossySamples;
        if (features->enableReflections && features->enableGlossyReflection && !area && N >= 2 && sceneBatchWindow >= 2
            && startDepth <= RT_MAX_RAY_DEPTH - 2 && !countRef && !features->enableTransparency) { // a counting pass walks only the reference's rays
            batchWindow = std::min(N, sceneBatchWindow);
            batchRays = batchWindow * (batchWindow + 1) / 2;
        }
    }
    // rays of one round: every state's largest batch at worst; a round of a completed frame tells what to expect (never fewer than
    // one ray per pixel: the camera's)
    size_t pathRaysWanted = (size_t)nWork * (batchRays ? batchRays : 1u);
    if (seen && held.rays >= 0.0) pathRaysWanted = std::min<size_t>(pathRaysWanted, std::max<size_t>(nWork, sized_for(held.rays)));
    pathRaysWanted = std::max<size_t>(pathRaysWanted, std::min<size_t>(scene->wantPathRays, (size_t)nWork * (batchRays ? batchRays : 1u)));
    // path states: the rounds keep 64-byte headers here and the loop frames / pass records in arrays of their own; a persistent
    // launch keeps a state's header, frames and pass records together (kStateDwords)
    const uint32_t stateStride = usePath ? (uint32_t)kStateDwords : 16u;
    if (w.stateWords < (size_t)nWork * stateStride) {
        (void)hipFree(w.d_states);
        w.d_states = nullptr, w.stateWords = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_states), (size_t)nWork * stateStride * sizeof(float)));
        w.stateWords = (size_t)nWork * stateStride;
    }
    // a loop is left pending at the rayDepth of its hit: startDepth at the least, one short of the depth limit at the most
    const int firstLevel = std::min(std::max(startDepth, 0), RT_MAX_RAY_DEPTH - 1);
    const size_t levelStates = (size_t)nWork * (size_t)(RT_MAX_RAY_DEPTH - firstLevel);
    if (!usePath && w.frameStates < levelStates) {
        (void)hipFree(w.d_frames);
        w.d_frames = nullptr, w.frameStates = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_frames), levelStates * kFrameDwords * sizeof(float)));
        w.frameStates = levelStates;
    }
    if (!usePath && features->enableTransparency && w.passStates < levelStates) {
        (void)hipFree(w.d_passes);
        w.d_passes = nullptr, w.passStates = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_passes), levelStates * kPassDwords * sizeof(float)));
        w.passStates = levelStates;
    }
    if (w.stateCap < nWork) {
        (void)hipFree(w.d_active[0]), (void)hipFree(w.d_active[1]);
        (void)hipFree(w.d_activeRay[0]), (void)hipFree(w.d_activeRay[1]);
        w.d_active[0] = w.d_active[1] = nullptr, w.stateCap = 0;
        w.d_activeRay[0] = w.d_activeRay[1] = nullptr;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_active[0]), (size_t)nWork * sizeof(uint32_t)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_active[1]), (size_t)nWork * sizeof(uint32_t)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_activeRay[0]), (size_t)nWork * sizeof(uint32_t)));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_activeRay[1]), (size_t)nWork * sizeof(uint32_t)));
        w.stateCap = nWork;
    }
    if (w.pathRayCap < pathRaysWanted) {
        (void)hipFree(w.d_pathRays[0]), (void)hipFree(w.d_pathRays[1]);
        w.d_pathRays[0] = w.d_pathRays[1] = nullptr, w.pathRayCap = 0;
        RT_HIP(hipMalloc(&w.d_pathRays[0], pathRaysWanted * sizeof(RayRec)));
        RT_HIP(hipMalloc(&w.d_pathRays[1], pathRaysWanted * sizeof(RayRec)));
        w.pathRayCap = pathRaysWanted;
    }
    const size_t countWords = kCountWords;
    if (!w.d_roundCounts) RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_roundCounts), countWords * sizeof(uint32_t)));
    // shadow rays: one per point light and numShadowSamples per area light for every shade task, generated and
    // traced in chunks of tasks that fit the shadow-ray buffer
    uint32_t raysPerTask = 0, lightDraws = 0;
    for (uint32_t i = 0; i < scene->view.nLights; i++) {
        const int ty = scene->lightTypes[i];
        raysPerTask += (ty == 0) ? 1u : features->numShadowSamples;
        if (ty == 1) lightDraws += features->numShadowSamples;
        else if (ty == 2) lightDraws += 2 * features->numShadowSamples;
    }
    const bool shadows = features->enableShadows && raysPerTask > 0;
    uint32_t chunkTasks = w.shadeCap, nChunks = 1;
    if (shadows) {
        if (raysPerTask > kMaxShadowRays) {
            rt::set_error("rt_render: %u shadow rays per hit exceed the limit of %llu", raysPerTask, (unsigned long long)kMaxShadowRays);
            return RT_ERR_LIMIT;
        }
        uint64_t raysWanted = (uint64_t)w.shadeCap * raysPerTask;
        if (raysWanted > kMaxShadowRays) raysWanted = kMaxShadowRays - (kMaxShadowRays % raysPerTask);
        chunkTasks = (uint32_t)(raysWanted / raysPerTask);
        nChunks = (w.shadeCap + chunkTasks - 1) / chunkTasks;
        if (w.shadowCap < raysWanted) {
            (void)hipFree(w.d_shadowRays);
            w.d_shadowRays = nullptr, w.shadowCap = 0;
            RT_HIP(hipMalloc(&w.d_shadowRays, raysWanted * sizeof(RayRec)));
            w.shadowCap = raysWanted;
        }
    }
    {
        size_t hitsWanted = pathRaysWanted;
        if (hitsWanted < w.leafCap) hitsWanted = w.leafCap;
        if (shadows && hitsWanted < (size_t)chunkTasks * raysPerTask) hitsWanted = (size_t)chunkTasks * raysPerTask;
        if (w.hitCap < hitsWanted) {
            (void)hipFree(w.d_hits), (void)hipFree(w.d_longList), (void)hipFree(w.d_longList2), (void)hipFree(w.d_fallList);
            (void)hipFree(w.d_orderKeys), (void)hipFree(w.d_order);
            w.d_hits = nullptr, w.d_longList = nullptr, w.d_longList2 = nullptr, w.d_fallList = nullptr, w.hitCap = 0;
            w.d_orderKeys = nullptr, w.d_order = nullptr;
            RT_HIP(hipMalloc(&w.d_hits, hitsWanted * sizeof(HitRec)));
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_longList), hitsWanted * sizeof(uint32_t)));
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_longList2), hitsWanted * sizeof(uint32_t)));
            if (scene->orderRaysFrom) {
                RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_orderKeys), hitsWanted * sizeof(uint32_t)));
                RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_order), hitsWanted * sizeof(uint32_t)));
                if (!w.d_orderHist) RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_orderHist), kOrderBuckets * sizeof(uint32_t)));
            }
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_fallList), hitsWanted * sizeof(uint32_t)));
            w.hitCap = hitsWanted;
        }
    }
    // Rounds needed by a full tree: one ray per level of the sequential part, times the fan-out
    uint64_t chain = 1;
    {
        const int lastLevel = (lightDraws == 0 && features->enableGlossyReflection) ? RT_MAX_RAY_DEPTH - 1 : RT_MAX_RAY_DEPTH;
        const int levels = lastLevel - startDepth + 1; // ray depths traced by the path stage
        const uint64_t fan = (features->enableReflections && features->enableGlossyReflection) ? features->numGlossySamples : 1;
        uint64_t term = 1;
        chain = 0;
        for (int l = 0; l < levels && chain < kMaxRounds; l++) {
            // the deepest level of a batched frame costs one round per PARENT ray, not one per ray
            chain += (batchRays && l == levels - 1 && l > 0) ? term / (fan ? fan : 1) * ((fan + batchWindow - 1) / batchWindow) : term;
            term *= (fan ? fan : 1);
            if (!features->enableReflections) break;
        }
        if (chain < 1) chain = 1;
        // rays per pixel as pixel_ray_count has them: generatePixelRaysStratified takes round(sqrt(N))^2 (src/render.cpp:100-123)
        uint64_t nRays = features->numPixelSamples > 1 ? features->numPixelSamples : 1;
        if (features->numPixelSamples > 1 && features->enableJitteredSampling) {
            const uint64_t ns = (uint64_t)roundf(sqrtf((float)features->numPixelSamples));
            nRays = ns * ns;
        }
        chain *= nRays;
    }
    if (features->enableTransparency) chain = transparentChain;
    if (usePath) chain = 0; // no rounds: a pixel's chain may be as long as it likes
    if (chain > kMaxRounds - 1 || kTraceSlotsPerRound * nChunks + 32 > kMaxTraceLaunches - kTraceSlotsPerRound * kMaxRounds) {
        rt::set_error("rt_render: a full recursion tree needs %llu sequential rays per pixel (limit %u) and %u shadow chunks - "
                      "use RT_RENDER_SERIAL",
                      (unsigned long long)chain, kMaxRounds - 1, nChunks);
        return RT_ERR_LIMIT;
    }
    RT_HIP(hipMemsetAsync(w.d_roundCounts, 0, countWords * sizeof(uint32_t), st));
    w.fallSlots.clear(), w.longSlots.clear();
    WaveParams P;
    P.S = scene->view;
    P.F = *features;
    P.W = width, P.H = height, P.startDepth = startDepth;
    P.tilesX = tilesX, P.myTileRows = myTileRows, P.stripeTileRows = stripeTileRows;
    P.rank = reg.rank, P.world = reg.world;
    P.frames = nFrames;
    { // the cameras reach the kernels through device memory: a lane picks its frame's
        if (!w.d_cams) RT_HIP(hipMalloc(reinterpret_cast<void**>(&w.d_cams), RT_MAX_FRAMES * sizeof(rt_camera)));
        CameraPack pack;
        for (int f = 0; f < nFrames; f++) pack.c[f] = camera[f];
        for (int f = nFrames; f < RT_MAX_FRAMES; f++) pack.c[f] = camera[0];
        hipLaunchKernelGGL(store_cameras_kernel, dim3(1), dim3(64), 0, st, pack, w.d_cams);
        P.cams = w.d_cams;
    }
    P.nWork = nWork;
    P.rgb = d_rgb, P.primId = d_primId, P.primT = d_primT;
    P.stats = w.d_stats, P.work = w.d_work;
    P.accum = w.d_accum, P.nonfinite = w.d_nonfinite;
    P.shadeQ = static_cast<ShadeTask*>(w.d_shadeQ), P.leafQ = static_cast<LeafTask*>(w.d_leafQ);
    P.qCount = w.d_work + 8;
    P.shadeCap = w.shadeCap, P.leafCap = w.leafCap;
    // (the persistent schedules reserve a glossy loop's leaf slots all at once, at multiples of the loop's size: a queue that
    // ends at such a multiple has no reservation across its end, whose slots before the end nobody would write)
    if (usePath && features->numGlossySamples > 1) P.leafCap -= P.leafCap % features->numGlossySamples;
    w.lastLeafCap = P.leafCap;
    P.lightDraws = lightDraws;
    P.states = w.d_states, P.stateStride = stateStride;
    // (addressed by rayDepth: the levels below firstLevel are never touched and have no storage)
    P.frameBase = w.d_frames ? w.d_frames - (size_t)firstLevel * nWork * kFrameDwords : nullptr;
    P.passBase = (features->enableTransparency && w.d_passes) ? w.d_passes - (size_t)firstLevel * nWork * kPassDwords : nullptr;
    P.stateCap = nWork;
    P.pathRays = static_cast<RayRec*>(w.d_pathRays[0]), P.pathRaysIn = nullptr;
    P.rayCap = (uint32_t)(w.pathRayCap > 0xfffffff0ull ? 0xfffffff0ull : w.pathRayCap);
    P.batchRays = batchRays, P.batchWindow = batchWindow;
    P.hits = static_cast<const HitRec*>(w.d_hits);
    P.shadowRays = static_cast<RayRec*>(w.d_shadowRays);
    P.chunkBase = 0, P.chunkTasks = chunkTasks, P.raysPerTask = raysPerTask;
    const uint32_t taskGrid = (uint32_t)scene->numCUs * 8u;
    uint32_t resolveGrid = fullGrid < taskGrid ? fullGrid : taskGrid;
    const uint32_t rounds = (uint32_t)chain; // the camera rays are the first round
    uint64_t key = 1469598103934665603ull; // FNV-1a over what shapes the frame's rounds
    {
        auto mix = [&](const void* p, size_t n) {
            const unsigned char* b = static_cast<const unsigned char*>(p);
            for (size_t i = 0; i < n; i++) key = (key ^ b[i]) * 1099511628211ull;
        };
        const int32_t dims[6] = { width, height, startDepth, (int32_t)flags, (int32_t)batchRays, nFrames };
        mix(dims, sizeof(dims)), mix(features, sizeof(*features)), mix(&reg, sizeof(reg));
    }
    const bool planned = !countRef && key == scene->planKey && scene->roundRays.size() == rounds && !scene->noPlan;
    w.lastKey = key, w.lastRounds = usePath ? 0u : rounds;
    w.lastWork = nWork;

    TraceParams T;
    T.S = scene->view;
    T.hits = static_cast<HitRec*>(w.d_hits);
    T.stats = w.d_stats;
    T.useBvh = features->enableAccelStructure ? 1u : 0u;
    T.waveTimes = nullptr;
    int roundNow = 0; // the path round being launched (0: another stage)
    uint32_t nextFetch = 3 * kMaxRounds;
    uint32_t* const rayCounts = w.d_roundCounts + kMaxRounds;
    uint32_t* const backCounts = w.d_roundCounts + 2 * kMaxRounds; // states at the back of each round's active list (path_logic_kernel)
    const int stackClass = scene->treeDepth <= 16 ? 0 : (scene->treeDepth <= 24 ? 1 : 2);
    // LDS per workgroup = stack entries * 1 KiB: the grid keeps every CU's share of waves resident
    const uint32_t traceGrid = (uint32_t)scene->numCUs * ((overlapped || nFrames > 1) ? scene->laneBlocksPerCUPooled : scene->laneBlocksPerCU);
    // every ray of a group launch (at most laneModeAbove of them) gets its group from the grid position
    const uint32_t groupRaysPerBlock = kBlock / (scene->groupLanes > 1 ? scene->groupLanes : 8u);
    const uint32_t groupGrid = scene->groupBlocksPerCU ? (uint32_t)scene->numCUs * scene->groupBlocksPerCU
                                                       : (laneModeAbove + groupRaysPerBlock - 1) / groupRaysPerBlock;
    const int reserve = (int)scene->treeDepth + 2; // stack room a group keeps free (see group_node_step)
    uint32_t gridNow = groupGrid; // workgroups of the next group launch: fewer where few rays are expected (the rest is fetched in chunks)
    bool packetRound = false; // the stage being launched holds the camera rays in tile order and RT_AMD_PACKETS asks for packets
    bool anyHit = false; // the stage being launched holds shadow rays whose tag word is the shaded point's distance (see any_bound)
    auto launch_one = [&](uint32_t groupLanes, bool query) {
        T.fetch = w.d_roundCounts + nextFetch++;
        T.S = query ? scene->queryView : scene->view;
        if (query) { // only with culling, without counters, with 8- or 16-lane groups
            const int qreserve = (int)scene->queryDepth + 2;
            // (RT_AMD_GROUP_NODE4) eight lanes per ray in the tree's 4-wide form, where its stack bound fits the group's stack
            const int qreserve4 = 3 * ((int)scene->queryDepth4 + 1) + 3;
            if (scene->groupNode4 && groupLanes == 8 && !count && !scene->quadTrace && qreserve4 + 4 <= GroupLayout<8, false>::kQ) {
                if (anyHit) hipLaunchKernelGGL((group_trace_kernel<true, false, 8, true, true, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve4);
                else hipLaunchKernelGGL((group_trace_kernel<true, false, 8, true, false, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve4);
                return;
            }
            if (anyHit && !scene->quadTrace) {
                if (count) {
                    if (groupLanes == 8) hipLaunchKernelGGL((group_trace_kernel<true, true, 8, true, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
                    else if (groupLanes == 16)
                        hipLaunchKernelGGL((group_trace_kernel<true, true, 16, true, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
                    else hipLaunchKernelGGL((trace_kernel<true, true, kQueryStack, true, true>), dim3(traceGrid), dim3(kBlock), 0, st, T);
                    return;
                }
                if (groupLanes == 8) hipLaunchKernelGGL((group_trace_kernel<true, false, 8, true, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
                else if (groupLanes == 16)
                    hipLaunchKernelGGL((group_trace_kernel<true, false, 16, true, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
                else hipLaunchKernelGGL((trace_kernel<true, false, kQueryStack, true, true>), dim3(traceGrid), dim3(kBlock), 0, st, T);
                return;
            }
            if (groupLanes >= 8 && scene->quadTrace && scene->queryView.quadNodes && !count) { // four lanes per ray instead
                hipLaunchKernelGGL(quad_trace_kernel, dim3(std::max(1u, gridNow * 4u / groupLanes)), dim3(kBlock), 0, st, T);
                return;
            }
            if (count) {
                if (groupLanes == 8) hipLaunchKernelGGL((group_trace_kernel<true, true, 8, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
                else if (groupLanes == 16)
                    hipLaunchKernelGGL((group_trace_kernel<true, true, 16, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
                else hipLaunchKernelGGL((trace_kernel<true, true, kQueryStack, true>), dim3(traceGrid), dim3(kBlock), 0, st, T);
                return;
            }
            if (groupLanes == 8) hipLaunchKernelGGL((group_trace_kernel<true, false, 8, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
            else if (groupLanes == 16)
                hipLaunchKernelGGL((group_trace_kernel<true, false, 16, true>), dim3(gridNow), dim3(kBlock), 0, st, T, qreserve);
            else hipLaunchKernelGGL((trace_kernel<true, false, kQueryStack, true>), dim3(traceGrid), dim3(kBlock), 0, st, T);
            return;
        }
#define RT_TRACE_CASE(CULL, COUNT)                                                                                        \
    do {                                                                                                                  \
        if (groupLanes == 4)                                                                                              \
            hipLaunchKernelGGL((group_trace_kernel<CULL, COUNT, 4>), dim3(gridNow), dim3(kBlock), 0, st, T, reserve);   \
        else if (groupLanes == 8)                                                                                         \
            hipLaunchKernelGGL((group_trace_kernel<CULL, COUNT, 8>), dim3(gridNow), dim3(kBlock), 0, st, T, reserve);   \
        else if (groupLanes == 16)                                                                                        \
            hipLaunchKernelGGL((group_trace_kernel<CULL, COUNT, 16>), dim3(gridNow), dim3(kBlock), 0, st, T, reserve);  \
        else if (groupLanes == 32)                                                                                        \
            hipLaunchKernelGGL((group_trace_kernel<CULL, COUNT, 32>), dim3(gridNow), dim3(kBlock), 0, st, T, reserve);  \
        else if (stackClass == 0) hipLaunchKernelGGL((trace_kernel<CULL, COUNT, 16>), dim3(traceGrid), dim3(kBlock), 0, st, T); \
        else if (stackClass == 1) hipLaunchKernelGGL((trace_kernel<CULL, COUNT, 24>), dim3(traceGrid), dim3(kBlock), 0, st, T); \
        else hipLaunchKernelGGL((trace_kernel<CULL, COUNT, 32>), dim3(traceGrid), dim3(kBlock), 0, st, T);                \
    } while (0)
        if (cull && !count) RT_TRACE_CASE(true, false);
        else if (cull && count) RT_TRACE_CASE(true, true);
        else if (!cull && !count) RT_TRACE_CASE(false, false);
        else RT_TRACE_CASE(false, true);
#undef RT_TRACE_CASE
    };
    // One stage of closest-hit queries.  Many rays: one ray per lane (throughput), and the rays still unfinished
    // after `stepBudget` steps start again with several lanes per ray, so that the stage does not wait for its few
    // longest rays at one-lane speed.  Few rays: several lanes per ray from the start.  The ray count is only
    // known on the device, so every variant is launched and those whose range does not hold the count return at once.
    // With a query tree the rays walk that tree first; the few whose result is not provably the reference's are
    // traced again in the reference's tree.
    // order: 0 = the rays are traced as they lie; 1 / 2 = a deferred stage, whose rays are first put into a traversal order (by origin
    // cell first / by direction first) when there are many of them
    auto launch_trace = [&](const void* rays, uint32_t stride, const uint32_t* countPtr, uint32_t sub, uint32_t maxItems, uint32_t mul, bool smallRound = false,
                            uint32_t expected = 0, int order = 0) {
        T.rays = static_cast<const char*>(rays), T.rayStride = stride;
        T.countPtr = countPtr, T.countSub = sub, T.countMaxItems = maxItems, T.countMul = mul;
        T.indirect = nullptr, T.indirectFrom = 0, T.stepBudget = 0, T.longList = w.d_longList, T.longCount = nullptr;
        if (order && scene->orderRaysFrom && groupOk && !count) {
            OrderParams O;
            O.rays = T.rays, O.rayStride = stride;
            O.countPtr = countPtr, O.countSub = sub, O.countMaxItems = maxItems, O.countMul = mul;
            O.minRays = scene->orderRaysFrom;
            for (int a = 0; a < 3; a++) {
                const float ext = scene->view.rootHi[a] - scene->view.rootLo[a];
                O.lo[a] = scene->view.rootLo[a], O.inv[a] = ext > 0.0f ? 1.0f / ext : 0.0f;
            }
            O.dirMajor = order == 2 ? 1u : 0u;
            O.keys = w.d_orderKeys, O.hist = w.d_orderHist, O.order = w.d_order;
            (void)hipMemsetAsync(w.d_orderHist, 0, kOrderBuckets * sizeof(uint32_t), st);
            hipLaunchKernelGGL(order_key_kernel, dim3(taskGrid), dim3(kBlock), 0, st, O);
            hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, st, O);
            hipLaunchKernelGGL(order_scatter_kernel, dim3(taskGrid), dim3(kBlock), 0, st, O);
            T.indirect = w.d_order, T.indirectFrom = scene->orderRaysFrom;
        }

        T.fallList = w.d_fallList, T.fallCount = nullptr;
        T.runLo = 0, T.runHi = 0xffffffffu;
        if (!groupOk) {
            launch_one(1, false);
            return;
        }
        w.longSlots.push_back(nextFetch);
        uint32_t* longCount = w.d_roundCounts + nextFetch++;
        if (useQuery) w.fallSlots.push_back(nextFetch);
        uint32_t* fallCount = w.d_roundCounts + nextFetch++;
        T.fallCount = fallCount;
        if (packetRound && useQuery && 3 * scene->queryDepth4 + 1 <= (uint32_t)kPacketStack) { // (a packet whose stack overflows all the same is traced again)
            // camera rays, tile by tile: one walk per wave for 64 rays (packet_trace_kernel), whatever their number; what it
            // cannot settle goes to the fall-back launch below like every query-tree launch's leftovers
            T.S = scene->queryView;
            T.fetch = w.d_roundCounts + nextFetch++;
            if (count) hipLaunchKernelGGL(packet_trace_kernel<true>, dim3(traceGrid), dim3(kBlock), 0, st, T);
            else hipLaunchKernelGGL(packet_trace_kernel<false>, dim3(traceGrid), dim3(kBlock), 0, st, T);
            T.runLo = 0, T.runHi = 0xffffffffu;
            T.indirect = w.d_fallList, T.indirectFrom = 0;
            T.countPtr = fallCount, T.countSub = 0, T.countMaxItems = 0xffffffffu, T.countMul = 1;
            gridNow = std::min(groupGrid, 256u);
            launch_one(scene->groupLanes, false);
            gridNow = groupGrid;
            return;
        }
        T.runLo = laneModeAbove;
        const uint32_t budget = (countRef || smallRound) ? 0u : (useQuery ? queryStepBudget : scene->stepBudget);
        T.stepBudget = budget, T.longCount = longCount;
        if (!smallRound) launch_one(1, useQuery); // the plan says few rays: the group launch takes them, however many there are
        T.runLo = 0, T.runHi = smallRound ? 0xffffffffu : laneModeAbove, T.stepBudget = 0, T.longCount = nullptr;
        // a round of few rays is bound by its longest ray: 16 lanes per ray shorten that chain (more rays: fewer rays per wave cost more)
        const bool wide = smallRound && useQuery && scene->groupLanes == 8 && expected < scene->wideBelow;
        const uint32_t rpb = wide ? groupRaysPerBlock / 2 : groupRaysPerBlock;
        if (smallRound) gridNow = std::min(wide ? 2 * groupGrid : groupGrid, std::max(256u, ((2u * expected + rpb - 1) / rpb + 7u) & ~7u));
        if (roundNow != 0 && roundNow == scene->waveTimesRound) { // diagnostics: this launch's waves note their start and end
            const uint32_t waves = gridNow * (kBlock / 64);
            if (w.waveTimesWaves < waves) {
                (void)hipFree(w.d_waveTimes);
                w.d_waveTimes = nullptr, w.waveTimesWaves = 0;
                if (hipMalloc(reinterpret_cast<void**>(&w.d_waveTimes), (size_t)waves * 16) == hipSuccess) w.waveTimesWaves = waves;
            }
            if (w.d_waveTimes) {
                (void)hipMemsetAsync(w.d_waveTimes, 0, (size_t)w.waveTimesWaves * 16, st);
                T.waveTimes = w.d_waveTimes;
            }
        }
        launch_one(wide ? 16u : scene->groupLanes, useQuery);
        T.waveTimes = nullptr;
        gridNow = groupGrid;
        if (budget) { // the handed-over rays (none unless the first launch ran)
            T.runLo = 0, T.runHi = 0xffffffffu;
            T.indirect = w.d_longList, T.indirectFrom = 0;
            T.countPtr = longCount, T.countSub = 0, T.countMaxItems = 0xffffffffu, T.countMul = 1;
            const uint32_t budget2 = useQuery ? (overlapped ? scene->secondPassBudgetOverlapped : scene->secondPassBudget) : 0u;
            if (budget2) {
                // Second pass, again one ray per lane: the handed-over rays are the tail of the first launch, about a tenth
                // of its rays, and packed into full waves they cost a fifth of what they cost with several lanes per ray.
                // Every ray is a lane's first here, so the budget is a cap: what is still unfinished after it goes on to the
                // group launch.
                uint32_t* longCount2 = w.d_roundCounts + nextFetch++;
                T.stepBudget = budget2, T.longList = w.d_longList2, T.longCount = longCount2;
                launch_one(1, useQuery);
                T.stepBudget = 0, T.longList = w.d_longList, T.longCount = nullptr;
                T.indirect = w.d_longList2, T.countPtr = longCount2;
            }
            launch_one(scene->groupLanes, useQuery);
        }
        if (useQuery) { // the unsettled rays, in the reference's tree
            T.runLo = 0, T.runHi = 0xffffffffu;
            T.indirect = w.d_fallList, T.indirectFrom = 0;
            T.countPtr = fallCount, T.countSub = 0, T.countMaxItems = 0xffffffffu, T.countMul = 1;
            gridNow = std::min(groupGrid, 256u); // a handful of rays as a rule
            launch_one(scene->groupLanes, false);
            gridNow = groupGrid;
        }
    };

    P.activeIn = nullptr, P.activeInCount = nullptr;
    P.activeInCountBack = nullptr, P.activeOutCountBack = backCounts; // (the camera kernel fills the front only)
    P.activeOut = w.d_active[0], P.activeOutCount = w.d_roundCounts;
    P.activeRayIn = nullptr, P.activeRayOut = w.d_activeRay[0], P.rayCountOut = rayCounts;
    hipLaunchKernelGGL(camera_kernel, dim3(taskGrid < fullGrid ? taskGrid : fullGrid), dim3(kBlock), 0, st, P);
    if (usePathLane) {
        PathKParams K;
        const uint32_t N = features->numGlossySamples ? features->numGlossySamples : 1u;
        K.shadeChunk = 256u, K.leafChunk = N * 64u;
        K.advanceAt = scene->advanceAt, K.advanceEvery = scene->advanceEvery;
        K.scatter = scene->scatter ? 1u : 0u;
        K.contList = nullptr, K.contCount = nullptr, K.linkBudget = 0, K.minLanes = 0;
        const uint32_t pathGrid = (uint32_t)scene->numCUs * scene->pathBlocksPerCU;
        const bool query = features->enableAccelStructure && scene->queryView.cert != nullptr && !(flags & RT_RENDER_REFERENCE_TREE);
        const uint32_t suspSlot = nextFetch++;
        K.T = query ? scene->queryView : scene->view;
        K.list = nullptr, K.countPtr = w.d_roundCounts; // the camera kernel's states, one per pixel with a ray to trace
        K.fetch = w.d_roundCounts + nextFetch++;
        K.suspList = query ? w.d_fallList : nullptr, K.suspCount = query ? w.d_roundCounts + suspSlot : nullptr;
        K.reserve = 0;
        K.dbg = nullptr;
#ifdef RT_PATH_PROFILE
        K.dbg = reinterpret_cast<unsigned long long*>(w.d_activeRay[1]);
        (void)hipMemsetAsync(K.dbg, 0, 16 * 4 * pathGrid, st);
        w.dbgWaves = 4 * pathGrid;
#endif
        uint32_t grid = pathGrid;
        auto launch_ref = [&]() {
            if (stackClass == 0) hipLaunchKernelGGL((lane_path_kernel<16, false>), dim3(grid), dim3(kBlock), 0, st, P, K);
            else if (stackClass == 1) hipLaunchKernelGGL((lane_path_kernel<24, false>), dim3(grid), dim3(kBlock), 0, st, P, K);
            else hipLaunchKernelGGL((lane_path_kernel<32, false>), dim3(grid), dim3(kBlock), 0, st, P, K);
        };
        // slices: every launch but the last sets unfinished pixels aside after linkBudget rays (or when its wave runs thin)
        // and the next one packs them into full waves again; the last launch walks what is left to its end
        uint64_t chainLinks = 1;
        {
            const uint64_t fan = (features->enableReflections && features->enableGlossyReflection) ? std::max(1u, features->numGlossySamples) : 1;
            uint64_t term = 1;
            chainLinks = 0;
            for (int l = startDepth; l <= RT_MAX_RAY_DEPTH && chainLinks < (1u << 20); l++) {
                chainLinks += term;
                term *= fan;
                if (!features->enableReflections) break;
            }
            chainLinks *= std::max<uint64_t>(1, features->numPixelSamples) + 3; // jittered sampling rounds the sample count up
        }
        const uint32_t linkBudget = std::max<uint32_t>(scene->linkBudget, (uint32_t)(chainLinks / 48u) + 1u);
        const uint32_t slices = scene->linkBudget ? (uint32_t)std::min<uint64_t>(chainLinks / linkBudget + 2u, 64u) : 1u;
        K.linkBudget = linkBudget, K.minLanes = scene->minLanes;
        const int quads = (query && scene->queryView.quadNodes) ? scene->pathQuads : 0;
        for (uint32_t sl = 0; sl < slices; sl++) {
            const bool last = sl + 1 == slices;
            const uint32_t contSlot = nextFetch++;
            if (sl == 0) w.sliceSlots.clear();
            w.sliceSlots.push_back(contSlot);
            K.contList = last ? nullptr : w.d_active[(sl + 1) & 1], K.contCount = last ? nullptr : w.d_roundCounts + contSlot;
            K.fetch = w.d_roundCounts + nextFetch++;
            if (!query) launch_ref();
            else if (quads == 1) hipLaunchKernelGGL((quad_path_kernel<true>), dim3(grid), dim3(kBlock), 0, st, P, K);
            else if (quads == 2) hipLaunchKernelGGL((quad_path_kernel<false>), dim3(grid), dim3(kBlock), 0, st, P, K);
            else hipLaunchKernelGGL((lane_path_kernel<16, true>), dim3(grid), dim3(kBlock), 0, st, P, K);
            K.list = w.d_active[(sl + 1) & 1], K.countPtr = w.d_roundCounts + contSlot;
        }
        if (query) {
            // the pixels that met a ray the query tree could not answer with proof go on in the reference's tree
            w.fallSlots.push_back(suspSlot);
            K.T = scene->view;
            K.list = w.d_fallList, K.countPtr = w.d_roundCounts + suspSlot;
            K.fetch = w.d_roundCounts + nextFetch++;
            K.suspList = nullptr, K.suspCount = nullptr, K.contList = nullptr, K.contCount = nullptr;
            grid = std::min(pathGrid, 256u); // a handful of pixels as a rule; any number is walked
            launch_ref();
        }
    } else if (usePathGroup) {
        PathKParams K;
        K.advanceAt = K.advanceEvery = 0;
        K.dbg = nullptr, K.scatter = 0;
        K.contList = nullptr, K.contCount = nullptr, K.linkBudget = 0, K.minLanes = 0;
        const uint32_t N = features->numGlossySamples ? features->numGlossySamples : 1u;
        K.shadeChunk = 128u, K.leafChunk = N * 32u;
        const uint32_t pathGrid = (uint32_t)scene->numCUs * scene->pathBlocksPerCU;
        const uint32_t suspSlot = nextFetch++;
        K.T = useQuery ? scene->queryView : scene->view;
        K.list = nullptr, K.countPtr = w.d_roundCounts; // the camera kernel's states, one per pixel with a ray to trace
        K.fetch = w.d_roundCounts + nextFetch++;
        K.suspList = useQuery ? w.d_fallList : nullptr, K.suspCount = useQuery ? w.d_roundCounts + suspSlot : nullptr;
        K.reserve = (int)(useQuery ? scene->queryDepth : scene->treeDepth) + 2;
        const bool wide = scene->groupLanes == 16;
        if (useQuery) {
            if (wide) hipLaunchKernelGGL((path_kernel<16, true>), dim3(pathGrid), dim3(kBlock), 0, st, P, K);
            else hipLaunchKernelGGL((path_kernel<8, true>), dim3(pathGrid), dim3(kBlock), 0, st, P, K);
            // the pixels that met a ray the query tree could not answer with proof go on in the reference's tree
            w.fallSlots.push_back(suspSlot);
            K.T = scene->view;
            K.list = w.d_fallList, K.countPtr = w.d_roundCounts + suspSlot;
            K.fetch = w.d_roundCounts + nextFetch++;
            K.suspList = nullptr, K.suspCount = nullptr;
            K.reserve = (int)scene->treeDepth + 2;
            const uint32_t grid2 = std::min(pathGrid, 256u); // a handful of pixels as a rule; any number is walked
            if (wide) hipLaunchKernelGGL((path_kernel<16, false>), dim3(grid2), dim3(kBlock), 0, st, P, K);
            else hipLaunchKernelGGL((path_kernel<8, false>), dim3(grid2), dim3(kBlock), 0, st, P, K);
        } else {
            if (wide) hipLaunchKernelGGL((path_kernel<16, false>), dim3(pathGrid), dim3(kBlock), 0, st, P, K);
            else hipLaunchKernelGGL((path_kernel<8, false>), dim3(pathGrid), dim3(kBlock), 0, st, P, K);
        }
    }
    for (uint32_t r = 1; r <= rounds && !usePath; r++) {
        // the camera kernel emits exactly one ray per active state and counts them once
        // a round counts as small when the plan's frame would have sent it to the group kernel (which is about as fast as the
        // other schedule up to half as many rays again, should this frame's round be larger)
        const bool smallRound = planned && scene->roundRays[r - 1] < laneModeAbove;
        packetRound = r == 1 && scene->packets && features->numPixelSamples <= 1;
        roundNow = (int)r;
        launch_trace(w.d_pathRays[(r - 1) & 1], sizeof(RayRec), r == 1 ? w.d_roundCounts : rayCounts + (r - 1), 0, P.rayCap, 1, smallRound,
                     smallRound ? scene->roundRays[r - 1] : 0u);
        packetRound = false;
        roundNow = 0;
        P.pathRaysIn = static_cast<const RayRec*>(w.d_pathRays[(r - 1) & 1]), P.pathRays = static_cast<RayRec*>(w.d_pathRays[r & 1]);
        P.activeIn = w.d_active[(r - 1) & 1], P.activeInCount = w.d_roundCounts + (r - 1);
        P.activeOut = w.d_active[r & 1], P.activeOutCount = w.d_roundCounts + r;
        P.activeInCountBack = backCounts + (r - 1), P.activeOutCountBack = backCounts + r;
        P.activeRayIn = w.d_activeRay[(r - 1) & 1], P.activeRayOut = w.d_activeRay[r & 1], P.rayCountOut = rayCounts + r;
        const uint32_t logicGrid = (uint32_t)scene->numCUs * scene->logicBlocksPerCU;
        if (features->enableTransparency) hipLaunchKernelGGL(path_logic_kernel<true>, dim3(logicGrid), dim3(kBlock), 0, st, P);
        else hipLaunchKernelGGL(path_logic_kernel<false>, dim3(logicGrid), dim3(kBlock), 0, st, P);
    }
    // what the batched loops left for later: their children's hit records and the grandchildren's rays
    if (batchRays) hipLaunchKernelGGL(shade_expand_kernel, dim3(taskGrid), dim3(kBlock), 0, st, P);
    launch_trace(w.d_leafQ, sizeof(LeafTask), P.qCount + 1, 0, P.leafCap, 1, false, 0, 1);
    hipLaunchKernelGGL(leaf_logic_kernel, dim3(taskGrid), dim3(kBlock), 0, st, P);
    for (uint32_t c = 0; c < nChunks; c++) {
        P.chunkBase = c * chunkTasks;
        P.chunkTasks = (w.shadeCap - P.chunkBase < chunkTasks) ? w.shadeCap - P.chunkBase : chunkTasks;
        if (shadows) {
            hipLaunchKernelGGL(shadow_gen_kernel, dim3(taskGrid), dim3(kBlock), 0, st, P);
            if (features->enableTransparency) { // every sample is followed through the surfaces on its way, in the caller's tree
                T.S = scene->view;
                T.fetch = w.d_roundCounts + nextFetch++;
#define RT_WALK_CASE(CULL, COUNT)                                                                                              \
    do {                                                                                                                       \
        if (stackClass == 0) hipLaunchKernelGGL((shadow_walk_kernel<CULL, COUNT, 16>), dim3(traceGrid), dim3(kBlock), 0, st, P, T);     \
        else if (stackClass == 1) hipLaunchKernelGGL((shadow_walk_kernel<CULL, COUNT, 24>), dim3(traceGrid), dim3(kBlock), 0, st, P, T); \
        else hipLaunchKernelGGL((shadow_walk_kernel<CULL, COUNT, 32>), dim3(traceGrid), dim3(kBlock), 0, st, P, T);            \
    } while (0)
                if (cull && !count) RT_WALK_CASE(true, false);
                else if (cull && count) RT_WALK_CASE(true, true);
                else if (!cull && !count) RT_WALK_CASE(false, false);
                else RT_WALK_CASE(false, true);
#undef RT_WALK_CASE
            } else {
                anyHit = !scene->noShadowBound;
                launch_trace(w.d_shadowRays, sizeof(RayRec), P.qCount + 0, P.chunkBase, P.chunkTasks, raysPerTask, false, 0, 2);
                anyHit = false;
            }
            P.hits = static_cast<const HitRec*>(w.d_hits);
        } else {
            P.hits = nullptr;
        }
        hipLaunchKernelGGL(shade_logic_kernel, dim3(taskGrid), dim3(kBlock), 0, st, P);
    }
    hipLaunchKernelGGL(resolve_kernel, dim3(resolveGrid), dim3(kBlock), 0, st, P);
    RT_HIP(hipGetLastError());
    w.lastWasWavefront = true;
    queue_status(w, st, nextFetch);
    return RT_OK;
}

int rt_render_frames(rt_scene* scene, const rt_camera* cameras, int nFrames, const rt_features* features, int width, int height, int startDepth,
                     const rt_region* region, uint32_t flags, float* d_rgb, uint32_t* d_primId, float* d_primT, void* stream)
{
    int rc = render_frames_impl(scene, cameras, nFrames, features, width, height, startDepth, region, flags, d_rgb, d_primId, d_primT, stream, false);
    if (rc != kNotPooled) return rc;
    // schedules that take one frame at a time (depth-first, a pixel per lane): the frames one after the other on the stream
    const size_t px = (size_t)width * height;
    for (int f = 0; f < nFrames; f++) {
        rc = render_frames_impl(scene, cameras + f, 1, features, width, height, startDepth, region, flags, d_rgb + 3 * px * f,
                                d_primId ? d_primId + px * f : nullptr, d_primT ? d_primT + px * f : nullptr, stream, f > 0);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_render(rt_scene* scene, const rt_camera* camera, const rt_features* features, int width, int height, int startDepth,
              const rt_region* region, uint32_t flags, float* d_rgb, uint32_t* d_primId, float* d_primT, void* stream)
{
    return render_frames_impl(scene, camera, 1, features, width, height, startDepth, region, flags, d_rgb, d_primId, d_primT, stream, false);
}

size_t rt_bloom_scratch_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)width * (size_t)height * 3 * sizeof(float);
}

int rt_bloom(float* d_rgb, float* d_scratch, int width, int height, uint32_t filterSize, float threshold, float intensity,
             const rt_region* region, void* stream)
{
    if (!d_rgb || !d_scratch || width <= 0 || height <= 0) {
        rt::set_error("rt_bloom: null argument or empty frame");
        return RT_ERR_INVALID;
    }
    if (filterSize > (uint32_t)kBloomMaxF) { // binomial() overflows beyond 2F = 62 (src/extra.cpp:358)
        rt::set_error("rt_bloom: filter size %u exceeds the limit of %d", filterSize, kBloomMaxF);
        return RT_ERR_LIMIT;
    }
    rt_region reg { 8, 0, 1 };
    if (region && region->world > 1) reg = *region;
    if (reg.stripeRows <= 0 || reg.rank < 0 || reg.rank >= reg.world) {
        rt::set_error("rt_bloom: invalid region");
        return RT_ERR_INVALID;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    BloomParams P;
    P.src = d_rgb, P.rgb = d_rgb, P.tmp = d_scratch;
    P.W = width, P.H = height, P.F = (int)filterSize;
    P.threshold = threshold, P.intensity = intensity;
    P.stripeRows = reg.stripeRows, P.rank = reg.rank, P.world = reg.world;
    gaussian_taps(filterSize, P.taps);
    hipLaunchKernelGGL(bloom_rows_kernel, dim3((width + kBloomRowTile - 1) / kBloomRowTile, height), dim3(kBloomRowTile), 0, st, P);
    RT_HIP(hipGetLastError());
    hipLaunchKernelGGL(bloom_cols_kernel, dim3((width + kBloomColX - 1) / kBloomColX, (height + kBloomColY - 1) / kBloomColY),
                       dim3(kBloomColX * 4), 0, st, P);
    RT_HIP(hipGetLastError());
    return RT_OK;
}

// rt_bloom for callers that cannot allocate: the row-filtered intermediate image lives in a file-scope device array.
// Each GPU holds its own copy of the array at its own address, looked up once per device; host code never uses the
// array's name for anything else (on the host the name is a stand-in, not a device address).
int rt_bloom_static(float* d_rgb, int width, int height, uint32_t filterSize, float threshold, float intensity, void* stream)
{
    if (!d_rgb || width <= 0 || height <= 0) {
        rt::set_error("rt_bloom_static: null argument or empty frame");
        return RT_ERR_INVALID;
    }
    if ((unsigned long long)width * (unsigned long long)height > RT_BLOOM_STATIC_MAX_PIXELS) {
        rt::set_error("rt_bloom_static: a frame of %dx%d pixels exceeds the static scratch array of %llu pixels - use rt_bloom with "
                      "a workspace of rt_bloom_scratch_bytes()",
                      width, height, (unsigned long long)RT_BLOOM_STATIC_MAX_PIXELS);
        return RT_ERR_LIMIT;
    }
    int device = -1;
    RT_HIP(hipGetDevice(&device));
    if (stream) { // the array used is the one of the GPU the kernels will run on: that GPU must be the current one
        hipDevice_t streamDevice = -1;
        RT_HIP(hipStreamGetDevice(static_cast<hipStream_t>(stream), &streamDevice));
        if ((int)streamDevice != device) {
            rt::set_error("rt_bloom_static: the stream belongs to device %d, the current device is %d", (int)streamDevice, device);
            return RT_ERR_INVALID;
        }
    }
    float* scratch = nullptr;
    const int rc = bloom_static_scratch(device, &scratch);
    if (rc != RT_OK) return rc;
    // pass 1 writes every pixel of the intermediate image that pass 2 reads: nothing of an earlier call is read
    return rt_bloom(d_rgb, scratch, width, height, filterSize, threshold, intensity, nullptr, stream);
}

// What rt_sync reports for the last frame queued with the buffers `w` (whose stream has been waited for): its counters,
// and whether it is complete.  Also notes the frame's round sizes for the next frame's launch plan.
static int collect_frame(rt_scene* scene, Workspace& w, rt_stats* stats)
{
    (void)scene;
    unsigned long long h[S_COUNT] = {};
    unsigned long long carried[3] = {}; // S_QMAX_SHADE, S_QMAX_LEAF, S_FALL_SUM: earlier frames of a sequence rendered frame by frame
    // (the stream has been waited for: the copies the sequence ended with are complete)
    const bool copied = w.statusQueued && w.h_status != nullptr;
    auto read_counts = [&](const uint32_t lo, const size_t n, uint32_t* dst) -> hipError_t {
        if (copied && lo + n <= w.statusCountWords) {
            memcpy(dst, w.h_status + kStatusStatsBytes + kStatusWorkBytes + (size_t)lo * sizeof(uint32_t), n * sizeof(uint32_t));
            return hipSuccess;
        }
        return hipMemcpy(dst, w.d_roundCounts + lo, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    };
    {
        unsigned long long all[kStatCopies * kStatStride];
        if (copied) memcpy(all, w.h_status, sizeof(all));
        else RT_HIP(hipMemcpy(all, w.d_stats, sizeof(all), hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < kStatCopies; c++)
            for (int i = 0; i < S_COUNT; i++) h[i] += all[c * kStatStride + i];
        for (int i = 0; i < 3; i++) carried[i] = all[S_COUNT + i];
    }
#ifdef RT_PATH_PROFILE
    {
        unsigned long long all[kStatCopies * kStatStride];
        RT_HIP(hipMemcpy(all, w.d_stats, sizeof(all), hipMemcpyDeviceToHost));
        fprintf(stderr, "path_kernel slowest wave: total %llu cycles, fetch %llu, advance %llu, begin %llu, steps %llu; iterations %llu, advances %llu, step iterations %llu\n",
                all[16], all[17], all[18], all[19], all[20], all[21], all[22], all[23]);
        fprintf(stderr, "path_kernel: longest pixel chain %llu rays, longest ray %llu steps, %llu rays, %llu steps in all\n", all[24], all[25], all[26], all[27]);
        fprintf(stderr, "path_kernel: %llu waves, mean wave time %llu cycles, mean step iterations per wave %llu\n", all[29], all[28] / (all[29] ? all[29] : 1), all[30] / (all[29] ? all[29] : 1));
        if (w.dbgWaves) {
            std::vector<unsigned long long> tv(2 * w.dbgWaves);
            RT_HIP(hipMemcpy(tv.data(), w.d_activeRay[1], tv.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> us;
            unsigned long long start = ~0ull;
            for (uint32_t i = 0; i < w.dbgWaves; i++) if (tv[2 * i]) start = std::min(start, tv[2 * i]);
            std::vector<double> st_, du;
            for (uint32_t i = 0; i < w.dbgWaves; i++)
                if (tv[2 * i + 1]) us.push_back((double)(tv[2 * i + 1] - start)), st_.push_back((double)(tv[2 * i] - start)), du.push_back((double)(tv[2 * i + 1] - tv[2 * i]));
            std::sort(us.begin(), us.end()), std::sort(st_.begin(), st_.end()), std::sort(du.begin(), du.end());
            if (!us.empty()) {
                const double unit = 100.0 / us.back(); // percent of the kernel's duration
                const char* names[3] = { "end", "start", "duration" };
                std::vector<double>* vs[3] = { &us, &st_, &du };
                for (int k = 0; k < 3; k++) {
                    fprintf(stderr, "path_kernel: wave %s (%% of the launch), percentiles 10 25 50 75 90 95 99 100:", names[k]);
                    for (double q : { 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99, 1.0 }) fprintf(stderr, " %.1f", unit * (*vs[k])[std::min(vs[k]->size() - 1, (size_t)(q * vs[k]->size()))]);
                    fprintf(stderr, "\n");
                }
            }
        }
        fprintf(stderr, "path_kernel: pixels by chain length (bins of 4 rays):");
        for (int b = 0; b < 23; b++) fprintf(stderr, " %llu", all[32 + 9 + b]);
        fprintf(stderr, "\n");
        fprintf(stderr, "path_kernel: lanes per step iteration %.1f, advances %llu with %.1f lanes each; mean advance cycles %llu per wave, mean step cycles %llu per wave\n",
                (double)all[31] / (double)(all[30] ? all[30] : 1), all[15], (double)all[14] / (double)(all[15] ? all[15] : 1), all[13] / (all[29] ? all[29] : 1), all[12] / (all[29] ? all[29] : 1));
    }
#endif
    if (stats) {
        stats->primaryRays = h[S_PRIMARY], stats->shadowRays = h[S_SHADOW], stats->secondaryRays = h[S_SECONDARY];
        stats->nodePairSteps = h[S_NODEPAIRS], stats->triangleTests = h[S_TRITESTS], stats->queryNodeSteps = h[S_NODE4];
        stats->pixels = h[S_PIXELS], stats->stackOverflows = h[S_OVERFLOW], stats->referenceTreeQueries = 0;
        stats->pipeline = !w.lastWasWavefront ? RT_PIPELINE_DEPTH_FIRST : (w.lastRounds ? RT_PIPELINE_ROUNDS : RT_PIPELINE_PERSISTENT);
        if (w.lastWasWavefront && !w.fallSlots.empty()) {
            const uint32_t lo = w.fallSlots.front(), hi = w.fallSlots.back();
            std::vector<uint32_t> c(hi - lo + 1);
            RT_HIP(read_counts(lo, c.size(), c.data()));
            for (uint32_t slot : w.fallSlots) stats->referenceTreeQueries += c[slot - lo];
        }
        stats->referenceTreeQueries += carried[S_FALL_SUM - S_COUNT]; // earlier frames of a sequence rendered frame by frame
    }
    if (h[S_OVERFLOW]) {
        rt::set_error("traversal stack overflowed %llu times: results are incomplete", h[S_OVERFLOW]);
        return RT_ERR_LIMIT;
    }
    if (w.lastWasWavefront) { // did every deferred task fit its queue?  The counters keep counting past the capacity.
        uint32_t q[2];
        if (copied) memcpy(q, w.h_status + kStatusStatsBytes + 8 * sizeof(uint32_t), sizeof(q));
        else RT_HIP(hipMemcpy(q, w.d_work + 8, sizeof(q), hipMemcpyDeviceToHost));
        // (a sequence rendered frame by frame: the fullest queues any of its frames had)
        q[0] = (uint32_t)std::max<unsigned long long>(q[0], carried[S_QMAX_SHADE - S_COUNT]);
        q[1] = (uint32_t)std::max<unsigned long long>(q[1], carried[S_QMAX_LEAF - S_COUNT]);
        if (scene->verbose)
            fprintf(stderr, "rt_amd: task queues: %u shade tasks of %u, %u leaf tasks of %u, for %u pixels; path rays held %llu\n", q[0], w.shadeCap, q[1],
                    w.leafCap, w.lastWork, (unsigned long long)w.pathRayCap);
        if (w.lastWork) {
            rt_scene::SeenQueues& held = scene->seenQueues[w.lastCfg];
            held.shade = std::max(held.shade, (double)q[0] / (double)w.lastWork);
            held.leaf = std::max(held.leaf, (double)q[1] / (double)w.lastWork);
        }
        if (q[0] > w.shadeCap || q[1] > w.lastLeafCap) {
            // the next frame gets queues of the size this one asked for and a quarter more
            const bool canGrow = (q[0] > w.shadeCap && w.shadeCap < kMaxQueueTasks) || (q[1] > w.lastLeafCap && w.leafCap < kMaxQueueTasks);
            // (leaf tasks that found no room left no shade task behind: the shade queue is sized for every leaf ray hitting)
            const uint64_t most = std::max(q[0], q[1]);
            scene->wantShadeTasks = std::max<uint64_t>(scene->wantShadeTasks, most + most / 4);
            if (q[1] > w.lastLeafCap) scene->wantLeafTasks = std::max<uint64_t>(scene->wantLeafTasks, (uint64_t)q[1] + q[1] / 4);
            rt::set_error("task queues overflowed (%u shade tasks for %u slots, %u leaf tasks for %u slots): the frame is incomplete; %s",
                          q[0], w.shadeCap, q[1], w.leafCap,
                          canGrow ? "the queues grow with the next rt_render of this scene: render the frame again"
                                  : "render with RT_RENDER_SERIAL or in smaller regions");
            return canGrow ? RT_ERR_AGAIN : RT_ERR_LIMIT;
        }
    }
    std::vector<uint32_t> roundRays; // rays of the path rounds (the counters keep counting past the buffer's capacity)
    if (w.lastWasWavefront && w.lastRounds) {
        const uint32_t n = w.lastRounds;
        roundRays.assign(n, 0u);
        RT_HIP(read_counts(0, 1, &roundRays[0])); // round 1: the camera's rays
        if (n > 1) RT_HIP(read_counts(kMaxRounds + 1, n - 1, &roundRays[1]));
        uint32_t most = 0;
        for (uint32_t v : roundRays) most = std::max(most, v);
        if (w.lastWork) {
            rt_scene::SeenQueues& held = scene->seenQueues[w.lastCfg];
            held.rays = std::max(held.rays, (double)most / (double)w.lastWork);
        }
        if (most > w.pathRayCap) { // (sized from what earlier frames needed: this one needs more)
            scene->wantPathRays = std::max<uint64_t>(scene->wantPathRays, (uint64_t)most + most / 4);
            rt::set_error("a path round held %u rays for %llu places: the frame is incomplete; the buffers grow with the next rt_render of this "
                          "scene: render the frame again", most, (unsigned long long)w.pathRayCap);
            return RT_ERR_AGAIN;
        }
    }
    if (h[S_DROPPED]) {
        rt::set_error("%llu rays or tasks found no room in the frame's buffers: the frame is incomplete; render with RT_RENDER_SERIAL "
                      "or in smaller regions", h[S_DROPPED]);
        return RT_ERR_LIMIT;
    }
    if (h[S_BIGVALUE]) {
        rt::set_error("%llu finite contributions of magnitude >= 4e6 exceed the fixed-point pixel sums (recorded as infinite): render "
                      "with RT_RENDER_SERIAL", h[S_BIGVALUE]);
        return RT_ERR_LIMIT;
    }
    if (w.lastWasWavefront && w.lastRounds) { // a pixel still had rays to trace when the rounds ended
        uint32_t left = 0;
        RT_HIP(read_counts(w.lastRounds, 1, &left));
        {
            uint32_t back = 0; // the active list's other end
            RT_HIP(read_counts(2 * kMaxRounds + w.lastRounds, 1, &back));
            left += back;
        }
        if (left) {
            rt::set_error("%u pixels were unfinished after the frame's %u path rounds: the frame is incomplete; render with RT_RENDER_SERIAL",
                          left, w.lastRounds);
            return RT_ERR_LIMIT;
        }
    }
    if (w.lastWasWavefront && !w.sliceSlots.empty() && scene->verbose) {
        fprintf(stderr, "rt_amd: pixels set aside by the slices of the path stage:");
        for (uint32_t slot : w.sliceSlots) {
            uint32_t v = 0;
            RT_HIP(hipMemcpy(&v, w.d_roundCounts + slot, sizeof(v), hipMemcpyDeviceToHost));
            fprintf(stderr, " %u", v);
        }
        fprintf(stderr, "\n");
    }
    if (w.lastWasWavefront && scene->verbose && w.d_waveTimes && w.waveTimesWaves) {
        std::vector<unsigned long long> tv((size_t)w.waveTimesWaves * 2);
        RT_HIP(hipMemcpy(tv.data(), w.d_waveTimes, tv.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull, last = 0;
        for (uint32_t i = 0; i < w.waveTimesWaves; i++)
            if (tv[2 * i + 1]) first = std::min(first, tv[2 * i]), last = std::max(last, tv[2 * i + 1]);
        std::vector<double> ends, durs;
        for (uint32_t i = 0; i < w.waveTimesWaves; i++)
            if (tv[2 * i + 1]) ends.push_back((double)(tv[2 * i + 1] - first)), durs.push_back((double)(tv[2 * i + 1] - tv[2 * i]));
        std::sort(ends.begin(), ends.end()), std::sort(durs.begin(), durs.end());
        if (!ends.empty()) {
            // (the clock ticks at 100 MHz)
            fprintf(stderr, "rt_amd: round %d group launch: %zu waves over %.1f us; waves ended by (us), percentiles 10 25 50 75 90 95 99 100:", scene->waveTimesRound,
                    ends.size(), (double)(last - first) / 100.0);
            for (double q : { 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99, 1.0 }) fprintf(stderr, " %.1f", ends[std::min(ends.size() - 1, (size_t)(q * ends.size()))] / 100.0);
            fprintf(stderr, "; wave lifetimes (us), same percentiles:");
            for (double q : { 0.10, 0.25, 0.50, 0.75, 0.90, 0.95, 0.99, 1.0 }) fprintf(stderr, " %.1f", durs[std::min(durs.size() - 1, (size_t)(q * durs.size()))] / 100.0);
            fprintf(stderr, "\n");
        }
    }
    if (w.lastWasWavefront && scene->verbose && !w.longSlots.empty()) {
        const uint32_t lo = w.longSlots.front(), hi = w.longSlots.back();
        std::vector<uint32_t> c(hi - lo + 1);
        RT_HIP(hipMemcpy(c.data(), w.d_roundCounts + lo, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (uint32_t slot : w.longSlots) sum += c[slot - lo];
        fprintf(stderr, "rt_amd: rays handed over by one-ray-per-lane launches, stage by stage (%llu in all):", sum);
        for (uint32_t slot : w.longSlots) fprintf(stderr, " %u", c[slot - lo]);
        fprintf(stderr, "\n");
    }
    if (w.lastWasWavefront && w.lastRounds) { // the next frame's plan
        const uint32_t n = w.lastRounds;
        scene->roundRays = roundRays;
        scene->planKey = w.lastKey;
        if (scene->verbose) { // rays traced by the path rounds (candidate batches included), round by round
            unsigned long long sum = 0;
            for (uint32_t v : roundRays) sum += v;
            fprintf(stderr, "rt_amd: %u path rounds traced %llu rays:", n, sum);
            for (uint32_t v : roundRays) fprintf(stderr, " %u", v);
            fprintf(stderr, "\n");
        }
    }
    return RT_OK;
}

int rt_sync(rt_scene* scene, void* stream, rt_stats* stats)
{
    if (!scene) {
        rt::set_error("rt_sync: null scene");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    Workspace* const wsp = workspace_for(scene, static_cast<hipStream_t>(stream), false);
    if (!wsp) { // no frame on this stream, or its buffers have gone to another stream since: what was noted then
        if (stats) memset(stats, 0, sizeof(*stats));
        for (const RetiredFrame& r : scene->retired)
            if (r.stream == static_cast<hipStream_t>(stream)) {
                if (stats) *stats = r.stats;
                if (r.status != RT_OK) rt::set_error("%s", r.message.c_str());
                return r.status;
            }
        return RT_OK;
    }
    return collect_frame(scene, *wsp, stats);
}

int rt_render_image(rt_scene* scene, const rt_camera* camera, const rt_features* features, int width, int height, int startDepth,
                    uint32_t flags, float* rgb, uint32_t* primId, float* primT, rt_stats* stats)
{
    if (!scene || !rgb || !features || width <= 0 || height <= 0) {
        rt::set_error("rt_render_image: null argument or empty frame");
        return RT_ERR_INVALID;
    }
    // the image alone: the camera loop's route for one camera (the frame leaves through pinned memory, copied out by several threads)
    if (!primId && !primT) return rt_render_images(scene, camera, 1, features, width, height, startDepth, flags, &rgb, stats);
    RT_HIP(hipSetDevice(scene->device));
    const size_t px = (size_t)width * height;
    if (px > scene->fbPixels) {
        (void)hipFree(scene->d_rgb), (void)hipFree(scene->d_scratch), (void)hipFree(scene->d_primId), (void)hipFree(scene->d_primT);
        scene->d_rgb = scene->d_scratch = nullptr, scene->d_primId = nullptr, scene->d_primT = nullptr;
        scene->fbPixels = 0;
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_rgb), px * 12));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_scratch), px * 12));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_primId), px * 4));
        RT_HIP(hipMalloc(reinterpret_cast<void**>(&scene->d_primT), px * 4));
        scene->fbPixels = px;
    }
    int rc = rt_render(scene, camera, features, width, height, startDepth, nullptr, flags, scene->d_rgb, primId ? scene->d_primId : nullptr,
                       primT ? scene->d_primT : nullptr, nullptr);
    if (rc == RT_ERR_LIMIT && !(flags & RT_RENDER_SERIAL)) // a frame beyond the default pipeline's limits: evaluate it depth-first instead
        return rt_render_image(scene, camera, features, width, height, startDepth, flags | RT_RENDER_SERIAL, rgb, primId, primT, stats);
    if (rc != RT_OK) return rc;
    if (features->enableBloomEffect) { // src/render.cpp:48-50
        rc = rt_bloom(scene->d_rgb, scene->d_scratch, width, height, features->bloomFilterSize, features->bloomFilterThreshold,
                      features->bloomFilterIntensity, nullptr, nullptr);
        if (rc != RT_OK) return rc;
    }
    rc = rt_sync(scene, nullptr, stats);
    for (int again = 0; rc == RT_ERR_AGAIN && again < 3; again++) { // queues too small for this frame: they have grown, render it again
        rc = rt_render(scene, camera, features, width, height, startDepth, nullptr, flags, scene->d_rgb, primId ? scene->d_primId : nullptr,
                       primT ? scene->d_primT : nullptr, nullptr);
        if (rc == RT_OK && features->enableBloomEffect)
            rc = rt_bloom(scene->d_rgb, scene->d_scratch, width, height, features->bloomFilterSize, features->bloomFilterThreshold,
                          features->bloomFilterIntensity, nullptr, nullptr);
        if (rc == RT_OK) rc = rt_sync(scene, nullptr, stats);
    }
    if ((rc == RT_ERR_LIMIT || rc == RT_ERR_AGAIN) && !(flags & RT_RENDER_SERIAL)) // beyond the default pipeline's limits: evaluate it depth-first instead
        return rt_render_image(scene, camera, features, width, height, startDepth, flags | RT_RENDER_SERIAL, rgb, primId, primT, stats);
    if (rc != RT_OK) return rc;
    RT_HIP(hipMemcpy(rgb, scene->d_rgb, px * 12, hipMemcpyDeviceToHost));
    if (primId) RT_HIP(hipMemcpy(primId, scene->d_primId, px * 4, hipMemcpyDeviceToHost));
    if (primT) RT_HIP(hipMemcpy(primT, scene->d_primT, px * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render_images(rt_scene* scene, const rt_camera* cameras, int nFrames, const rt_features* features, int width, int height, int startDepth,
                     uint32_t flags, float* const* rgb, rt_stats* stats)
{
    if (!scene || !cameras || !rgb || !features || width <= 0 || height <= 0 || nFrames < 1) {
        rt::set_error("rt_render_images: null argument, no frame or empty frames");
        return RT_ERR_INVALID;
    }
    for (int f = 0; f < nFrames; f++)
        if (!rgb[f]) {
            rt::set_error("rt_render_images: frame %d has no image", f);
            return RT_ERR_INVALID;
        }
    RT_HIP(hipSetDevice(scene->device));
    const size_t px = (size_t)width * height;
    // The frames go as launch sequences of up to 8 frames and 16.6 M pixels (about 3 KB of task queues and path states per pixel),
    // up to four of them in flight on streams of their own within 41.5 M pixels in all (20 frames of 1920 x 1080: four sequences
    // of five): while one sequence's dependent rounds leave the chip half empty another fills it, and the finished frames of a
    // third are on their way to the host.
    const size_t kSequencePixels = (size_t)8 * 1920 * 1080, kFlightPixels = (size_t)20 * 1920 * 1080;
    // (more streams with shorter sequences before fewer with longer ones: 4 x 5 frames render faster than 2 x 8)
    int lanes = (int)std::max<size_t>(1, std::min<size_t>(std::min(nFrames, 4), kFlightPixels / px));
    int perSequence = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8, kSequencePixels / px), kFlightPixels / ((size_t)lanes * px)));
    perSequence = std::min(perSequence, std::max(1, 65535 / height)); // rows of a launch sequence (rt_render_frames)
    perSequence = std::min(perSequence, (nFrames + lanes - 1) / lanes);
    lanes = std::min(lanes, (nFrames + perSequence - 1) / perSequence);
    for (int k = 0; k < lanes; k++) {
        rt_scene::ImageLane& l = scene->imageLanes[k];
        void* made = nullptr;
        const int rcLane = rt_scene_stream(scene, k, &made);
        if (rcLane != RT_OK) return rcLane;
        if (l.px * l.frames < px * perSequence || l.px < px) {
            RT_HIP(hipStreamSynchronize(l.stream));
            (void)hipFree(l.d_rgb), (void)hipFree(l.d_scratch);
            if (l.h_stage) (void)hipHostFree(l.h_stage);
            l.d_rgb = l.d_scratch = l.h_stage = nullptr, l.frames = l.px = 0;
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&l.d_rgb), px * perSequence * 12));
            RT_HIP(hipMalloc(reinterpret_cast<void**>(&l.d_scratch), px * 12));
            RT_HIP(hipHostMalloc(reinterpret_cast<void**>(&l.h_stage), px * perSequence * 12, hipHostMallocDefault));
            l.frames = (size_t)perSequence, l.px = px;
        }
    }
    struct Pending {
        int first = 0, n = 0;
        uint32_t flags = 0;
        bool active = false;
    } pending[4];
    rt_stats total {};
    // queues lane k's sequence: render, bloom (src/render.cpp:48-50), copy to the lane's pinned buffer
    auto launch = [&](int k, int first, int n, uint32_t fl) -> int {
        rt_scene::ImageLane& l = scene->imageLanes[k];
        int rc = rt_render_frames(scene, cameras + first, n, features, width, height, startDepth, nullptr, fl, l.d_rgb, nullptr, nullptr, l.stream);
        for (int f = 0; rc == RT_OK && features->enableBloomEffect && f < n; f++)
            rc = rt_bloom(l.d_rgb + 3 * px * f, l.d_scratch, width, height, features->bloomFilterSize, features->bloomFilterThreshold,
                          features->bloomFilterIntensity, nullptr, l.stream);
        if (rc == RT_OK) RT_HIP(hipMemcpyAsync(l.h_stage, l.d_rgb, px * n * 12, hipMemcpyDeviceToHost, l.stream));
        return rc;
    };
    // waits for lane k's sequence, renders it again where the library asks for that, and hands its frames to the caller
    auto collect = [&](int k) -> int {
        Pending& p = pending[k];
        if (!p.active) return RT_OK;
        p.active = false;
        rt_scene::ImageLane& l = scene->imageLanes[k];
        rt_stats st {};
        int rc = rt_sync(scene, l.stream, &st);
        for (int attempt = 0; rc != RT_OK && attempt < 5; attempt++) {
            if (rc == RT_ERR_AGAIN && attempt < 3) { // the task queues have grown: render the sequence again
            } else if ((rc == RT_ERR_LIMIT || rc == RT_ERR_AGAIN) && !(p.flags & RT_RENDER_SERIAL)) { // beyond the default pipeline's limits: depth-first instead
                p.flags |= RT_RENDER_SERIAL;
            } else {
                break;
            }
            rc = launch(k, p.first, p.n, p.flags);
            if (rc == RT_OK) rc = rt_sync(scene, l.stream, &st);
        }
        if (rc != RT_OK) return rc;
        // out of the pinned buffer into the caller's images: one thread moves a 1080p frame in 1.4 ms, which a sweep's last
        // sequences would wait for frame after frame - several threads, a slice of every frame each
        const unsigned hw = std::thread::hardware_concurrency();
        const int workers = (px * p.n * 12 < ((size_t)4 << 20)) ? 1 : (int)std::min(8u, std::max(1u, hw));
        auto copySlice = [&](int t) {
            const size_t lo = px * 12 * (size_t)t / workers & ~(size_t)63, hi = (t + 1 == workers) ? px * 12 : (px * 12 * (size_t)(t + 1) / workers & ~(size_t)63);
            for (int f = 0; f < p.n; f++)
                memcpy(reinterpret_cast<char*>(rgb[p.first + f]) + lo, reinterpret_cast<const char*>(l.h_stage + 3 * px * f) + lo, hi - lo);
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < workers; t++) pool.emplace_back(copySlice, t);
        copySlice(0);
        for (std::thread& th : pool) th.join();
        total.primaryRays += st.primaryRays, total.shadowRays += st.shadowRays, total.secondaryRays += st.secondaryRays;
        total.nodePairSteps += st.nodePairSteps, total.triangleTests += st.triangleTests, total.queryNodeSteps += st.queryNodeSteps;
        total.pixels += st.pixels, total.stackOverflows += st.stackOverflows, total.referenceTreeQueries += st.referenceTreeQueries;
        total.pipeline = st.pipeline;
        return RT_OK;
    };
    auto drain = [&](int rc) { // an error: nothing of this call stays queued
        for (int k = 0; k < lanes; k++) (void)hipStreamSynchronize(scene->imageLanes[k].stream);
        return rc;
    };
    uint32_t fl = flags | (lanes > 1 ? (uint32_t)RT_RENDER_OVERLAPPED : 0u);
    int i = 0;
    for (int first = 0; first < nFrames; first += perSequence, i++) {
        const int k = i % lanes, n = std::min(perSequence, nFrames - first);
        int rc = collect(k); // the sequence this lane rendered before
        if (rc != RT_OK) return drain(rc);
        rc = launch(k, first, n, fl);
        if (rc == RT_ERR_LIMIT && !(fl & RT_RENDER_SERIAL)) { // refused before anything was queued (a recursion tree beyond the rounds' limit)
            fl |= RT_RENDER_SERIAL;
            rc = launch(k, first, n, fl);
        }
        if (rc != RT_OK) return drain(rc);
        pending[k].first = first, pending[k].n = n, pending[k].flags = fl, pending[k].active = true;
    }
    for (int j = 0; j < lanes; j++) { // in the order the sequences were queued
        const int rc = collect((i + j) % lanes);
        if (rc != RT_OK) return drain(rc);
    }
    if (stats) *stats = total;
    return RT_OK;
}

int rt_selftest_division(uint32_t seed, uint64_t* mismatches, uint64_t* checked)
{
    if (!mismatches || !checked) {
        rt::set_error("rt_selftest_division: null argument");
        return RT_ERR_INVALID;
    }
    unsigned long long* d = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&d), 512));
    hipError_t e = hipMemset(d, 0, 512);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(division_selftest_kernel, dim3(4096), dim3(kBlock), 0, nullptr, seed, 1024u, d);
        e = hipGetLastError();
    }
    unsigned long long h[64] = { 0 };
    if (e == hipSuccess) e = hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) {
        rt::set_error("rt_selftest_division: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    *mismatches = h[0], *checked = h[1];
    if (h[0]) // rt_last_error names the first mismatching operands (the call itself still succeeds)
        rt::set_error("rt_selftest_division: %llu mismatches, first: a=%08x d=%08x a/d=%08x reciprocal form=%08x", h[0],
                      (unsigned)(h[4] >> 32), (unsigned)h[4], (unsigned)(h[5] >> 32), (unsigned)h[5]);
    return RT_OK;
}

int rt_intersect(rt_scene* scene, const rt_features* features, float* rays, uint32_t nRays, void* hits)
{
    if (!scene || !features || (nRays && (!rays || !hits))) {
        rt::set_error("rt_intersect: null argument");
        return RT_ERR_INVALID;
    }
    if (nRays == 0) return RT_OK;
    RT_HIP(hipSetDevice(scene->device));
    float* d_rays = nullptr;
    uint32_t* d_hits = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void**>(&d_rays), (size_t)nRays * 28));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_hits), (size_t)nRays * 48);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)nRays * 28, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        IntersectParams P;
        P.S = scene->view;
        P.F = *features;
        P.rays = d_rays, P.hits = d_hits, P.n = nRays;
        hipLaunchKernelGGL(intersect_kernel, dim3((nRays + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, P);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(rays, d_rays, (size_t)nRays * 28, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hits, d_hits, (size_t)nRays * 48, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays);
    (void)hipFree(d_hits);
    if (e != hipSuccess) {
        rt::set_error("rt_intersect: %s", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    return RT_OK;
}

} // extern "C"
