// The loop over cameras (reference src/main.cpp:515-526) on several GPUs from ONE process.
//
// Pixels are independent in the reference's render loop (src/render.cpp:27-44) and a pixel's sampler seed depends on its own
// (x, y) only (src/render.cpp:38), so a frame is cut into stripes of 8 image rows, stripe s belongs to participant s % N
// (rt_region), and the participants render their stripes with no exchange.  Each participant is an rt_scene on some GPU
// (rt_scene_clone puts a scene on another GPU) driven by one host thread.  Frame f of a launch sequence is completed on
// participant f % N: it copies every other participant's stripes of that frame into its own copy of the frame (one peer copy
// per stripe, GPU to GPU), applies the bloom (src/render.cpp:48-50) - whose filter reaches across stripes, so it runs on the
// whole frame - and hands the frame to the caller through pinned memory.  Every frame therefore crosses to the host once, from
// one GPU, and bloom runs once per frame, as in the reference's loop.
//
// The threads meet on the host, once per launch sequence: when every participant's stripes of sequence q are rendered, the
// frames' roots start pulling them - on a second stream, while the next sequence is rendered into the other of two frame
// buffers.  A buffer is rendered into again two sequences later, after another meeting: every root has finished pulling by
// then.  Nothing queued on one GPU waits on the device for another GPU.
#include "rt_types.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- the plan: who renders what, who completes which frame (no GPU needed) ----
struct StripeSpan { // memory rows [firstRow, firstRow + rows) of a Screen-order frame
    int firstRow, rows;
};

// Image rows [8 s, 8 s + 8) of stripe s are memory rows [H - 8 s - 8, H - 8 s) of the Screen-order frame (row-flipped,
// src/screen.cpp:119-122); the last stripe of a frame whose height is no multiple of 8 is shorter.
StripeSpan stripe_span(const int height, const int stripe)
{
    const int lo = stripe * 8, hi = std::min(height, lo + 8);
    return StripeSpan { height - hi, hi - lo };
}

int stripe_count(const int height) { return (height + 7) / 8; }

// Meeting point of the participants' threads; `broken` ends every wait at once (a participant failed).
class Meeting {
public:
    explicit Meeting(int n) : m_n(n) {}
    bool arrive()
    {
        std::unique_lock<std::mutex> lock(m_mutex);
        if (m_broken) return false;
        const unsigned long long gen = m_generation;
        if (++m_waiting == m_n) {
            m_waiting = 0, m_generation++;
            m_cv.notify_all();
        } else {
            m_cv.wait(lock, [&] { return m_generation != gen || m_broken; });
        }
        return !m_broken;
    }
    void fail()
    {
        std::lock_guard<std::mutex> lock(m_mutex);
        m_broken = true;
        m_cv.notify_all();
    }

private:
    std::mutex m_mutex;
    std::condition_variable m_cv;
    int m_n, m_waiting = 0;
    unsigned long long m_generation = 0;
    bool m_broken = false;
};

struct Participant {
    rt_scene* scene = nullptr;
    int device = 0;
    hipStream_t renderStream = nullptr, pullStream = nullptr;
    float* d_frames[2] = { nullptr, nullptr }; // two sets of K frames
    float* d_scratch = nullptr;                // bloom's intermediate image
    float* h_stage = nullptr;                  // pinned: the frames this participant completes, on their way to the caller
    rt_stats stats {};
    int status = RT_OK;
    std::string message;
};

#define RT_MULTI_HIP(call)                                                                                 \
    do {                                                                                                   \
        const hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                            \
            me.status = RT_ERR_HIP;                                                                        \
            me.message = std::string(#call " failed: ") + hipGetErrorString(e_);                           \
            (void)hipGetLastError();                                                                       \
            return false;                                                                                  \
        }                                                                                                  \
    } while (0)

void add_stats(rt_stats& a, const rt_stats& b)
{
    a.primaryRays += b.primaryRays, a.shadowRays += b.shadowRays, a.secondaryRays += b.secondaryRays;
    a.nodePairSteps += b.nodePairSteps, a.triangleTests += b.triangleTests, a.queryNodeSteps += b.queryNodeSteps;
    a.pixels += b.pixels, a.stackOverflows += b.stackOverflows, a.referenceTreeQueries += b.referenceTreeQueries;
    a.pipeline = b.pipeline;
}

} // namespace

extern "C" {

// (exported for the tests: the plan is checked without a GPU)
int rt_stripe_owner(int height, int imageRow, int participants)
{
    if (height <= 0 || imageRow < 0 || imageRow >= height || participants <= 0) return -1;
    return (imageRow / 8) % participants;
}

int rt_stripe_plan(int height, int participants, int participant, int* firstRows, int* rowCounts, int capacity)
{
    if (height <= 0 || participants <= 0 || participant < 0 || participant >= participants) return -1;
    int n = 0;
    for (int s = participant; s < stripe_count(height); s += participants, n++) {
        const StripeSpan span = stripe_span(height, s);
        if (n < capacity && firstRows && rowCounts) firstRows[n] = span.firstRow, rowCounts[n] = span.rows;
    }
    return n;
}

int rt_render_images_multi(rt_scene* const* scenes, int nScenes, const rt_camera* cameras, int nFrames, const rt_features* features, int width,
                           int height, int startDepth, uint32_t flags, float* const* rgb, rt_stats* stats)
{
    if (!scenes || nScenes < 1 || !cameras || !features || !rgb || nFrames < 1 || width <= 0 || height <= 0) {
        rt::set_error("rt_render_images_multi: null argument, no scene, no frame or empty frames");
        return RT_ERR_INVALID;
    }
    for (int k = 0; k < nScenes; k++) {
        if (!scenes[k]) {
            rt::set_error("rt_render_images_multi: scene %d is null", k);
            return RT_ERR_INVALID;
        }
        for (int j = 0; j < k; j++)
            if (scenes[j] == scenes[k]) { // calls on one rt_scene must not overlap: every participant needs a scene of its own
                rt::set_error("rt_render_images_multi: scenes %d and %d are the same object (use rt_scene_clone)", j, k);
                return RT_ERR_INVALID;
            }
    }
    for (int f = 0; f < nFrames; f++)
        if (!rgb[f]) {
            rt::set_error("rt_render_images_multi: frame %d has no image", f);
            return RT_ERR_INVALID;
        }
    if (nScenes == 1) return rt_render_images(scenes[0], cameras, nFrames, features, width, height, startDepth, flags, rgb, stats);

    const int N = nScenes;
    const size_t px = (size_t)width * height, frameBytes = px * 12;
    // frames per launch sequence: a participant renders 1 / N of every frame, so sequences are as long as the library takes
    // (RT_MAX_FRAMES, and 65535 rows per sequence), within 16.6 M pixels of its own stripes
    int K = std::min<int>(RT_MAX_FRAMES, nFrames);
    K = std::min<int>(K, std::max(1, 65535 / height));
    K = (int)std::max<size_t>(1, std::min<size_t>((size_t)K, (size_t)8 * 1920 * 1080 * (size_t)N / px));
    const int nSequences = (nFrames + K - 1) / K;
    const int rootsPerSequence = (K + N - 1) / N;

    std::vector<Participant> parts((size_t)N);
    Meeting meeting(N);
    for (int k = 0; k < N; k++) {
        parts[(size_t)k].scene = scenes[k];
        if (rt_scene_device(scenes[k], &parts[(size_t)k].device) != RT_OK) return RT_ERR_INVALID;
    }

    auto work = [&](const int k) {
        Participant& me = parts[(size_t)k];
        const rt_region region { 8, k, N };
        auto setup = [&]() -> bool {
            RT_MULTI_HIP(hipSetDevice(me.device));
            for (int p = 0; p < N; p++) { // the pulls read the other participants' frames in place
                const int other = parts[(size_t)p].device;
                if (other == me.device) continue;
                int can = 0;
                RT_MULTI_HIP(hipDeviceCanAccessPeer(&can, me.device, other));
                if (can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) RT_MULTI_HIP(e);
                    (void)hipGetLastError();
                }
            }
            { // frames are queued on a stream the scene owns: the scene keeps frame buffers per stream for as long as it lives
                void* own = nullptr;
                if (rt_scene_stream(me.scene, 2, &own) != RT_OK) {
                    me.status = RT_ERR_HIP, me.message = rt_last_error();
                    return false;
                }
                me.renderStream = static_cast<hipStream_t>(own);
            }
            RT_MULTI_HIP(hipStreamCreateWithFlags(&me.pullStream, hipStreamNonBlocking));
            for (int b = 0; b < 2; b++) {
                RT_MULTI_HIP(hipMalloc(reinterpret_cast<void**>(&me.d_frames[b]), frameBytes * (size_t)K));
                RT_MULTI_HIP(hipMemsetAsync(me.d_frames[b], 0, frameBytes * (size_t)K, me.renderStream));
            }
            RT_MULTI_HIP(hipMalloc(reinterpret_cast<void**>(&me.d_scratch), rt_bloom_scratch_bytes(width, height)));
            RT_MULTI_HIP(hipHostMalloc(reinterpret_cast<void**>(&me.h_stage), frameBytes * (size_t)rootsPerSequence, hipHostMallocDefault));
            RT_MULTI_HIP(hipStreamSynchronize(me.renderStream));
            return true;
        };
        // this participant's stripes of sequence q, rendered and checked; queues that overflowed have grown: render again
        uint32_t myFlags = flags;
        auto render = [&](const int q) -> bool {
            const int first = q * K, n = std::min(K, nFrames - first);
            for (int attempt = 0;; attempt++) {
                int rc = rt_render_frames(me.scene, cameras + first, n, features, width, height, startDepth, &region, myFlags, me.d_frames[q & 1],
                                          nullptr, nullptr, me.renderStream);
                rt_stats st {};
                if (rc == RT_OK) rc = rt_sync(me.scene, me.renderStream, &st);
                if (rc == RT_OK) {
                    add_stats(me.stats, st);
                    return true;
                }
                if (rc == RT_ERR_AGAIN && attempt < 3) continue;
                if ((rc == RT_ERR_LIMIT || rc == RT_ERR_AGAIN) && !(myFlags & RT_RENDER_SERIAL) && attempt < 5) { // beyond the default pipeline's limits
                    myFlags |= RT_RENDER_SERIAL;
                    continue;
                }
                me.status = rc, me.message = rt_last_error();
                return false;
            }
        };
        // the frames of sequence q this participant completes: the others' stripes in, bloom, out to the caller
        auto complete = [&](const int q) -> bool {
            const int first = q * K, n = std::min(K, nFrames - first);
            int mine = 0;
            for (int f = k; f < n; f += N, mine++) {
                float* const frame = me.d_frames[q & 1] + 3 * px * (size_t)f;
                for (int p = 0; p < N; p++) {
                    if (p == k) continue;
                    const Participant& other = parts[(size_t)p];
                    const float* const theirs = other.d_frames[q & 1] + 3 * px * (size_t)f;
                    for (int s = p; s < stripe_count(height); s += N) {
                        const StripeSpan span = stripe_span(height, s);
                        const size_t off = (size_t)span.firstRow * width * 3, bytes = (size_t)span.rows * width * 12;
                        if (other.device == me.device) RT_MULTI_HIP(hipMemcpyAsync(frame + off, theirs + off, bytes, hipMemcpyDeviceToDevice, me.pullStream));
                        else RT_MULTI_HIP(hipMemcpyPeerAsync(frame + off, me.device, theirs + off, other.device, bytes, me.pullStream));
                    }
                }
                if (features->enableBloomEffect) {
                    const int rc = rt_bloom(frame, me.d_scratch, width, height, features->bloomFilterSize, features->bloomFilterThreshold,
                                            features->bloomFilterIntensity, nullptr, me.pullStream);
                    if (rc != RT_OK) {
                        me.status = rc, me.message = rt_last_error();
                        return false;
                    }
                }
                RT_MULTI_HIP(hipMemcpyAsync(me.h_stage + 3 * px * (size_t)mine, frame, frameBytes, hipMemcpyDeviceToHost, me.pullStream));
            }
            return true;
        };
        auto deliver = [&](const int q) -> bool { // (after the pull stream has been waited for)
            const int first = q * K, n = std::min(K, nFrames - first);
            RT_MULTI_HIP(hipStreamSynchronize(me.pullStream));
            int mine = 0;
            for (int f = k; f < n; f += N, mine++) memcpy(rgb[first + f], me.h_stage + 3 * px * (size_t)mine, frameBytes);
            return true;
        };
        bool ok = setup();
        if (!ok) meeting.fail();
        if (ok && !meeting.arrive()) ok = false; // every participant's buffers exist before anyone pulls from them
        // sequence q: render; deliver what was pulled for q - 1 meanwhile; meet; start pulling q
        for (int q = 0; ok && q < nSequences; q++) {
            ok = render(q);
            if (ok && q > 0) ok = deliver(q - 1);
            if (!ok) {
                meeting.fail();
                break;
            }
            if (!meeting.arrive()) {
                ok = false;
                break;
            }
            ok = complete(q);
            if (!ok) meeting.fail();
        }
        if (ok) ok = deliver(nSequences - 1);
        if (!ok) meeting.fail();
        // nobody frees a buffer another participant may still be pulling from
        (void)meeting.arrive();
        (void)hipSetDevice(me.device);
        if (me.pullStream) (void)hipStreamSynchronize(me.pullStream);
        if (me.renderStream) (void)hipStreamSynchronize(me.renderStream);
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < N; k++) threads.emplace_back(work, k);
    work(0);
    for (std::thread& t : threads) t.join();

    int rc = RT_OK;
    std::string message;
    rt_stats total {};
    for (Participant& p : parts) {
        (void)hipSetDevice(p.device);
        // (a failed participant broke the meeting: the others may have stopped short of their last wait)
        if (p.pullStream) (void)hipStreamSynchronize(p.pullStream), (void)hipStreamDestroy(p.pullStream);
        if (p.renderStream) (void)hipStreamSynchronize(p.renderStream); // (the scene's own stream: it stays)
        (void)hipFree(p.d_frames[0]), (void)hipFree(p.d_frames[1]), (void)hipFree(p.d_scratch);
        if (p.h_stage) (void)hipHostFree(p.h_stage);
        if (p.status != RT_OK && rc == RT_OK) rc = p.status, message = p.message;
        add_stats(total, p.stats);
    }
    (void)hipSetDevice(parts[0].device);
    if (rc != RT_OK) {
        rt::set_error("rt_render_images_multi: %s", message.c_str());
        return rc;
    }
    if (stats) *stats = total;
    return RT_OK;
}

} // extern "C"
