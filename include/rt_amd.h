/* rt_amd.h - C ABI of the MI355X (gfx950) render path.
 *
 * One shared library, librt_amd.so, exports exactly the symbols declared here.  The reference
 * renderer (zkkv/raytracer) has no FFI layer: its seam is the C++ interface BVHInterface /
 * renderImage / Screen.  Each entry point below names the reference interface it stands in for
 * (file:line relative to the reference tree); a binding on the reference side hands over the
 * spans and PODs those interfaces already hold.
 *
 * Conventions: every function returns an rt_status (0 = ok) unless stated; rt_last_error() gives
 * the message of the calling thread's last failure; nothing throws across this boundary; all
 * pointers are plain host pointers unless a parameter is documented as a device pointer.  Calls
 * on one rt_scene must be serialised by the caller; different scenes / devices are independent.
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = 1,     /* bad argument (null pointer, size mismatch, index out of range) */
    RT_ERR_HIP = 2,         /* a HIP runtime call failed; message holds hipGetErrorString */
    RT_ERR_UNSUPPORTED = 3, /* reserved: no feature flag of the path is rejected at present */
    RT_ERR_NOMEM = 4,
    RT_ERR_LIMIT = 5,       /* a size beyond a documented limit (tree depth, leaf size, filter size) */
    RT_ERR_AGAIN = 6        /* rt_sync: the frame overflowed the scene's task queues and is incomplete; the queues grow with the
                               next rt_render of the scene - render the frame again (rt_render_image does so by itself) */
} rt_status;

/* POD mirror of Features + ExtraFeatures (src/common.h:62-115).  int32 for every bool. */
typedef struct rt_features {
    int32_t enableShading, enableReflections, enableShadows, enableNormalInterp, enableTextureMapping;
    int32_t enableAccelStructure, enableBilinearTextureFiltering, enableTransparency, enableJitteredSampling;
    int32_t shadingModel;      /* ShadingModel: 0 Lambertian, 1 Phong, 2 BlinnPhong, 3 LinearGradient */
    uint32_t numPixelSamples;  /* 1 */
    uint32_t numShadowSamples; /* 4 */
    int32_t enableBloomEffect, enableGlossyReflection;
    uint32_t bloomFilterSize;  /* 15, at most 30 (src/extra.cpp:358) */
    float bloomFilterThreshold, bloomFilterIntensity; /* 0.5, 0.8 */
    uint32_t numGlossySamples; /* 1 */
    int32_t reserved[2];
} rt_features;

/* Camera state the reference recomputes per ray inside Trackball::generateRay
 * (framework/src/trackball.cpp:71-74, 101-110): position() and quat(eulerAngles), plus the
 * half extents of the image plane (trackball.cpp:26-27). */
typedef struct rt_camera {
    float position[3];
    float quat[4]; /* w, x, y, z */
    float halfWidth, halfHeight;
} rt_camera;

/* Which rows of the frame this call renders: rows are cut into stripes of `stripeRows` rows and
 * stripe s belongs to rank (s % world).  {0,0,1} or a null pointer = the whole frame. */
typedef struct rt_region {
    int32_t stripeRows;
    int32_t rank;
    int32_t world;
} rt_region;

/* Ray counters of the last render on a scene: one ray = one closest-hit query that contributes to
 * the image (the reference's unused glossy debug query, src/extra.cpp:510-511, is not traced). */
typedef struct rt_stats {
    uint64_t primaryRays, shadowRays, secondaryRays;
    uint64_t nodePairSteps, triangleTests; /* filled only by rt_render with RT_RENDER_COUNT_WORK */
    uint64_t pixels;
    uint64_t stackOverflows; /* must be 0 */
    uint64_t referenceTreeQueries; /* queries of the last wavefront frame that were traced again in the caller's tree
                                      because the query tree's result could not be proven equal (see RT_RENDER_REFERENCE_TREE) */
    uint64_t pipeline; /* which schedule rendered the frame: RT_PIPELINE_* */
    uint64_t queryNodeSteps; /* RT_RENDER_COUNT_SCHEDULE: 4-wide nodes (128 B) visited in the library's query tree */
} rt_stats;

enum {
    RT_PIPELINE_DEPTH_FIRST = 0, /* one kernel, a lane walks a pixel's whole recursion (RT_RENDER_SERIAL, or a recursion tree
                                    too long for the staged schedules) */
    RT_PIPELINE_ROUNDS = 1,      /* staged: the rays a pixel's sampler depends on in frame-wide rounds, the rest as queued tasks */
    RT_PIPELINE_PERSISTENT = 2   /* staged, with those rays walked pixel by pixel inside one persistent launch */
};

enum {
    RT_MAX_RAY_DEPTH = 6,      /* recursion continues while rayDepth < 6 (src/recursive.cpp:62) */
    RT_RENDER_COUNT_WORK = 1,  /* flag: also count node-pair steps and triangle tests (slower) */
    RT_RENDER_NO_CULL = 2,     /* flag: visit every box the ray enters, as the reference does
                                  (src/bvh.cpp:289-297), instead of skipping boxes beyond the closest hit */
    RT_RENDER_REFERENCE_TREE = 8, /* flag: walk only the tree passed to rt_scene_create.  By default closest-hit
                                  queries first walk a second tree that the library builds over the same triangles
                                  for traversal cost alone, and accept what they find only when it is provably what
                                  the reference's walk (src/bvh.cpp:235-370) reports; the rest is traced again in
                                  the reference's tree.  Results are identical either way. */
    RT_RENDER_COUNT_SCHEDULE = 32, /* flag: count the traversal work of the frame as it is scheduled by default - the query
                                  tree's 4-wide nodes (queryNodeSteps), node pairs of either tree walked by several lanes
                                  per ray or in the caller's tree (nodePairSteps), triangle tests, including candidate
                                  rays that are later discarded and rays traced a second time.  RT_RENDER_COUNT_WORK
                                  counts the reference's own walk instead and wins when both are set. */
    RT_RENDER_OVERLAPPED = 16, /* flag: the caller keeps several frames or launch sequences (rt_render_frames) queued on
                                  different streams (frames are independent: main.cpp renders one per camera,
                                  src/main.cpp:515-526).  A scene holds one set of working buffers per stream (up to eight),
                                  so they overlap on the GPU; with this flag the launches are sized for the throughput of
                                  the overlapped work rather than for the latency of a frame that has the GPU to itself.
                                  Frames are identical either way. */
    RT_RENDER_SERIAL = 4       /* flag: evaluate each pixel as one depth-first recursion and sum radiance in the
                                  reference's nested float order (src/recursive.cpp:25-83); the default pipeline
                                  defers shadow rays and last-level rays and sums per pixel in fixed point.  Taken by
                                  itself only for frames beyond the default pipeline's limits (enableTransparency with a
                                  recursion tree of more than 4095 sequential rays per pixel).  With enableTransparency
                                  either pipeline follows a light sample through transparent occluders
                                  (src/light.cpp:107-136) for at most 64 surfaces: the reference's walk never ends
                                  when the shaded surface is met more than 1e-5 off */
};

typedef struct rt_bvh rt_bvh;     /* host-side BVH in the reference's node / primitive layout */
typedef struct rt_scene rt_scene; /* device-resident flattened scene */

const char* rt_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Host side (no GPU needed)
 * ------------------------------------------------------------------------------------------- */

/* BVH::BVH(scene, features) with median split (src/bvh.cpp:48-95, 429-512).  prims = the scene's
 * triangles as BVHInterface::Primitive records (100 bytes: meshID, 3 x {position, normal, texCoord};
 * src/bvh_interface.h:14-23) in gather order (mesh order, then triangle order; src/bvh.cpp:63-74). */
int rt_bvh_build(const void* prims, uint32_t nPrims, rt_bvh** out);
/* The same constructor with features.extra.enableBvhSahBinning (src/bvh.cpp:472-495): spans are split by
 * splitPrimitivesBySAHBin (src/extra.cpp:582-699: sorted along the longest axis, up to 50 bins over the centroid
 * interval, the cheapest split after a bin) and a span whose cheapest split is no cheaper than testing all of
 * its triangles becomes a leaf of that many triangles.  RT_ERR_INVALID for input the reference's code is
 * undefined on: more than four triangles with one centroid coordinate on their split axis (the bin index is
 * 0/0 converted to size_t).  rt_scene_create accepts leaves of any size and trees of up to 32 levels. */
int rt_bvh_build_sah_binning(const void* prims, uint32_t nPrims, rt_bvh** out);
/* The same two constructors with the spans settled on the current GPU (sahBinning: 0 = rt_bvh_build's tree, non-zero =
 * rt_bvh_build_sah_binning's): per level one pass computes every span's box, sorts every span by its centroid coordinate on the
 * span's longest axis (src/bvh.cpp:181-214) and decides its split (median, or src/extra.cpp:623-699); the node numbering and the
 * doubled primitive array (src/bvh.cpp:387-392, 429-512) are then written on the host.  nodes() / primitives() come out bit for
 * bit as from the host builds.  Where the reference's result depends on the ORDER its code meets the primitives in - two equal
 * sort keys in a span (std::sort's order), or a box coordinate that is zero with both signs present - the span is handed to the
 * host builder together with everything below it.  info (may be null): [0] spans settled on the device, [1] spans handed to
 * the host, [2] levels launched, [3] primitives inside the spans handed to the host.  RT_ERR_HIP when no GPU can be used
 * (rt_bvh_build needs none); the errors of the host builds otherwise. */
int rt_bvh_build_device(const void* prims, uint32_t nPrims, int sahBinning, rt_bvh** out, uint32_t info[4]);
void rt_bvh_destroy(rt_bvh* bvh);
/* BVHInterface::nodes() / primitives() / numLevels() / numLeaves() (src/bvh_interface.h:64-71).
 * Nodes are 32 bytes (src/bvh_interface.h:28-54), index 1 is the reference's unused dummy; primitives
 * are stored twice per leaf exactly as the reference's build leaves them (src/bvh.cpp:387-392, 458). */
uint32_t rt_bvh_num_nodes(const rt_bvh* bvh);
uint32_t rt_bvh_num_primitives(const rt_bvh* bvh);
uint32_t rt_bvh_num_levels(const rt_bvh* bvh);
uint32_t rt_bvh_num_leaves(const rt_bvh* bvh);
const void* rt_bvh_nodes(const rt_bvh* bvh);
const void* rt_bvh_primitives(const rt_bvh* bvh);
/* Gather-order index of each stored primitive (canonical triangle id; no reference counterpart -
 * the reference identifies triangles only by position in the doubled array). */
const uint32_t* rt_bvh_primitive_ids(const rt_bvh* bvh);

/* Trackball(fovy, lookAt, dist, rotX, rotY) + position() + quat(eulerAngles)
 * (framework/src/trackball.cpp:23-31, 71-74; src/main.cpp:519-520).  Angles in degrees as in
 * CameraConfig (src/config.h:18-23); aspect = width / height (framework/src/window.cpp:413-418). */
int rt_camera_from_trackball(float fovDegrees, float distance, const float lookAt[3], const float rotationDegrees[3],
                             float aspect, rt_camera* out);

/* ---------------------------------------------------------------------------------------------
 * Device side
 * ------------------------------------------------------------------------------------------- */

/* Number of visible GPUs / select the device used by later calls of this thread. */
int rt_device_count(int* count);
int rt_set_device(int device);

/* Flatten + upload, once per scene.  Takes what `const BVHInterface&` and `const Scene&` give the
 * reference's renderImage (src/render.h:34): bvh.nodes() (nNodes x 32 B), bvh.primitives()
 * (nPrims x 100 B), per-mesh materials in mesh order as 8 floats {kd3, ks3, shininess, transparency}
 * (framework/include/framework/mesh.h:22-34; textures: rt_scene_set_texture), lights as 22 floats
 * {type: 0 point / 1 segment / 2 parallelogram, then the struct's floats, src/common.h:45-60},
 * spheres as 12 floats {center3, radius, kd3, ks3, shininess, transparency} (src/common.h:39-43).
 * primIds (optional, nPrims) = canonical id reported for each stored primitive; null = its index.
 * The tree is walked from node 0; unreachable nodes (the dummy) and the dead first copy of each
 * leaf's primitives are dropped.  Everything is copied; the caller may free its arrays. */
int rt_scene_create(const void* nodes, uint32_t nNodes, const void* prims, uint32_t nPrims, const uint32_t* primIds,
                    const float* materials8, uint32_t nMaterials, const float* lights22, uint32_t nLights,
                    const float* spheres12, uint32_t nSpheres, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
/* Releases what the scene has allocated for rendering - the working buffers of every stream frames were queued on (about
 * 3 KB per pixel for a scene's first frame, which is given room for the worst case the pipeline accepts - eight tasks per pixel;
 * later allocations hold half as much again as the fullest task queues and the largest round of the scene's completed frames,
 * about 1 KB per pixel for the headline frame, and a frame that needs more reports RT_ERR_AGAIN and makes them grow;
 * RT_AMD_FULL_QUEUES=1 keeps the worst case), the frame buffers of rt_render_image(s) - after waiting for the
 * device; the scene itself stays uploaded and the next render allocates again.  No counterpart in the reference (a
 * Scene there owns no render state); for callers that keep many scenes alive on one GPU.  Counters of frames not yet
 * fetched with rt_sync are lost. */
int rt_scene_trim(rt_scene* scene);

/* Material::kdTexture of mesh `material` (framework/include/framework/mesh.h:33): width*height RGB floats with the
 * rows as decoded, top row first - Image::pixels (framework/src/image.cpp:33-55), i.e. `&image.pixels[0].x`.
 * With enableTextureMapping a hit on that mesh takes its kd from the texture as sampleMaterialKd does
 * (src/shading.cpp:13-24: sampleTextureNearest / sampleTextureBilinear, src/texture.cpp:5-114, at the interpolated
 * texture coordinates of updateHitInfo, src/bvh.cpp:36-38).  Null rgb removes the texture.  The texels are copied;
 * the call waits for frames in flight on the scene's device. */
int rt_scene_set_texture(rt_scene* scene, uint32_t material, const float* rgb, int width, int height);

/* The per-pixel loop of renderImage (src/render.cpp:27-44) for the rows in `region`:
 * camera ray, renderRays(state, rays, startDepth) (src/recursive.cpp:10-83), Screen::setPixel.
 * d_rgb: DEVICE pointer, W*H*3 floats in Screen order, index (H-1-y)*W + x (src/screen.cpp:36-42).
 * d_primId / d_primT: optional DEVICE pointers (W*H, pixel order y*W + x) receiving the primary
 * ray's canonical triangle id (0xffffffff = miss, 0xfffffffe = sphere) and t.
 * startDepth is the rayDepth the reference passes to renderRays (0 = full depth 6; B bounces = 6-B).
 * stream: a hipStream_t (null = default stream).  Asynchronous; rt_stats are valid after rt_sync. */
int rt_render(rt_scene* scene, const rt_camera* camera, const rt_features* features, int width, int height,
              int startDepth, const rt_region* region, uint32_t flags, float* d_rgb, uint32_t* d_primId,
              float* d_primT, void* stream);

/* The loop that renders one frame per camera (src/main.cpp:515-526: for every camera of the configuration, renderImage
 * into a Screen and write it out) as ONE launch sequence: nFrames frames of the same scene, features, size and region,
 * frame f seen from cameras[f].  The frames are independent; rendering them together lets every stage of the pipeline
 * trace the rays of all frames in the same launches (a stage of one 1080p frame is too small to fill 256 CUs, and the
 * dependent stages of a frame are many).  d_rgb: DEVICE, nFrames * W*H*3 floats, frame f at offset f * W*H*3, each in
 * Screen order; d_primId / d_primT likewise (optional, nFrames * W*H).  Every frame is bit for bit what rt_render
 * writes for its camera.  rt_sync reports the counters of all nFrames frames together.  At most RT_MAX_FRAMES frames
 * and nFrames * height <= 65535; frames that need RT_RENDER_SERIAL or the pixel-per-lane schedule are rendered one
 * after the other on the stream. */
#define RT_MAX_FRAMES 16
int rt_render_frames(rt_scene* scene, const rt_camera* cameras, int nFrames, const rt_features* features, int width,
                     int height, int startDepth, const rt_region* region, uint32_t flags, float* d_rgb,
                     uint32_t* d_primId, float* d_primT, void* stream);

/* postprocessImageWithBloom (src/extra.cpp:330-416) in place on a DEVICE Screen-order image.
 * d_scratch: DEVICE buffer of at least rt_bloom_scratch_bytes(W,H) bytes.  With a region, only that
 * region's rows are written (all rows of the input must hold the pre-bloom frame). */
size_t rt_bloom_scratch_bytes(int width, int height);
int rt_bloom(float* d_rgb, float* d_scratch, int width, int height, uint32_t filterSize, float threshold,
             float intensity, const rt_region* region, void* stream);

/* rt_bloom over the whole frame for callers that cannot allocate device memory or pass a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * intermediate image lives in a static device array of the library (3 * RT_BLOOM_STATIC_MAX_PIXELS floats = 24 MiB on
 * every GPU the library runs on, held until the process ends); the call allocates and frees nothing.  Frames of more
 * than RT_BLOOM_STATIC_MAX_PIXELS pixels are rejected with RT_ERR_LIMIT.  Every call on a GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams, two threads) race; calls queued on one stream do not.  Results are
 * those of rt_bloom bit for bit.  The GPU that owns `stream` must be the calling thread's current device (RT_ERR_INVALID
 * otherwise).  The array's address on a GPU is looked up once, under a lock, by the first of rt_set_device, rt_scene_create
 * or rt_bloom_static that runs with that GPU current: a caller under graph capture makes one of those calls beforehand. */
#define RT_BLOOM_STATIC_MAX_PIXELS (1u << 21) /* 2,097,152: a 1920x1080 frame fits */
int rt_bloom_static(float* d_rgb, int width, int height, uint32_t filterSize, float threshold, float intensity,
                    void* stream);

/* Wait for the work queued on `stream` by the calls above and fetch the ray counters of the last frame queued on that
 * stream (a scene keeps the buffers and counters of a frame per stream: see RT_RENDER_OVERLAPPED; a stream must not be
 * destroyed while the scene lives).  A frame that is incomplete is reported in one of two ways.  RT_ERR_AGAIN: a task queue
 * of the default pipeline overflowed and can grow - the next rt_render of the scene allocates what the frame asked for:
 * render the frame again.  RT_ERR_LIMIT: a queue overflowed at its largest size, or a single contribution to a pixel
 * reached 4e6 in magnitude, beyond the fixed-point pixel sums of the default pipeline: render with RT_RENDER_SERIAL, which
 * has neither limit (rt_render_image and rt_render_images do both by themselves).  A launch sequence that was rendered frame
 * by frame is reported as a whole: incomplete if any of its frames is.  It also notes how many
 * rays each round of the frame held: the next rt_render of the same size, depth, flags, features and region leaves out
 * the launches that only serve larger rounds (a hint - the frame is the same whatever the plan says). */
int rt_sync(rt_scene* scene, void* stream, rt_stats* stats);
/* renderImage (src/render.cpp:19-51) for host callers: render + optional bloom + copy back.
 * rgb: HOST W*H*3 floats = Screen::pixels() (src/screen.h:25-26).  primId / primT: optional HOST
 * W*H arrays.  Multi-sample, depth-of-field and motion-blur drivers are not part of this path. */
int rt_render_image(rt_scene* scene, const rt_camera* camera, const rt_features* features, int width, int height,
                    int startDepth, uint32_t flags, float* rgb, uint32_t* primId, float* primT, rt_stats* stats);

/* The loop over cameras of src/main.cpp:515-526 for host callers: frame f of nFrames (any number) is rendered from
 * cameras[f] into rgb[f] (HOST, W*H*3 floats = that frame's Screen::pixels()), bloom included when enabled.  The frames
 * are rendered by rt_render_frames, up to eight at a time, with up to four such launch sequences in flight on streams the
 * scene owns (within 20 frames' worth of 1920 x 1080 pixels), their finished frames copied out through pinned memory while
 * the next ones render - the throughput of the camera sweep from one blocking call; every image is the one rt_render_image
 * writes for its camera.  stats: the counters of all frames together.  (Two of the four streams are created at the
 * device's higher stream priority: equal streams with equal work would run in step.) */
int rt_render_images(rt_scene* scene, const rt_camera* cameras, int nFrames, const rt_features* features, int width,
                     int height, int startDepth, uint32_t flags, float* const* rgb, rt_stats* stats);

/* The same loop on several GPUs of one node, from one process: scenes[k] is the same scene resident on some GPU (rt_scene_clone
 * puts a scene on another GPU; one scene object per participant, two of them may share a GPU).  Every frame is cut into
 * stripes of 8 image rows, stripe s is rendered by participant s % nScenes (pixels are independent in the reference's loop,
 * src/render.cpp:27-44, and a pixel's sampler seed depends on its own position only, src/render.cpp:38: nothing is exchanged
 * while rendering); frame f of a launch sequence is completed on participant f % nScenes, which copies the other participants'
 * stripes of it GPU to GPU, applies the bloom to the whole frame and hands it to the caller - every frame reaches host memory
 * once, from one GPU.  One host thread per participant; the threads meet once per launch sequence, and the stripe copies of
 * one sequence run while the next is rendered.  Images and counters are those of rt_render_images.  nScenes == 1 is
 * rt_render_images. */
int rt_render_images_multi(rt_scene* const* scenes, int nScenes, const rt_camera* cameras, int nFrames, const rt_features* features,
                           int width, int height, int startDepth, uint32_t flags, float* const* rgb, rt_stats* stats);
/* A copy of `scene` on GPU `device` (any visible GPU, also the one it is on): the uploaded arrays and textures are copied GPU
 * to GPU; working buffers, streams and launch plans are the copy's own.  No counterpart in the reference (a Scene there lives in
 * host memory that every thread shares); destroyed with rt_scene_destroy like any scene. */
int rt_scene_clone(const rt_scene* scene, int device, rt_scene** out);
/* The GPU a scene is resident on. */
int rt_scene_device(const rt_scene* scene, int* device);
/* One of the scene's own four streams (hipStream_t; created on first use, alive as long as the scene; the ones rt_render_images
 * queues its launch sequences on - streams 0 and 1 at the device's higher stream priority).  For callers that want frames in
 * flight without owning streams; do not use them while rt_render_images runs on the scene. */
int rt_scene_stream(rt_scene* scene, int index, void** stream);
/* The stripe partition, on the host (no GPU needed).  rt_stripe_owner: the participant that renders image row `imageRow`
 * (row 0 is the bottom row of the picture, as in Screen::setPixel, src/screen.cpp:36-42).  rt_stripe_plan: the stripes of
 * one participant as runs of memory rows of the Screen-order frame (memory row m holds image row height - 1 - m,
 * src/screen.cpp:119-122): up to `capacity` entries into firstRows / rowCounts; returns the number of stripes (-1: bad
 * arguments). */
int rt_stripe_owner(int height, int imageRow, int participants);
int rt_stripe_plan(int height, int participants, int participant, int* firstRows, int* rowCounts, int capacity);

/* BVHInterface::intersect (src/bvh_interface.h:61) for a batch of rays on the GPU.
 * rays: HOST n x 7 floats {origin3, direction3, t}; t is updated in place.
 * hits: HOST n x 12 dwords {normal3, barycentric3, texCoord2 (floats), primId, meshID, hit(0/1), 0}. */
int rt_intersect(rt_scene* scene, const rt_features* features, float* rays, uint32_t nRays, void* hits);

/* Diagnostic with no counterpart in the reference: checks on the GPU that the traversal's
 * reciprocal-based slab distance (one IEEE reciprocal per ray + fused corrections) is bit-identical to
 * the reference's `(bound - origin) / direction` over ~4M pseudo-random and edge-case operand pairs. */
int rt_selftest_division(uint32_t seed, uint64_t* mismatches, uint64_t* checked);

/* Diagnostic with no counterpart in the reference, host only (no GPU needed): flattens the tree as rt_scene_create
 * does, builds the library's query tree (see RT_RENDER_REFERENCE_TREE) and checks it - every triangle of the
 * reference's leaves in exactly one query-tree leaf, every child box inside the box stored for it and around its
 * triangles, both forms of the tree holding the same leaves, and each triangle's record naming its reference leaf's
 * box and route.  info (8 words): usable (0/1), triangles, binary pairs, binary depth, 4-wide nodes, 4-wide depth,
 * reference pairs, reference depth.  Returns RT_OK when the checks hold (usable = 0 when the reference's boxes are not
 * nested or too few triangles: queries then walk the caller's tree only). */
int rt_selftest_query_tree(const void* nodes, uint32_t nNodes, const void* prims, uint32_t nPrims, uint32_t nMaterials, uint32_t info[8]);
/* The same checks on a query tree whose hierarchy is built as rt_scene_create builds it: onDevice != 0 = on the current GPU
 * (binned surface-area splits, one tree level per launch; replaces the role of the reference's host-side build,
 * src/bvh.cpp:429-512, for the library's own tree - the reference-layout tree is still built on the host).  times (4 floats,
 * may be NULL): hierarchy build ms (wall, incl. transfers), device launches ms, layout writing ms (host), 1 if the device
 * build was used (0: it was not available and the host built the tree). */
int rt_selftest_query_tree_built(const void* nodes, uint32_t nNodes, const void* prims, uint32_t nPrims, uint32_t nMaterials, int onDevice,
                                 uint32_t info[8], float times[4]);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
