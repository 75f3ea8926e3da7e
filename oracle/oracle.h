/* TEST INFRASTRUCTURE - CPU oracle for the render hot path.
 *
 * A plain C++ restatement of the reference renderer's algorithm (zkkv/raytracer), used only
 * as the checker in tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline leg.
 * Nothing in the product (raytracer_amd/, include/) may include, link or call this.
 *
 * Pinning status (what the restatement has been checked against):
 *   - sampler, barycentric/interpolation, shading models, light sampling, point-light
 *     visibility logic, reflection rays, glossy basis / disk sampling, texture sampling, bloom filter taps and
 *     the scene inputs are pinned bit-exactly against the reference's own code, compiled from
 *     its sources (oracle/ref/ref_harness.cpp -> tests/golden/ref_vectors.npz, scenes.npz).
 *   - PARITY UNPINNED: ray/triangle, ray/AABB and ray/sphere arithmetic (the reference ships it
 *     only as a prebuilt archive, which is never loaded here; restated from its disassembly
 *     listing), BVH build + traversal + hit record (src/bvh.cpp), the renderRay recursion
 *     driver, the glossy loop, the bloom driver, the camera and the Screen index (those
 *     reference files do not build in this image without GLU headers).  They are restated
 *     line by line with file:line citations in oracle.cpp.  What does pin them end to end is the
 *     number of rays the unmodified reference binary traces per frame (five configurations on
 *     Cornell, monkey and teapot), which the restatement reproduces exactly
 *     (tests/test_host_cpu.py), and the tree shapes the reference reports.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* POD mirror of Features/ExtraFeatures (reference src/common.h:62-115). */
typedef struct orc_features {
    int32_t enableShading, enableReflections, enableShadows, enableNormalInterp, enableTextureMapping;
    int32_t enableAccelStructure, enableBilinearTextureFiltering, enableTransparency, enableJitteredSampling;
    int32_t shadingModel;      /* 0 Lambertian, 1 Phong, 2 BlinnPhong, 3 LinearGradient */
    uint32_t numPixelSamples;  /* default 1 */
    uint32_t numShadowSamples; /* default 4 */
    int32_t enableBloomEffect, enableGlossyReflection;
    uint32_t bloomFilterSize;  /* default 15 */
    float bloomFilterThreshold, bloomFilterIntensity; /* 0.5, 0.8 */
    uint32_t numGlossySamples; /* default 1 */
    int32_t glossyDebugRay;    /* oracle only: also trace the reference's unused debug ray (src/extra.cpp:510-511) */
    int32_t reserved;
} orc_features;

/* Camera state of Trackball (framework/src/trackball.cpp:71-74,101-110), trig done by the caller. */
typedef struct orc_camera {
    float position[3];
    float quat[4]; /* w, x, y, z of glm::quat(eulerAngles) */
    float halfWidth, halfHeight;
} orc_camera;

enum { ORC_STAT_PRIMARY = 0, ORC_STAT_SHADOW, ORC_STAT_SECONDARY, ORC_STAT_DEBUG,
       ORC_STAT_NODE_PAIRS, ORC_STAT_TRI_TESTS, ORC_STAT_LEAVES, ORC_STAT_COUNT };

typedef struct orc_scene orc_scene;

/* prims25: nPrims x 25 dwords = BVHInterface::Primitive (meshID, 3 x {pos3, normal3, uv2}) in gather order.
 * materials9: kd3 ks3 shininess transparency hasTexture.  lights22: type + 21 floats.  spheres12. */
orc_scene* orc_scene_create(const uint32_t* prims25, uint32_t nPrims, const float* materials9, uint32_t nMaterials,
                            const float* lights22, uint32_t nLights, const float* spheres12, uint32_t nSpheres);
void orc_scene_destroy(orc_scene* s);
/* Builds the scene's BVH again: median splits (sahBinning = 0, what orc_scene_create does; src/bvh.cpp:499) or
 * features.extra.enableBvhSahBinning (src/bvh.cpp:472-495, src/extra.cpp:582-699).  Returns 1 when the SAH build
 * met a span whose centroids coincide on the split axis: the reference's bin index is 0/0 converted to size_t
 * there, which is undefined. */
int orc_scene_rebuild_bvh(orc_scene* s, int sahBinning);
/* Test aid, not reference behaviour: stop the walk through transparent occluders (src/light.cpp:107-136) after this many
   surfaces (0 = never, as the reference). */
void orc_set_shadow_walk_cap(int surfaces);
/* Material::kdTexture of mesh `material`: width*height RGB floats, rows as decoded (top row first) - the layout of
 * Image::pixels (framework/src/image.cpp:33-55).  Null rgb removes it.  Used when enableTextureMapping is set. */
int orc_scene_set_texture(orc_scene* s, uint32_t material, const float* rgb, int width, int height);

/* BVH in the reference layout: nodes 8 dwords each (incl. the dummy at index 1), primitives 25 dwords
 * each (stored twice per leaf, as the reference does); origIndex = gather index of each stored primitive. */
uint32_t orc_bvh_num_nodes(const orc_scene* s);
uint32_t orc_bvh_num_prims(const orc_scene* s);
uint32_t orc_bvh_num_levels(const orc_scene* s);
uint32_t orc_bvh_num_leaves(const orc_scene* s);
void orc_bvh_copy(const orc_scene* s, uint32_t* nodes8, uint32_t* prims25, uint32_t* origIndex);

/* Closest hit of one ray.  ray7 = origin3, direction3, t (in/out).  hit12 out = normal3, bary3, uv2,
 * then as dwords: gather index of the triangle (0xffffffff = none, 0xfffffffe = sphere), meshID/sphere index, 2 spare.
 * Returns 1 on hit. */
int orc_intersect(const orc_scene* s, const orc_features* f, float* ray7, float* hit12);

/* Whole frame (reference renderImage minus bloom), rows y0, y0+yStep, ... < y1.  rgb is W*H*3 in Screen order
 * (row-flipped, src/screen.cpp:36-42).  primId/primT (optional, W*H, pixel order y*W+x) receive the
 * primary ray's gather index and t.  stats (optional) += ORC_STAT_* counters. */
int orc_render(const orc_scene* s, const orc_camera* cam, const orc_features* f, int W, int H, int startDepth,
               int y0, int y1, int yStep, float* rgb, uint32_t* primId, float* primT, uint64_t* stats, int nThreads);

/* postprocessImageWithBloom (src/extra.cpp:330-416) on a Screen-order W*H*3 image, in place. */
int orc_bloom(float* rgb, int W, int H, uint32_t filterSize, float threshold, float intensity);

/* ---- unit entry points (pinned against tests/golden/ref_vectors.npz) ---- */
void orc_sampler_draws(uint32_t seed, int n, float* out);
void orc_barycentric(const float* v0, const float* v1, const float* v2, const float* p, float* out3);
void orc_interp_normal(const float* n0, const float* n1, const float* n2, const float* bc, float* out3);
void orc_interp_uv(const float* t0, const float* t1, const float* t2, const float* bc, float* out2);
void orc_shading(int enableShading, int model, const float* cam3, const float* light3, const float* lightColor3,
                 const float* normal3, const float* kd3, const float* ks3, float shininess, float* out3);
/* sampleTextureNearest / sampleTextureBilinear (src/texture.cpp:20-114) and computeShading with sampleMaterialKd. */
void orc_sample_texture(const float* rgb, int width, int height, int bilinear, const float* uv2, float* out3);
void orc_shading_textured(int enableShading, int model, int enableTextureMapping, int bilinear, const float* rgb, int width, int height,
                          const float* uv2, const float* cam3, const float* light3, const float* lightColor3, const float* normal3,
                          const float* kd3, const float* ks3, float shininess, float* out3);
void orc_sample_segment(const float* light12, float sample, float* pos3, float* col3);
void orc_sample_parallelogram(const float* light21, const float* sample2, float* pos3, float* col3);
/* computeContributionPointLight with the shadow query answered by (hit, tHit); writes the shadow ray issued. */
void orc_point_light(int enableShadows, int hit, float tHit, const float* lightPos3, const float* lightCol3,
                     const float* ray7, const float* normal3, const float* kd3, const float* ks3, float shininess,
                     float* out3, float* shadowRay6, int* nQueries);
/* computeLightContribution (src/light.cpp:262-276) / visibilityOfLightSampleTransparency (:107-136) with every visibility
 * query answered from a script (7 floats per answer: hit, relative, value, kd3, transparency). */
void orc_light_contribution(int enableShading, int enableShadows, int enableTransparency, int model, uint32_t numShadowSamples,
                            const float* lights22, int nLights, uint32_t seed, const float* ray7, const float* normal3, const float* kd3,
                            const float* ks3, float shininess, float transparency, const float* script7, int nScript, float* out3,
                            float* nextDraw, float* rays6, int maxRays, int* nQueries);
void orc_visibility_transparency(const float* lightPos3, const float* lightCol3, const float* ray7, const float* script7, int nScript,
                                 float* out3, float* rays6, int maxRays, int* nQueries);
void orc_reflect(const float* ray7, const float* normal3, float* outRay7);
void orc_passthrough(const float* ray7, float* outRay7);
void orc_basis(const float* r3, float* out9);
void orc_sample_disk(uint32_t seed, float radius, int n, float* out2n);
int orc_gaussian_filter(uint32_t filterSize, float* outTaps);
uint64_t orc_binomial(uint32_t n, uint32_t k);
float orc_luminance(const float* rgb3);
/* camera: quaternion/position from Euler angles (radians), and one pixel ray from NDC. */
void orc_camera_make(const float* eulerRad3, const float* lookAt3, float dist, float fovyRad, float aspect, orc_camera* out);
void orc_camera_ray(const orc_camera* cam, float ndcX, float ndcY, float* ray7);
void orc_aabb_test(const float* box6, float* ray7, int* hit);
int orc_triangle_test(const float* v0, const float* v1, const float* v2, float* ray7);

#ifdef __cplusplus
}
#endif
#endif
