// TEST INFRASTRUCTURE - golden-vector generator. Not part of the product.
//
// This harness is compiled against the reference sources where they lie under
// /root/reference (never copied) and links ONLY reference translation units that
// build from their own files with no stand-ins: interpolate.cpp, shading.cpp,
// texture.cpp, light.cpp, recursive.cpp, extra.cpp, scene.cpp, framework mesh.cpp
// and image.cpp, tinyobjloader.  Functions that would need draw.cpp (GLU),
// screen.cpp (GLU), trackball.cpp (GLU), bvh.cpp (GLU) or the prebuilt intersect
// archive are never referenced from here; the link uses --gc-sections so the
// unreferenced reference functions that need them are discarded.
//
// Output: .npy arrays (inputs + expected outputs) written into the directory
// given as argv[1].  tests/golden/ holds the committed result.
#include "bvh_interface.h"
#include "common.h"
#include "interpolate.h"
#include "light.h"
#include "recursive.h"
#include "render.h"
#include "sampler.h"
#include "scene.h"
#include "shading.h"
#include "texture.h"
#include <framework/image.h>

#include <glm/gtc/quaternion.hpp>
#include <glm/geometric.hpp>

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <variant>
#include <vector>

// External-linkage helpers of src/extra.cpp that have no header declaration.
std::array<glm::vec3, 3> constructOrthonormalBasis(const glm::vec3& r);
std::array<float, 2> sampleDisk(RenderState& state, const float radius);
uint64_t binomial(const uint32_t n, const uint32_t k);
std::vector<glm::vec3> generateGaussianFilter(const uint32_t filterSize);
float perceivedLuminance(glm::vec3 colors);

static std::string g_out;

// ---- minimal .npy (v1.0) writer -------------------------------------------------
static void writeNpy(const std::string& name, const char* descr, const void* data,
                     size_t itemSize, const std::vector<size_t>& shape)
{
    std::string path = g_out + "/" + name + ".npy";
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(1); }
    std::string shp = "(";
    size_t count = 1;
    for (size_t i = 0; i < shape.size(); i++) {
        shp += std::to_string(shape[i]);
        shp += (shape.size() == 1 || i + 1 < shape.size()) ? "," : "";
        count *= shape[i];
    }
    shp += ")";
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + shp + ", }";
    size_t total = 10 + hdr.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    hdr.append(pad, ' ');
    hdr.push_back('\n');
    const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    fwrite(magic, 1, 8, f);
    uint16_t hl = (uint16_t)hdr.size();
    fwrite(&hl, 2, 1, f);
    fwrite(hdr.data(), 1, hdr.size(), f);
    fwrite(data, itemSize, count, f);
    fclose(f);
}
static void checkShape(const std::string& n, size_t have, const std::vector<size_t>& shape)
{
    size_t want = 1;
    for (size_t d : shape) want *= d;
    if (want != have) { fprintf(stderr, "%s: shape holds %zu elements, data has %zu\n", n.c_str(), want, have); exit(1); }
}
static void writeF32(const std::string& n, const std::vector<float>& v, std::vector<size_t> shape) { checkShape(n, v.size(), shape); writeNpy(n, "<f4", v.data(), 4, shape); }
static void writeU32(const std::string& n, const std::vector<uint32_t>& v, std::vector<size_t> shape) { checkShape(n, v.size(), shape); writeNpy(n, "<u4", v.data(), 4, shape); }

// ---- deterministic input generator (harness-owned, unrelated to the reference Sampler)
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) {}
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
    float uni(float a, float b) { return a + (b - a) * (float)(u32() >> 8) * (1.0f / 16777216.0f); }
    glm::vec3 v3(float a, float b) { float x = uni(a, b), y = uni(a, b), z = uni(a, b); return { x, y, z }; }
};
static void push3(std::vector<float>& v, const glm::vec3& a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }

// A BVHInterface whose intersect() answers from a script: the reference's light code
// (src/light.cpp) only needs `bool` + `ray.t` from it.
struct ScriptedBVH : public BVHInterface {
    bool hit = true;
    float t = 0.0f;
    mutable Ray lastRay;
    mutable int calls = 0;
    std::vector<Node> m_nodes;
    std::vector<Primitive> m_prims;
    bool intersect(RenderState&, Ray& ray, HitInfo&) const override
    {
        lastRay = ray;
        calls++;
        if (hit) ray.t = t;
        return hit;
    }
    std::span<const Node> nodes() const override { return m_nodes; }
    std::span<Node> nodes() override { return m_nodes; }
    std::span<const Primitive> primitives() const override { return m_prims; }
    std::span<Primitive> primitives() override { return m_prims; }
    uint32_t numLevels() const override { return 0; }
    uint32_t numLeaves() const override { return 0; }
};

// The same with a scripted SEQUENCE of answers, for code that queries the BVH several times per call (area lights,
// light through transparent occluders).  Answer k: {hit, relative, value, kd3, transparency}; the reported distance is
// `value`, or - with `relative` - the distance from the ray's origin to `target` plus `value`.  Past the script's end
// every query misses.  Every ray asked for is kept.
struct SequenceBVH : public BVHInterface {
    std::vector<float> script; // 7 floats per answer
    glm::vec3 target { 0 };
    mutable std::vector<Ray> rays;
    std::vector<Node> m_nodes;
    std::vector<Primitive> m_prims;
    bool intersect(RenderState&, Ray& ray, HitInfo& hitInfo) const override
    {
        const size_t k = rays.size();
        rays.push_back(ray);
        if (7 * k + 6 >= script.size()) return false;
        const float* a = &script[7 * k];
        if (a[0] == 0.0f) return false;
        ray.t = (a[1] != 0.0f) ? glm::length(target - ray.origin) + a[2] : a[2];
        hitInfo.material.kd = glm::vec3(a[3], a[4], a[5]);
        hitInfo.material.transparency = a[6];
        return true;
    }
    std::span<const Node> nodes() const override { return m_nodes; }
    std::span<Node> nodes() override { return m_nodes; }
    std::span<const Primitive> primitives() const override { return m_prims; }
    std::span<Primitive> primitives() override { return m_prims; }
    uint32_t numLevels() const override { return 0; }
    uint32_t numLeaves() const override { return 0; }
};

// ---- scene dumps: output of the reference's own loader (scene.cpp + mesh.cpp) ---
static void dumpScene(const char* name, SceneType type)
{
    Scene scene = loadScenePrebuilt(type, DATA_DIR);
    // Primitives in the gather order of BVH::BVH (src/bvh.cpp:63-74): mesh order, then triangle order.
    std::vector<uint32_t> prims; // 25 dwords each: meshID, 3 x (pos3, normal3, uv2)
    for (uint32_t meshID = 0; meshID < scene.meshes.size(); meshID++) {
        const auto& mesh = scene.meshes[meshID];
        for (const auto& tri : mesh.triangles) {
            BVHInterface::Primitive p { meshID, mesh.vertices[tri.x], mesh.vertices[tri.y], mesh.vertices[tri.z] };
            static_assert(sizeof(BVHInterface::Primitive) == 100);
            uint32_t w[25];
            memcpy(w, &p, 100);
            prims.insert(prims.end(), w, w + 25);
        }
    }
    std::vector<float> mats; // kd3 ks3 shininess transparency hasTexture(0/1)
    for (const auto& mesh : scene.meshes) {
        push3(mats, mesh.material.kd);
        push3(mats, mesh.material.ks);
        mats.push_back(mesh.material.shininess);
        mats.push_back(mesh.material.transparency);
        mats.push_back(mesh.material.kdTexture ? 1.0f : 0.0f);
    }
    std::vector<float> lights; // type(0 point,1 segment,2 parallelogram) + 21 floats
    for (const auto& l : scene.lights) {
        float rec[22] = { 0 };
        if (std::holds_alternative<PointLight>(l)) {
            const auto& p = std::get<PointLight>(l);
            rec[0] = 0;
            memcpy(rec + 1, &p, sizeof(p));
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& p = std::get<SegmentLight>(l);
            rec[0] = 1;
            memcpy(rec + 1, &p, sizeof(p));
        } else {
            const auto& p = std::get<ParallelogramLight>(l);
            rec[0] = 2;
            memcpy(rec + 1, &p, sizeof(p));
        }
        lights.insert(lights.end(), rec, rec + 22);
    }
    std::vector<float> spheres; // center3 radius kd3 ks3 shininess transparency
    for (const auto& s : scene.spheres) {
        push3(spheres, s.center);
        spheres.push_back(s.radius);
        push3(spheres, s.material.kd);
        push3(spheres, s.material.ks);
        spheres.push_back(s.material.shininess);
        spheres.push_back(s.material.transparency);
    }
    std::string base = std::string("scene_") + name;
    writeU32(base + "_prims", prims, { prims.size() / 25, 25 });
    writeF32(base + "_materials", mats, { mats.size() / 9, 9 });
    writeF32(base + "_lights", lights, { lights.size() / 22, 22 });
    writeF32(base + "_spheres", spheres, { spheres.size() / 12, 12 });
    printf("scene %-16s meshes=%zu tris=%zu lights=%zu spheres=%zu\n", name, scene.meshes.size(), prims.size() / 25, scene.lights.size(), scene.spheres.size());
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <outdir>\n", argv[0]); return 2; }
    g_out = argv[1];

    // ---------------- scenes (reference loader output) ----------------
    dumpScene("triangle", SingleTriangle);
    dumpScene("cube", Cube);
    dumpScene("cornell", CornellBox);
    dumpScene("cornell_plight", CornellBoxParallelogramLight);
    dumpScene("monkey", Monkey);
    dumpScene("teapot", Teapot);
    dumpScene("spheres", Spheres);

    // ---------------- Sampler (src/sampler.h) ----------------
    {
        const uint32_t seeds[] = { 0u, 1u, 4u, 0xffffffffu, 12345u, 512u * 100u + 37u };
        std::vector<uint32_t> sd;
        std::vector<float> draws;
        for (uint32_t s : seeds) {
            sd.push_back(s);
            Sampler smp(s);
            for (int i = 0; i < 32; i++) draws.push_back(smp.next_1d());
        }
        writeU32("sampler_seeds", sd, { sd.size() });
        writeF32("sampler_draws", draws, { sd.size(), 32 });
    }

    // ---------------- interpolate.cpp ----------------
    {
        Rng rng(1);
        const size_t N = 512;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            glm::vec3 v0 = rng.v3(-1, 1), v1 = rng.v3(-1, 1), v2 = rng.v3(-1, 1);
            float a = rng.uni(0, 1), b = rng.uni(0, 1 - a);
            glm::vec3 p = v0 + a * (v1 - v0) + b * (v2 - v0);
            glm::vec3 n0 = rng.v3(-1, 1), n1 = rng.v3(-1, 1), n2 = rng.v3(-1, 1);
            glm::vec2 t0 { rng.uni(0, 1), rng.uni(0, 1) }, t1 { rng.uni(0, 1), rng.uni(0, 1) }, t2 { rng.uni(0, 1), rng.uni(0, 1) };
            glm::vec3 bc = computeBarycentricCoord(v0, v1, v2, p);
            glm::vec3 nn = interpolateNormal(n0, n1, n2, bc);
            glm::vec2 tt = interpolateTexCoord(t0, t1, t2, bc);
            push3(in, v0); push3(in, v1); push3(in, v2); push3(in, p);
            push3(in, n0); push3(in, n1); push3(in, n2);
            in.insert(in.end(), { t0.x, t0.y, t1.x, t1.y, t2.x, t2.y });
            push3(out, bc); push3(out, nn); out.push_back(tt.x); out.push_back(tt.y);
        }
        writeF32("interp_in", in, { N, 27 });
        writeF32("interp_out", out, { N, 8 });
    }

    // Everything below needs a RenderState.
    Scene scene = loadScenePrebuilt(SingleTriangle, DATA_DIR);
    ScriptedBVH sbvh;

    // ---------------- shading.cpp ----------------
    {
        Rng rng(2);
        const size_t N = 1024;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            Features f {};
            uint32_t model = rng.u32() % 3; // Lambertian, Phong, BlinnPhong
            f.enableShading = (rng.u32() % 8) != 0;
            f.shadingModel = (ShadingModel)model;
            RenderState state { .scene = scene, .features = f, .bvh = sbvh, .sampler = { 7u } };
            HitInfo hi;
            hi.normal = rng.v3(-1, 1);
            hi.barycentricCoord = glm::vec3(0);
            hi.texCoord = glm::vec2(0);
            hi.material.kd = rng.v3(0, 1);
            hi.material.ks = rng.v3(0, 1);
            hi.material.shininess = (i % 3 == 0) ? 10.0f : rng.uni(1.0f, 260.0f);
            hi.material.transparency = 1.0f;
            glm::vec3 cam = rng.v3(-1, 1), l = rng.v3(-1, 1), lc = rng.v3(0, 1.5f);
            if (i % 5 == 0) l = glm::normalize(l);
            glm::vec3 r = computeShading(state, cam, l, lc, hi);
            in.push_back(f.enableShading ? 1.f : 0.f); in.push_back((float)model);
            push3(in, cam); push3(in, l); push3(in, lc); push3(in, hi.normal);
            push3(in, hi.material.kd); push3(in, hi.material.ks); in.push_back(hi.material.shininess);
            push3(out, r);
        }
        writeF32("shading_in", in, { N, 21 });
        writeF32("shading_out", out, { N, 3 });
    }

    // ---------------- light.cpp: samplers + point-light contribution with a scripted BVH ----------------
    {
        Rng rng(3);
        const size_t N = 256;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            SegmentLight sl { rng.v3(-2, 2), rng.v3(-2, 2), rng.v3(0, 1), rng.v3(0, 1) };
            ParallelogramLight pl { rng.v3(-1, 1), rng.v3(-1, 1), rng.v3(-1, 1), rng.v3(0, 1), rng.v3(0, 1), rng.v3(0, 1), rng.v3(0, 1) };
            float s = rng.uni(0, 1);
            glm::vec2 s2 { rng.uni(0, 1), rng.uni(0, 1) };
            glm::vec3 p, c, p2, c2;
            sampleSegmentLight(s, sl, p, c);
            sampleParallelogramLight(s2, pl, p2, c2);
            float rec[12 + 21 + 3];
            memcpy(rec, &sl, 48); memcpy(rec + 12, &pl, 84);
            rec[33] = s; rec[34] = s2.x; rec[35] = s2.y;
            in.insert(in.end(), rec, rec + 36);
            push3(out, p); push3(out, c); push3(out, p2); push3(out, c2);
        }
        writeF32("lightsample_in", in, { N, 36 });
        writeF32("lightsample_out", out, { N, 12 });
    }
    {
        // computeContributionPointLight (src/light.cpp:151-159) over a scripted intersect():
        // the scripted BVH reports a hit at t = |p - L| + delta, delta sweeping across the 5e-4 window.
        Rng rng(4);
        const size_t N = 512;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            Features f {};
            f.enableShading = true;
            f.enableShadows = (i % 7) != 0;
            f.shadingModel = ShadingModel::Phong;
            RenderState state { .scene = scene, .features = f, .bvh = sbvh, .sampler = { 7u } };
            PointLight L { rng.v3(-2, 2), rng.v3(0, 1) };
            Ray ray;
            ray.origin = rng.v3(-3, 3);
            ray.direction = glm::normalize(rng.v3(-1, 1));
            ray.t = rng.uni(0.5f, 4.0f);
            HitInfo hi;
            hi.normal = rng.v3(-1, 1);
            hi.barycentricCoord = glm::vec3(0);
            hi.texCoord = glm::vec2(0);
            hi.material.kd = rng.v3(0, 1);
            hi.material.ks = rng.v3(0, 1);
            hi.material.shininess = rng.uni(1.0f, 50.0f);
            glm::vec3 p = ray.origin + ray.t * ray.direction;
            float dist = glm::length(p - L.position);
            const float deltas[] = { 0.0f, 4e-4f, -4e-4f, 6e-4f, -6e-4f, 5e-4f, -0.3f, 0.2f };
            float delta = deltas[i % 8];
            sbvh.hit = (i % 11) != 0;
            sbvh.t = dist + delta;
            sbvh.calls = 0;
            glm::vec3 r = computeContributionPointLight(state, L, ray, hi);
            in.push_back(f.enableShadows ? 1.f : 0.f); in.push_back(sbvh.hit ? 1.f : 0.f); in.push_back(sbvh.t);
            push3(in, L.position); push3(in, L.color); push3(in, ray.origin); push3(in, ray.direction); in.push_back(ray.t);
            push3(in, hi.normal); push3(in, hi.material.kd); push3(in, hi.material.ks); in.push_back(hi.material.shininess);
            push3(out, r);
            // shadow ray actually issued by the reference (origin, direction)
            if (sbvh.calls) { push3(out, sbvh.lastRay.origin); push3(out, sbvh.lastRay.direction); }
            else { push3(out, glm::vec3(0)); push3(out, glm::vec3(0)); }
            out.push_back((float)sbvh.calls);
        }
        writeF32("pointlight_in", in, { N, 26 });
        writeF32("pointlight_out", out, { N, 10 });
    }

    // ---------------- light.cpp: computeLightContribution over point / segment / parallelogram lights ----------------
    // (src/light.cpp:151-232, 247-276) with scripted visibility: sampler draw order, the per-sample division, the sampled
    // POSITION handed to computeShading as the light direction, binary and transparent visibility of every sample.
    {
        Rng rng(12);
        const size_t N = 384, kMaxLights = 3, kScript = 24, kMaxRays = 24;
        std::vector<float> in, out;
        std::vector<uint32_t> seeds;
        for (size_t i = 0; i < N; i++) {
            Scene sc2 = scene;
            sc2.lights.clear();
            const size_t nLights = 1 + rng.u32() % kMaxLights;
            std::vector<float> lrec(22 * kMaxLights, 0.0f);
            for (size_t l = 0; l < kMaxLights; l++) lrec[22 * l] = -1.0f; // unused slot
            for (size_t l = 0; l < nLights; l++) {
                const uint32_t ty = (i < 32) ? (uint32_t)(i % 3) : rng.u32() % 3;
                float* r = &lrec[22 * l];
                r[0] = (float)ty;
                if (ty == 0) {
                    PointLight pl { rng.v3(-2, 2), rng.v3(0, 1) };
                    memcpy(r + 1, &pl, sizeof(pl));
                    sc2.lights.push_back(pl);
                } else if (ty == 1) {
                    SegmentLight sl { rng.v3(-2, 2), rng.v3(-2, 2), rng.v3(0, 1), rng.v3(0, 1) };
                    memcpy(r + 1, &sl, sizeof(sl));
                    sc2.lights.push_back(sl);
                } else {
                    ParallelogramLight pg { rng.v3(-1, 1), rng.v3(-1, 1), rng.v3(-1, 1), rng.v3(0, 1), rng.v3(0, 1), rng.v3(0, 1), rng.v3(0, 1) };
                    memcpy(r + 1, &pg, sizeof(pg));
                    sc2.lights.push_back(pg);
                }
            }
            Features f {};
            f.enableShading = (i % 9) != 8;
            f.enableShadows = (i % 5) != 4;
            f.enableTransparency = (i % 4) == 3;
            f.shadingModel = (i % 3) ? ShadingModel::Phong : ShadingModel::Lambertian;
            f.numShadowSamples = 1 + rng.u32() % 5;
            SequenceBVH qbvh;
            const uint32_t seed = rng.u32();
            RenderState state { .scene = sc2, .features = f, .bvh = qbvh, .sampler = { seed } };
            Ray ray;
            ray.origin = rng.v3(-3, 3);
            ray.direction = glm::normalize(rng.v3(-1, 1));
            ray.t = rng.uni(0.5f, 4.0f);
            HitInfo hi;
            hi.normal = rng.v3(-1, 1);
            hi.barycentricCoord = glm::vec3(0);
            hi.texCoord = glm::vec2(0);
            hi.material.kd = rng.v3(0, 1);
            hi.material.ks = rng.v3(0, 1);
            hi.material.shininess = rng.uni(1.0f, 50.0f);
            hi.material.transparency = (i % 2) ? 1.0f : rng.uni(0.1f, 0.9f);
            qbvh.target = ray.origin + ray.t * ray.direction;
            // answers: mostly "the shaded surface itself" (relative, within / just outside the windows of 5e-4 and 1e-5),
            // some occluders nearer the light, some misses
            const float deltas[] = { 0.0f, 4e-6f, -4e-6f, 4e-4f, -4e-4f, 6e-4f, 2e-5f, -0.3f };
            for (size_t k = 0; k < kScript; k++) {
                const uint32_t kind = rng.u32() % 8;
                float a[7] = { kind == 7 ? 0.0f : 1.0f, 1.0f, deltas[rng.u32() % 8], 0, 0, 0, 0 };
                if (kind >= 5 && kind < 7) a[1] = 0.0f, a[2] = rng.uni(0.05f, 0.6f); // an occluder close to the light
                glm::vec3 kd = rng.v3(0, 1);
                a[3] = kd.x, a[4] = kd.y, a[5] = kd.z, a[6] = (kind % 2) ? 1.0f : rng.uni(0.1f, 0.9f);
                // with transparency the walk of one sample only ends on a surface within 1e-5 of the shaded point or on a
                // miss: make every fourth answer such an end
                if (f.enableTransparency && k % 4 == 3) a[0] = (k % 8 == 3) ? 0.0f : 1.0f, a[1] = 1.0f, a[2] = 0.0f;
                qbvh.script.insert(qbvh.script.end(), a, a + 7);
            }
            glm::vec3 Lo = computeLightContribution(state, ray, hi);
            const float next = state.sampler.next_1d(); // pins the sampler's state after the call
            in.insert(in.end(), { f.enableShading ? 1.f : 0.f, f.enableShadows ? 1.f : 0.f, f.enableTransparency ? 1.f : 0.f,
                                  (float)(int)f.shadingModel, (float)f.numShadowSamples, (float)nLights });
            in.insert(in.end(), lrec.begin(), lrec.end());
            push3(in, ray.origin); push3(in, ray.direction); in.push_back(ray.t);
            push3(in, hi.normal); push3(in, hi.material.kd); push3(in, hi.material.ks); in.push_back(hi.material.shininess);
            in.push_back(hi.material.transparency);
            in.insert(in.end(), qbvh.script.begin(), qbvh.script.end());
            seeds.push_back(seed);
            push3(out, Lo);
            out.push_back(next);
            out.push_back((float)qbvh.rays.size());
            for (size_t k = 0; k < kMaxRays; k++) {
                if (k < qbvh.rays.size()) { push3(out, qbvh.rays[k].origin); push3(out, qbvh.rays[k].direction); }
                else { push3(out, glm::vec3(0)); push3(out, glm::vec3(0)); }
            }
        }
        writeF32("lightcontrib_in", in, { N, 6 + 22 * kMaxLights + 7 + 11 + 7 * kScript });
        writeU32("lightcontrib_seed", seeds, { N });
        writeF32("lightcontrib_out", out, { N, 5 + 6 * kMaxRays });
    }
    // ---------------- light.cpp: visibilityOfLightSampleTransparency (src/light.cpp:107-136) ----------------
    // 0 to 4 occluders between the light and the shaded point, then the shaded surface itself (within or outside the
    // 1e-5 window) or nothing.
    {
        Rng rng(13);
        const size_t N = 256, kScript = 8, kMaxRays = 8;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            Features f {};
            f.enableShadows = true, f.enableTransparency = true;
            SequenceBVH qbvh;
            RenderState state { .scene = scene, .features = f, .bvh = qbvh, .sampler = { 7u } };
            Ray ray;
            ray.origin = rng.v3(-3, 3);
            ray.direction = glm::normalize(rng.v3(-1, 1));
            ray.t = rng.uni(0.5f, 4.0f);
            HitInfo hi;
            hi.normal = rng.v3(-1, 1);
            hi.barycentricCoord = glm::vec3(0);
            hi.texCoord = glm::vec2(0);
            glm::vec3 lpos = rng.v3(-2, 2), lcol = rng.v3(0, 1.5f);
            qbvh.target = ray.origin + ray.t * ray.direction;
            const size_t occluders = i % 5;
            const float dist = glm::length(qbvh.target - lpos);
            for (size_t k = 0; k < kScript; k++) {
                glm::vec3 kd = rng.v3(0, 1);
                float a[7] = { 1.0f, 0.0f, 0.0f, kd.x, kd.y, kd.z, rng.uni(0.0f, 1.0f) };
                if (k < occluders) {
                    a[2] = dist * rng.uni(0.05f, 0.3f); // measured from the ray's own origin, which moves on with every surface
                } else if (k == occluders) {
                    const uint32_t end = (uint32_t)(i / 5) % 4;
                    a[1] = 1.0f;
                    if (end == 0) a[2] = 0.0f;            // the shaded surface, exactly
                    else if (end == 1) a[2] = 8e-6f;      // ... within the window
                    else if (end == 2) a[0] = 0.0f;       // nothing: the walk ends on a miss
                    else a[2] = 3e-5f;                    // just outside the window: the walk goes on (and then misses)
                } else {
                    a[0] = 0.0f;
                }
                qbvh.script.insert(qbvh.script.end(), a, a + 7);
            }
            glm::vec3 r = visibilityOfLightSampleTransparency(state, lpos, lcol, ray, hi);
            push3(in, lpos); push3(in, lcol); push3(in, ray.origin); push3(in, ray.direction); in.push_back(ray.t);
            in.insert(in.end(), qbvh.script.begin(), qbvh.script.end());
            push3(out, r);
            out.push_back((float)qbvh.rays.size());
            for (size_t k = 0; k < kMaxRays; k++) {
                if (k < qbvh.rays.size()) { push3(out, qbvh.rays[k].origin); push3(out, qbvh.rays[k].direction); }
                else { push3(out, glm::vec3(0)); push3(out, glm::vec3(0)); }
            }
        }
        writeF32("transvis_in", in, { N, 13 + 7 * kScript });
        writeF32("transvis_out", out, { N, 4 + 6 * kMaxRays });
    }

    // ---------------- recursive.cpp: generateReflectionRay / generatePassthroughRay ----------------
    {
        Rng rng(5);
        const size_t N = 512;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            Ray ray;
            ray.origin = rng.v3(-3, 3);
            ray.direction = (i % 2) ? glm::normalize(rng.v3(-1, 1)) : rng.v3(-1, 1);
            ray.t = rng.uni(0.1f, 5.0f);
            HitInfo hi;
            hi.normal = (i % 3) ? glm::normalize(rng.v3(-1, 1)) : rng.v3(-1, 1);
            hi.barycentricCoord = glm::vec3(0);
            hi.texCoord = glm::vec2(0);
            Ray r = generateReflectionRay(ray, hi);
            Ray q = generatePassthroughRay(ray, hi);
            push3(in, ray.origin); push3(in, ray.direction); in.push_back(ray.t); push3(in, hi.normal);
            push3(out, r.origin); push3(out, r.direction); out.push_back(r.t);
            push3(out, q.origin); push3(out, q.direction); out.push_back(q.t);
        }
        writeF32("reflect_in", in, { N, 10 });
        writeF32("reflect_out", out, { N, 14 });
    }

    // ---------------- extra.cpp: glossy helpers + bloom filter ----------------
    {
        Rng rng(6);
        const size_t N = 512;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            glm::vec3 r = rng.v3(-1, 1);
            if (i % 9 == 0) r.x = r.y; // exercise the tie branch of the smallest-component test
            if (i % 13 == 0) r = glm::vec3(0, 0, rng.uni(0.1f, 1));
            auto b = constructOrthonormalBasis(r);
            push3(in, r);
            push3(out, b[0]); push3(out, b[1]); push3(out, b[2]);
        }
        writeF32("basis_in", in, { N, 3 });
        writeF32("basis_out", out, { N, 9 });
    }
    {
        Features f {};
        const uint32_t seeds[] = { 0u, 3u, 99u, 0xdeadbeefu };
        const float radii[] = { 1.0f, 0.05f, 0.002f, 0.125f };
        std::vector<float> out;
        for (int k = 0; k < 4; k++) {
            RenderState state { .scene = scene, .features = f, .bvh = sbvh, .sampler = { seeds[k] } };
            for (int i = 0; i < 16; i++) {
                auto d = sampleDisk(state, radii[k]);
                out.push_back(d[0]); out.push_back(d[1]);
            }
        }
        writeF32("disk_out", out, { 4, 16, 2 });
        writeU32("disk_seeds", std::vector<uint32_t>(seeds, seeds + 4), { 4 });
        writeF32("disk_radii", std::vector<float>(radii, radii + 4), { 4 });
    }
    {
        for (uint32_t F : { 1u, 2u, 7u, 15u, 30u }) {
            auto flt = generateGaussianFilter(F);
            std::vector<float> v;
            for (auto& c : flt) v.push_back(c.x);
            writeF32("bloom_filter_F" + std::to_string(F), v, { v.size() });
        }
        std::vector<uint32_t> bn;
        for (uint32_t n : { 2u, 30u, 60u, 62u })
            for (uint32_t k = 0; k <= n; k++) {
                uint64_t c = binomial(n, k);
                bn.push_back(n); bn.push_back(k); bn.push_back((uint32_t)c); bn.push_back((uint32_t)(c >> 32));
            }
        writeU32("binomial", bn, { bn.size() / 4, 4 });
        Rng rng(8);
        std::vector<float> in, out;
        for (int i = 0; i < 256; i++) {
            glm::vec3 c = rng.v3(0, 1.2f);
            push3(in, c);
            out.push_back(perceivedLuminance(c));
        }
        writeF32("luminance_in", in, { 256, 3 });
        writeF32("luminance_out", out, { 256 });
    }

    // ---------------- texture.cpp + sampleMaterialKd (src/shading.cpp:13-24) ----------------
    // The image is the reference's own decode of data/default.png (framework/src/image.cpp); a second, odd-sized image
    // with harness-made texels exercises the clamping and the row flip of getPixelFromCoordinates.
    {
        Image img(std::string(DATA_DIR) + "default.png");
        std::vector<float> px;
        for (const auto& c : img.pixels) push3(px, c);
        writeF32("texture_default", px, { (size_t)img.height, (size_t)img.width, 3 });
        Image odd = img;
        odd.width = 5, odd.height = 3;
        odd.pixels.clear();
        Rng trng(10);
        for (int i = 0; i < odd.width * odd.height; i++) odd.pixels.push_back(trng.v3(0, 1));
        std::vector<float> opx;
        for (const auto& c : odd.pixels) push3(opx, c);
        writeF32("texture_odd", opx, { (size_t)odd.height, (size_t)odd.width, 3 });
        Rng rng(11);
        const size_t N = 2048;
        std::vector<float> in, outN, outB, outNo, outBo;
        for (size_t i = 0; i < N; i++) {
            glm::vec2 uv { rng.uni(0, 1), rng.uni(0, 1) };
            if (i % 16 == 1) uv = { rng.uni(-0.5f, 1.5f), rng.uni(-0.5f, 1.5f) }; // outside [0,1]: clamped texel coordinates
            if (i % 16 == 2) uv = { (float)(rng.u32() % 129) / 128.0f, (float)(rng.u32() % 129) / 128.0f }; // texel edges, incl. 1.0
            if (i % 16 == 3) uv = { ((float)(rng.u32() % 128) + 0.5f) / 128.0f, ((float)(rng.u32() % 128) + 0.5f) / 128.0f }; // texel centres
            if (i % 16 == 4) uv = { (float)(rng.u32() % 6) / 5.0f, (float)(rng.u32() % 4) / 3.0f };
            in.push_back(uv.x); in.push_back(uv.y);
            push3(outN, sampleTextureNearest(img, uv)); push3(outB, sampleTextureBilinear(img, uv));
            push3(outNo, sampleTextureNearest(odd, uv)); push3(outBo, sampleTextureBilinear(odd, uv));
        }
        writeF32("texture_uv", in, { N, 2 });
        writeF32("texture_default_nearest", outN, { N, 3 });
        writeF32("texture_default_bilinear", outB, { N, 3 });
        writeF32("texture_odd_nearest", outNo, { N, 3 });
        writeF32("texture_odd_bilinear", outBo, { N, 3 });
        // sampleMaterialKd through computeShading: Lambert and unshaded, texture mapping on / off, both filters
        std::vector<float> sin_, sout;
        auto tex = std::make_shared<Image>(odd);
        for (size_t i = 0; i < 512; i++) {
            Features f {};
            f.enableShading = (i % 4) != 0;
            f.shadingModel = (i % 2) ? ShadingModel::Lambertian : ShadingModel::Phong;
            f.enableTextureMapping = (i % 8) != 7;
            f.enableBilinearTextureFiltering = (i % 3) == 0;
            RenderState state { .scene = scene, .features = f, .bvh = sbvh, .sampler = { 7u } };
            HitInfo hi;
            hi.normal = rng.v3(-1, 1);
            hi.barycentricCoord = glm::vec3(0);
            hi.texCoord = { rng.uni(0, 1), rng.uni(0, 1) };
            hi.material.kd = rng.v3(0, 1);
            hi.material.ks = rng.v3(0, 1);
            hi.material.shininess = rng.uni(1.0f, 60.0f);
            hi.material.transparency = 1.0f;
            const bool hasTex = (i % 5) != 4;
            if (hasTex) hi.material.kdTexture = tex;
            glm::vec3 cam = rng.v3(-1, 1), l = rng.v3(-1, 1), lc = rng.v3(0, 1.5f);
            glm::vec3 r = computeShading(state, cam, l, lc, hi);
            sin_.insert(sin_.end(), { f.enableShading ? 1.f : 0.f, (float)(int)f.shadingModel, f.enableTextureMapping ? 1.f : 0.f,
                                      f.enableBilinearTextureFiltering ? 1.f : 0.f, hasTex ? 1.f : 0.f, hi.texCoord.x, hi.texCoord.y });
            push3(sin_, cam); push3(sin_, l); push3(sin_, lc); push3(sin_, hi.normal);
            push3(sin_, hi.material.kd); push3(sin_, hi.material.ks); sin_.push_back(hi.material.shininess);
            push3(sout, r);
        }
        writeF32("texshade_in", sin_, { 512, 26 });
        writeF32("texshade_out", sout, { 512, 3 });
    }

    // ---------------- glm semantics used by framework/src/trackball.cpp ----------------
    // trackball.cpp itself cannot be built here (its includes need GLU headers), so this pins
    // only the vendored glm 0.9.9 expressions it uses: quat(euler), quat*vec3, normalize, radians.
    {
        Rng rng(9);
        const size_t N = 256;
        std::vector<float> in, out;
        for (size_t i = 0; i < N; i++) {
            glm::vec3 eulerDeg = (i == 0) ? glm::vec3(20.f, 20.f, 0.f) : glm::vec3(rng.uni(-90, 90), rng.uni(-360, 360), 0.f);
            glm::vec3 euler = glm::radians(eulerDeg);
            float dist = (i == 0) ? 3.0f : rng.uni(0.5f, 6.0f);
            glm::vec3 lookAt = (i == 0) ? glm::vec3(0) : rng.v3(-0.5f, 0.5f);
            float halfH = std::tan(glm::radians(50.0f) / 2.0f);
            float halfW = (1920.0f / 1080.0f) * halfH;
            glm::vec2 pixel { rng.uni(-1, 1), rng.uni(-1, 1) };
            glm::quat q = glm::quat(euler);
            glm::vec3 pos = lookAt + q * glm::vec3(0, 0, -dist);
            glm::vec3 camDir = glm::normalize(glm::vec3(-pixel.x * halfW, pixel.y * halfH, 1.0f));
            glm::vec3 dir = q * camDir;
            push3(in, eulerDeg); in.push_back(dist); push3(in, lookAt); in.push_back(pixel.x); in.push_back(pixel.y);
            in.push_back(halfW); in.push_back(halfH);
            out.insert(out.end(), { q.w, q.x, q.y, q.z });
            push3(out, pos); push3(out, dir);
        }
        writeF32("camera_in", in, { N, 11 });
        writeF32("camera_out", out, { N, 10 });
    }
    printf("golden vectors written to %s\n", g_out.c_str());
    return 0;
}
