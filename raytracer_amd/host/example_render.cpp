// Renders a small procedural scene through the reference-shaped host interface and writes the frame as raw
// float RGB (Screen order) plus a BMP.  usage: example_render <out.rgbf32> [width height startDepth glossy]
#include "rt_host.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace rtamd;

static Mesh quad(vec3 a, vec3 b, vec3 c, vec3 d, vec3 n, Material m)
{
    Mesh out;
    out.vertices = { { a, n, {} }, { b, n, {} }, { c, n, {} }, { d, n, {} } };
    out.triangles = { { 0, 1, 2 }, { 0, 2, 3 } };
    out.material = m;
    return out;
}

Scene exampleScene()
{
    Scene s;
    Material floorM, wallM, mirrorM;
    floorM.kd = { 0.7f, 0.7f, 0.65f }, floorM.ks = { 0.1f, 0.1f, 0.1f }, floorM.shininess = 20.0f;
    wallM.kd = { 0.8f, 0.2f, 0.2f };
    mirrorM.kd = { 0.1f, 0.1f, 0.3f }, mirrorM.ks = { 0.8f, 0.8f, 0.8f }, mirrorM.shininess = 60.0f;
    s.meshes.push_back(quad({ -1, -0.5f, -1 }, { -1, -0.5f, 1 }, { 1, -0.5f, 1 }, { 1, -0.5f, -1 }, { 0, 1, 0 }, floorM));
    s.meshes.push_back(quad({ -1, -0.5f, -1 }, { 1, -0.5f, -1 }, { 1, 1, -1 }, { -1, 1, -1 }, { 0, 0, 1 }, wallM));
    // a pyramid of 4 mirror faces
    Mesh p;
    p.material = mirrorM;
    const vec3 apex { 0.1f, 0.45f, 0.05f };
    const vec3 base[4] = { { -0.3f, -0.5f, -0.3f }, { 0.5f, -0.5f, -0.3f }, { 0.5f, -0.5f, 0.5f }, { -0.3f, -0.5f, 0.5f } };
    for (int k = 0; k < 4; k++) {
        const vec3 a = base[(k + 1) % 4], b = base[k];
        const vec3 e1 { b.x - a.x, b.y - a.y, b.z - a.z }, e2 { apex.x - a.x, apex.y - a.y, apex.z - a.z };
        vec3 n { e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x };
        const float len = std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z);
        n = { n.x / len, n.y / len, n.z / len };
        const uint32_t i0 = (uint32_t)p.vertices.size();
        p.vertices.push_back({ a, n, {} }), p.vertices.push_back({ b, n, {} }), p.vertices.push_back({ apex, n, {} });
        p.triangles.push_back({ i0, i0 + 1, i0 + 2 });
    }
    s.meshes.push_back(p);
    s.lights.emplace_back(PointLight { { 0.4f, 1.2f, 1.5f }, { 1, 1, 1 } });
    return s;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <out.rgbf32> [width height startDepth glossySamples sahBinning]\n", argv[0]);
        return 2;
    }
    const int W = argc > 2 ? std::atoi(argv[2]) : 256, H = argc > 3 ? std::atoi(argv[3]) : 192;
    const int startDepth = argc > 4 ? std::atoi(argv[4]) : 2;
    const int glossy = argc > 5 ? std::atoi(argv[5]) : 0;
    try {
        const Scene scene = exampleScene();
        Features features;
        features.enableShading = features.enableShadows = features.enableReflections = features.enableNormalInterp = true;
        features.shadingModel = ShadingModel::Phong;
        features.extra.enableGlossyReflection = glossy > 0;
        features.extra.numGlossySamples = glossy > 0 ? (uint32_t)glossy : 1;
        features.extra.enableBloomEffect = true;
        features.extra.enableBvhSahBinning = argc > 6 && std::atoi(argv[6]) != 0;
        const BVH bvh(scene, features);
        const float kRad = 0.01745329251994329576923690768489f;
        const Trackball camera((float)W / (float)H, 50.0f * kRad, { 0, 0, 0 }, 3.0f, 20.0f * kRad, 20.0f * kRad);
        Screen screen({ W, H });
        const rt_stats st = renderImage(scene, bvh, features, camera, screen, startDepth);
        // single-ray entry point: a ray straight down onto the floor
        RenderState state { scene, features, bvh };
        Ray ray { { 0.8f, 1.0f, 0.8f }, { 0, -1, 0 } };
        HitInfo hit;
        const bool ok = bvh.intersect(state, ray, hit);
        std::printf("nodes %zu primitives %zu levels %u leaves %u\n", bvh.nodes().size(), bvh.primitives().size(), bvh.numLevels(), bvh.numLeaves());
        std::printf("rays primary %llu shadow %llu secondary %llu\n", (unsigned long long)st.primaryRays, (unsigned long long)st.shadowRays,
                    (unsigned long long)st.secondaryRays);
        std::printf("probe hit %d t %.9g normal %g %g %g kd %g\n", ok ? 1 : 0, ray.t, hit.normal.x, hit.normal.y, hit.normal.z, hit.material.kd.x);
        FILE* f = std::fopen(argv[1], "wb");
        if (!f) return 1;
        std::fwrite(screen.pixels().data(), sizeof(vec3), screen.pixels().size(), f);
        std::fclose(f);
        screen.writeBitmapToFile(std::string(argv[1]) + ".bmp");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
