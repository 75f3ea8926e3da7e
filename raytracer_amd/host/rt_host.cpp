#include "rt_host.h"

#include <cstdlib>
#include "../csrc/rt_math.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace rtamd {
namespace {

void check(int rc, const char* what)
{
    if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}

std::vector<float> packMaterials(const std::vector<Material>& mats)
{
    std::vector<float> out;
    for (const Material& m : mats) out.insert(out.end(), { m.kd.x, m.kd.y, m.kd.z, m.ks.x, m.ks.y, m.ks.z, m.shininess, m.transparency });
    return out;
}
std::vector<float> packLights(const Scene& scene)
{
    std::vector<float> out;
    for (const auto& l : scene.lights) {
        float rec[22] = { 0 };
        if (const auto* p = std::get_if<PointLight>(&l)) {
            rec[0] = 0;
            std::memcpy(rec + 1, p, sizeof(*p));
        } else if (const auto* s = std::get_if<SegmentLight>(&l)) {
            rec[0] = 1;
            std::memcpy(rec + 1, s, sizeof(*s));
        } else {
            rec[0] = 2;
            std::memcpy(rec + 1, &std::get<ParallelogramLight>(l), sizeof(ParallelogramLight));
        }
        out.insert(out.end(), rec, rec + 22);
    }
    return out;
}
std::vector<float> packSpheres(const Scene& scene)
{
    std::vector<float> out;
    for (const Sphere& s : scene.spheres)
        out.insert(out.end(), { s.center.x, s.center.y, s.center.z, s.radius, s.material.kd.x, s.material.kd.y, s.material.kd.z, s.material.ks.x,
                                s.material.ks.y, s.material.ks.z, s.material.shininess, s.material.transparency });
    return out;
}
rt_scene* createDeviceScene(std::span<const BVHInterface::Node> nodes, std::span<const BVHInterface::Primitive> prims, const uint32_t* ids,
                            const Scene& scene)
{
    std::vector<Material> mats;
    for (const Mesh& m : scene.meshes) mats.push_back(m.material);
    const std::vector<float> m8 = packMaterials(mats), l22 = packLights(scene), s12 = packSpheres(scene);
    rt_scene* out = nullptr;
    check(rt_scene_create(nodes.data(), (uint32_t)nodes.size(), prims.data(), (uint32_t)prims.size(), ids, m8.data(), (uint32_t)mats.size(),
                          l22.data(), (uint32_t)scene.lights.size(), s12.data(), (uint32_t)scene.spheres.size(), &out),
          "rt_scene_create");
    for (size_t m = 0; m < mats.size(); m++) {
        const Image* img = mats[m].kdTexture.get();
        if (img && img->width > 0 && img->height > 0 && img->pixels.size() == size_t(img->width) * size_t(img->height))
            check(rt_scene_set_texture(out, (uint32_t)m, &img->pixels[0].x, img->width, img->height), "rt_scene_set_texture");
    }
    return out;
}

} // namespace

rt_features toDevice(const Features& f)
{
    rt_features r {};
    r.enableShading = f.enableShading, r.enableReflections = f.enableReflections, r.enableShadows = f.enableShadows;
    r.enableNormalInterp = f.enableNormalInterp, r.enableTextureMapping = f.enableTextureMapping;
    r.enableAccelStructure = f.enableAccelStructure, r.enableBilinearTextureFiltering = f.enableBilinearTextureFiltering;
    r.enableTransparency = f.enableTransparency, r.enableJitteredSampling = f.enableJitteredSampling;
    r.shadingModel = (int32_t)f.shadingModel;
    r.numPixelSamples = f.numPixelSamples, r.numShadowSamples = f.numShadowSamples;
    r.enableBloomEffect = f.extra.enableBloomEffect, r.enableGlossyReflection = f.extra.enableGlossyReflection;
    r.bloomFilterSize = f.extra.bloomFilterSize;
    r.bloomFilterThreshold = f.extra.bloomFilterThreshold, r.bloomFilterIntensity = f.extra.bloomFilterIntensity;
    r.numGlossySamples = f.extra.numGlossySamples;
    return r;
}

// Gathers the scene's triangles in mesh order, then triangle order (src/bvh.cpp:63-74)
BVH::BVH(const Scene& scene, const Features& features)
{
    std::vector<Primitive> gathered;
    for (uint32_t m = 0; m < scene.meshes.size(); m++) {
        const Mesh& mesh = scene.meshes[m];
        m_materials.push_back(mesh.material);
        for (const uvec3& t : mesh.triangles) gathered.push_back(Primitive { m, mesh.vertices[t.x], mesh.vertices[t.y], mesh.vertices[t.z] });
    }
    rt_bvh* h = nullptr;
    // The spans are settled on the GPU the scene is about to live on (the same nodes() / primitives() bit for bit);
    // RT_AMD_HOST_BUILD=1 keeps the whole build on the host.
    const char* hostBuild = getenv("RT_AMD_HOST_BUILD");
    if (!hostBuild || !*hostBuild || *hostBuild == '0')
        check(rt_bvh_build_device(gathered.data(), (uint32_t)gathered.size(), features.extra.enableBvhSahBinning ? 1 : 0, &h, nullptr), "rt_bvh_build_device");
    else if (features.extra.enableBvhSahBinning) // src/bvh.cpp:472-499
        check(rt_bvh_build_sah_binning(gathered.data(), (uint32_t)gathered.size(), &h), "rt_bvh_build_sah_binning");
    else
        check(rt_bvh_build(gathered.data(), (uint32_t)gathered.size(), &h), "rt_bvh_build");
    const auto* n = static_cast<const Node*>(rt_bvh_nodes(h));
    const auto* p = static_cast<const Primitive*>(rt_bvh_primitives(h));
    m_nodes.assign(n, n + rt_bvh_num_nodes(h));
    m_primitives.assign(p, p + rt_bvh_num_primitives(h));
    m_numLevels = rt_bvh_num_levels(h), m_numLeaves = rt_bvh_num_leaves(h);
    try {
        m_device = createDeviceScene(m_nodes, m_primitives, rt_bvh_primitive_ids(h), scene);
    } catch (...) {
        rt_bvh_destroy(h);
        throw;
    }
    rt_bvh_destroy(h);
}
BVH::~BVH()
{
    for (size_t k = 1; k < m_perGpu.size(); k++) rt_scene_destroy(m_perGpu[k]);
    rt_scene_destroy(m_device);
}

std::span<rt_scene* const> BVH::deviceScenes(int gpus) const
{
    int visible = 1, home = 0;
    check(rt_device_count(&visible), "rt_device_count");
    check(rt_scene_device(m_device, &home), "rt_scene_device");
    const int want = std::max(1, gpus <= 0 ? visible : std::min(gpus, visible));
    if (m_perGpu.empty()) m_perGpu.push_back(m_device);
    // participant k lives on GPU (home + k) % visible: the scene's own GPU first
    while ((int)m_perGpu.size() < want) {
        rt_scene* copy = nullptr;
        check(rt_scene_clone(m_device, (home + (int)m_perGpu.size()) % visible, &copy), "rt_scene_clone");
        m_perGpu.push_back(copy);
    }
    return std::span<rt_scene* const>(m_perGpu.data(), (size_t)want);
}

bool BVH::intersect(RenderState& state, Ray& ray, HitInfo& hitInfo) const
{
    const rt_features f = toDevice(state.features);
    float r[7] = { ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z, ray.t };
    uint32_t h[12] = { 0 };
    check(rt_intersect(m_device, &f, r, 1, h), "rt_intersect");
    if (!h[10]) return false;
    ray.t = r[6];
    float v[8];
    std::memcpy(v, h, sizeof(v));
    hitInfo.normal = { v[0], v[1], v[2] };
    hitInfo.barycentricCoord = { v[3], v[4], v[5] };
    hitInfo.texCoord = { v[6], v[7] };
    const uint32_t mesh = h[9];
    if (h[8] == 0xfffffffeu) hitInfo.material = state.scene.spheres.at(mesh).material; // a sphere: meshID holds its index
    else hitInfo.material = m_materials.at(mesh);
    return true;
}

Trackball::Trackball(float aspectRatio, float fovy, vec3 lookAt, float distance, float rotationX, float rotationY)
    : m_aspect(aspectRatio), m_fovy(fovy), m_distance(distance), m_lookAt(lookAt), m_rotationEulerAngles { rotationX, rotationY, 0.0f }
{
}
void Trackball::setCamera(vec3 lookAt, vec3 rotations, float distance)
{
    m_lookAt = lookAt, m_rotationEulerAngles = rotations, m_distance = distance;
}
// position() and quat(eulerAngles) of the reference's Trackball (framework/src/trackball.cpp:71-74, 101-110) and the
// half extents of its image plane (trackball.cpp:26-27): the same float expressions as rt_camera_from_trackball,
// started from radians.  This file is built with -ffp-contract=off like the rest of the path.
rt_camera Trackball::deviceCamera() const
{
    rt_camera c;
    const float ex = m_rotationEulerAngles.x, ey = m_rotationEulerAngles.y, ez = m_rotationEulerAngles.z;
    const float cx = std::cos(ex * 0.5f), cy = std::cos(ey * 0.5f), cz = std::cos(ez * 0.5f);
    const float sx = std::sin(ex * 0.5f), sy = std::sin(ey * 0.5f), sz = std::sin(ez * 0.5f);
    c.quat[0] = cx * cy * cz + sx * sy * sz;
    c.quat[1] = sx * cy * cz - cx * sy * sz;
    c.quat[2] = cx * sy * cz + sx * cy * sz;
    c.quat[3] = cx * cy * sz - sx * sy * cz;
    const rt::vec3 pos = rt::make3(m_lookAt.x, m_lookAt.y, m_lookAt.z) + rt::quat_rotate(c.quat, rt::make3(0.0f, 0.0f, -m_distance));
    c.position[0] = pos.x, c.position[1] = pos.y, c.position[2] = pos.z;
    c.halfHeight = std::tan(m_fovy / 2.0f);
    c.halfWidth = m_aspect * c.halfHeight;
    return c;
}

Screen::Screen(const ivec2& resolution, bool) : m_resolution(resolution), m_textureData((size_t)resolution.x * resolution.y) {}
void Screen::clear(const vec3& color) { std::fill(m_textureData.begin(), m_textureData.end(), color); }
void Screen::setPixel(int x, int y, const vec3& color) { m_textureData[indexAt(x, y)] = color; }

// 24-bit BMP of the clamped frame: value * 255 truncated, as the reference's conversion (src/screen.cpp:44-55)
void Screen::writeBitmapToFile(const std::filesystem::path& filePath)
{
    const int W = m_resolution.x, H = m_resolution.y, rowBytes = (3 * W + 3) & ~3;
    std::vector<uint8_t> img((size_t)rowBytes * H, 0);
    for (int row = 0; row < H; row++) // memory row 0 is the top image row; BMP stores bottom-up
        for (int x = 0; x < W; x++) {
            const vec3& c = m_textureData[(size_t)row * W + x];
            uint8_t* o = &img[(size_t)(H - 1 - row) * rowBytes + 3 * x];
            o[2] = (uint8_t)(std::clamp(c.x, 0.0f, 1.0f) * 255.0f);
            o[1] = (uint8_t)(std::clamp(c.y, 0.0f, 1.0f) * 255.0f);
            o[0] = (uint8_t)(std::clamp(c.z, 0.0f, 1.0f) * 255.0f);
        }
    uint8_t hdr[54] = { 'B', 'M' };
    auto put = [&](int off, uint32_t v) { std::memcpy(hdr + off, &v, 4); };
    put(2, 54 + (uint32_t)img.size()), put(10, 54), put(14, 40), put(18, (uint32_t)W), put(22, (uint32_t)H);
    hdr[26] = 1, hdr[28] = 24;
    put(34, (uint32_t)img.size());
    FILE* f = std::fopen(filePath.string().c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + filePath.string());
    std::fwrite(hdr, 1, 54, f);
    std::fwrite(img.data(), 1, img.size(), f);
    std::fclose(f);
}

rt_stats renderImage(const Scene& scene, const BVHInterface& bvh, const Features& features, const Trackball& camera, Screen& screen,
                     int startDepth, uint32_t flags)
{
    const auto* own = dynamic_cast<const BVH*>(&bvh);
    rt_scene* dev = own ? own->deviceScene() : createDeviceScene(bvh.nodes(), bvh.primitives(), nullptr, scene);
    const rt_features f = toDevice(features);
    const rt_camera cam = camera.deviceCamera();
    rt_stats stats {};
    const ivec2 res = screen.resolution();
    const int rc = rt_render_image(dev, &cam, &f, res.x, res.y, startDepth, flags, &screen.pixels()[0].x, nullptr, nullptr, &stats);
    if (!own) rt_scene_destroy(dev);
    check(rc, "rt_render_image");
    return stats;
}
rt_stats renderImages(const Scene& scene, const BVHInterface& bvh, const Features& features, std::span<const Trackball> cameras,
                      std::span<Screen> screens, int startDepth, uint32_t flags, int gpus)
{
    if (cameras.size() != screens.size() || cameras.empty()) throw std::invalid_argument("renderImages: one screen per camera, at least one");
    const ivec2 res = screens[0].resolution();
    std::vector<rt_camera> cams;
    std::vector<float*> images;
    for (size_t k = 0; k < cameras.size(); k++) {
        if (screens[k].resolution().x != res.x || screens[k].resolution().y != res.y) throw std::invalid_argument("renderImages: screens of one resolution");
        cams.push_back(cameras[k].deviceCamera());
        images.push_back(&screens[k].pixels()[0].x);
    }
    const auto* own = dynamic_cast<const BVH*>(&bvh);
    rt_scene* dev = own ? own->deviceScene() : createDeviceScene(bvh.nodes(), bvh.primitives(), nullptr, scene);
    const rt_features f = toDevice(features);
    rt_stats stats {};
    int rc = RT_OK;
    if (own && gpus != 1) { // every visible GPU (or the first `gpus`) takes its stripes of every frame
        const std::span<rt_scene* const> team = own->deviceScenes(gpus);
        rc = rt_render_images_multi(team.data(), (int)team.size(), cams.data(), (int)cams.size(), &f, res.x, res.y, startDepth, flags, images.data(), &stats);
    } else {
        rc = rt_render_images(dev, cams.data(), (int)cams.size(), &f, res.x, res.y, startDepth, flags, images.data(), &stats);
    }
    if (!own) rt_scene_destroy(dev);
    check(rc, "rt_render_images");
    return stats;
}
void renderImage(const Scene& scene, const BVHInterface& bvh, const Features& features, const Trackball& camera, Screen& screen)
{
    (void)renderImage(scene, bvh, features, camera, screen, 0, 0);
}

} // namespace rtamd
