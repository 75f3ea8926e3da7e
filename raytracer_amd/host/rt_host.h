// Host-side mirror of the reference's render interface over the C ABI of include/rt_amd.h.
//
// Same names, members and argument meaning as the reference's seam, so that code written against
//   BVHInterface / BVH            (src/bvh_interface.h:9-71, src/bvh.h:14-31, 112-115)
//   renderImage                   (src/render.h:34)
//   Screen                        (src/screen.h:10-34)
//   Features / ExtraFeatures      (src/common.h:62-115)
//   Scene, Mesh, Vertex, Material (src/scene.h:27-35, framework/include/framework/mesh.h:14-43)
//   Ray, HitInfo, lights, Sphere  (framework/include/framework/ray.h:9-13, src/common.h:22-60)
//   Trackball                     (framework/include/framework/trackball.h, constructor + camera state)
// reads the same here.  The vector types are plain structs with glm's layout (glm::vec3 = 3 floats), so a
// build that has glm can alias them.  Everything lives in namespace rtamd.
//
// Differences a caller can observe: render calls throw std::runtime_error (message = rt_last_error())
// when the device path fails - the reference's cannot fail; Image holds decoded texels only (the reference's
// constructor decodes a file with stb_image, which is not part of this path); depth of field and motion blur are
// not part of this path.
#pragma once
#include "../../include/rt_amd.h"
#include <array>
#include <cstdint>
#include <filesystem>
#include <memory>
#include <span>
#include <variant>
#include <vector>

namespace rtamd {

struct vec2 { float x = 0, y = 0; };
struct vec3 { float x = 0, y = 0, z = 0; };
struct ivec2 { int x = 0, y = 0; };
struct uvec3 { uint32_t x = 0, y = 0, z = 0; };

struct Vertex {
    vec3 position, normal;
    vec2 texCoord;
};
struct Image { // framework/include/framework/image.h:14-24: rows as decoded (top row first)
    int width = 0, height = 0;
    std::vector<vec3> pixels;
};
struct Material {
    vec3 kd { 1, 1, 1 };
    vec3 ks { 0, 0, 0 };
    float shininess = 1.0f;
    float transparency = 1.0f;
    std::shared_ptr<Image> kdTexture; // sampled instead of kd when Features::enableTextureMapping is set
};
struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<uvec3> triangles;
    Material material;
};
struct Ray {
    vec3 origin;
    vec3 direction { 0, 0, -1 };
    float t = 3.402823466e+38f;
};
struct HitInfo {
    vec3 normal, barycentricCoord;
    vec2 texCoord;
    Material material;
};
struct AxisAlignedBox { vec3 lower, upper; };
struct Sphere {
    vec3 center;
    float radius = 1.0f;
    Material material;
};
struct PointLight { vec3 position, color; };
struct SegmentLight { vec3 endpoint0, endpoint1, color0, color1; };
struct ParallelogramLight { vec3 v0, edge01, edge02, color0, color1, color2, color3; };

enum class ShadingModel : int { Lambertian = 0, Phong = 1, BlinnPhong = 2, LinearGradient = 3 };
struct ExtraFeatures {
    bool enableBloomEffect = false;
    bool enableGlossyReflection = false;
    uint32_t bloomFilterSize = 15;
    float bloomFilterThreshold = 0.5f, bloomFilterIntensity = 0.8f;
    uint32_t numGlossySamples = 1;
    bool enableBvhSahBinning = false; // BVH::BVH splits spans by splitPrimitivesBySAHBin instead of at the median (src/common.h:63)
};
struct Features {
    bool enableShading = false, enableReflections = false, enableShadows = false, enableNormalInterp = false;
    bool enableTextureMapping = false, enableAccelStructure = true, enableBilinearTextureFiltering = false;
    bool enableTransparency = false, enableJitteredSampling = false;
    ShadingModel shadingModel = ShadingModel::Lambertian;
    uint32_t numPixelSamples = 1, numShadowSamples = 4;
    ExtraFeatures extra;
};

struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>> lights;
};

struct RenderState; // the reference threads {scene, features, bvh, sampler} through every call; here it only tags intersect()

struct BVHInterface {
    struct Primitive {
        uint32_t meshID;
        Vertex v0, v1, v2;
    };
    static_assert(sizeof(Primitive) == 100);
    struct Node {
        static constexpr uint32_t LeafBit = 1u << 31;
        AxisAlignedBox aabb;
        std::array<uint32_t, 2> data;
        bool isLeaf() const { return (data[0] & LeafBit) == LeafBit; }
        uint32_t primitiveOffset() const { return data[0] & (~LeafBit); }
        uint32_t primitiveCount() const { return data[1]; }
        uint32_t leftChild() const { return data[0]; }
        uint32_t rightChild() const { return data[1]; }
    };
    static_assert(sizeof(Node) == 32);

    virtual ~BVHInterface() {}
    virtual bool intersect(RenderState& state, Ray& ray, HitInfo& hitInfo) const = 0;
    virtual std::span<const Node> nodes() const = 0;
    virtual std::span<Node> nodes() = 0;
    virtual std::span<const Primitive> primitives() const = 0;
    virtual std::span<Primitive> primitives() = 0;
    virtual uint32_t numLevels() const = 0;
    virtual uint32_t numLeaves() const = 0;
};

struct RenderState {
    const Scene& scene;
    const Features& features;
    const BVHInterface& bvh;
};

// BVH::BVH(scene, features): the reference's build - median splits (rt_bvh_build) or, with
// features.extra.enableBvhSahBinning, SAH-binned ones (rt_bvh_build_sah_binning) - kept on the host in the
// reference layout for nodes() / primitives(), and flattened + uploaded once for the device path.
class BVH : public BVHInterface {
public:
    BVH(const Scene& scene, const Features& features);
    ~BVH() override;
    BVH(const BVH&) = delete;
    BVH& operator=(const BVH&) = delete;

    // closest hit of one ray on the GPU (rt_intersect with a batch of one); hitInfo.material = the hit mesh's
    bool intersect(RenderState& state, Ray& ray, HitInfo& hitInfo) const override;
    std::span<const Node> nodes() const override { return m_nodes; }
    std::span<Node> nodes() override { return m_nodes; }
    std::span<const Primitive> primitives() const override { return m_primitives; }
    std::span<Primitive> primitives() override { return m_primitives; }
    uint32_t numLevels() const override { return m_numLevels; }
    uint32_t numLeaves() const override { return m_numLeaves; }

    rt_scene* deviceScene() const { return m_device; }
    // The device scene on each of the first `gpus` visible GPUs (0 = all of them): the one uploaded at construction and
    // copies of it made GPU to GPU on first use (rt_scene_clone), kept until the BVH is destroyed.
    std::span<rt_scene* const> deviceScenes(int gpus = 0) const;

private:
    std::vector<Node> m_nodes;
    std::vector<Primitive> m_primitives;
    std::vector<Material> m_materials;
    uint32_t m_numLevels = 0, m_numLeaves = 0;
    rt_scene* m_device = nullptr;
    mutable std::vector<rt_scene*> m_perGpu; // [0] = m_device
};

// Camera state of the reference's Trackball: Trackball(window, fovy, lookAt, distance, rotationX, rotationY)
// without the window (the aspect ratio is given instead); angles in radians as in the reference's constructor.
class Trackball {
public:
    Trackball(float aspectRatio, float fovy, vec3 lookAt = {}, float distanceFromLookAt = 4.0f, float rotationX = 0.0f, float rotationY = 0.0f);
    void setCamera(vec3 lookAt, vec3 rotations, float distance);
    rt_camera deviceCamera() const;

private:
    float m_aspect, m_fovy, m_distance;
    vec3 m_lookAt, m_rotationEulerAngles;
};

class Screen {
public:
    explicit Screen(const ivec2& resolution, bool presentable = false);
    void clear(const vec3& color);
    void setPixel(int x, int y, const vec3& color);
    void writeBitmapToFile(const std::filesystem::path& filePath);
    ivec2 resolution() const { return m_resolution; }
    int indexAt(int x, int y) const { return (m_resolution.y - 1 - y) * m_resolution.x + x; }
    const std::vector<vec3>& pixels() const { return m_textureData; }
    std::vector<vec3>& pixels() { return m_textureData; }

private:
    ivec2 m_resolution;
    std::vector<vec3> m_textureData;
};

// renderImage (src/render.cpp:19-51): every pixel of `screen`, full recursion depth, bloom when enabled.
// A BVH of this library renders from its resident device scene; any other BVHInterface is flattened from
// its nodes() / primitives() for the call.
void renderImage(const Scene& scene, const BVHInterface& bvh, const Features& features, const Trackball& camera, Screen& screen);
// The same with the recursion started at rayDepth `startDepth` (B bounces = 6 - B) and the ray counters returned.
rt_stats renderImage(const Scene& scene, const BVHInterface& bvh, const Features& features, const Trackball& camera, Screen& screen,
                     int startDepth, uint32_t flags = 0);

// The loop that renders one frame per camera (src/main.cpp:515-526): screens[k] from cameras[k], all of one resolution.  The frames
// are rendered several at a time by one launch sequence each (rt_render_images); every screen holds what renderImage writes for its camera.
// With a BVH of this library and more than one visible GPU the frames are rendered on all of them (`gpus` = 0) or on the first
// `gpus`: every frame in interleaved stripes of 8 rows, one GPU per stripe in turn, assembled and bloomed on one GPU per frame
// (rt_render_images_multi).  gpus = 1 keeps to the GPU the BVH was uploaded to.
rt_stats renderImages(const Scene& scene, const BVHInterface& bvh, const Features& features, std::span<const Trackball> cameras,
                      std::span<Screen> screens, int startDepth = 0, uint32_t flags = 0, int gpus = 0);

rt_features toDevice(const Features& features);

} // namespace rtamd
