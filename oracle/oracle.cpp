// TEST INFRASTRUCTURE - CPU oracle (see oracle.h for scope and pinning status).
//
// Restates, operation by operation, the reference renderer's hot path.  All arithmetic is IEEE
// fp32 with no contraction (build with -ffp-contract=off): every expression below keeps the
// operand order of the reference expression it cites, because float ties decide hit triangles.
// Citations are file:line relative to the reference tree (zkkv/raytracer @ 2024_10_08).
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <list>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------
// glm 0.9.9 vec3 semantics (framework/third_party/glm/glm/detail/func_geometric.inl:48-89)
// ------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
struct V2 {
    float x, y;
};
inline V3 v3(float a) { return { a, a, a }; }
inline V3 v3(const float* p) { return { p[0], p[1], p[2] }; }
inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 operator-(V3 a) { return { -a.x, -a.y, -a.z }; }
inline V3 operator*(V3 a, V3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline V3 operator*(float s, V3 v) { return { s * v.x, s * v.y, s * v.z }; }
inline V3 operator*(V3 v, float s) { return { v.x * s, v.y * s, v.z * s }; }
inline V3 operator/(V3 v, float s) { return { v.x / s, v.y / s, v.z / s }; }
inline V3 operator/(V3 a, V3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V2 operator+(V2 a, V2 b) { return { a.x + b.x, a.y + b.y }; }
inline V2 operator*(float s, V2 v) { return { s * v.x, s * v.y }; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; } // tmp.x + tmp.y + tmp.z
inline V3 cross(V3 x, V3 y) { return { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); } // v * inversesqrt(dot(v,v))
inline float gmin(float x, float y) { return (y < x) ? y : x; }          // glm::min
inline float gmax(float x, float y) { return (x < y) ? y : x; }          // glm::max
inline V3 gmin(V3 a, V3 b) { return { gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z) }; }
inline V3 gmax(V3 a, V3 b) { return { gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z) }; }
inline void put(float* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

// ------------------------------------------------------------------------------------------
// Data types (framework/include/framework/mesh.h:14-34, ray.h:9-13, src/common.h:22-60,
// src/bvh_interface.h:14-54)
// ------------------------------------------------------------------------------------------
struct Vertex {
    V3 position, normal;
    V2 texCoord;
};
struct Primitive {
    uint32_t meshID;
    Vertex v0, v1, v2;
};
static_assert(sizeof(Primitive) == 100, "Primitive must match BVHInterface::Primitive");
struct StoredPrim { // Primitive + the index it had in gather order (for canonical triangle ids)
    Primitive p;
    uint32_t orig;
};
struct Image { // framework/include/framework/image.h: rows as decoded (top row first), RGB floats
    int width = 0, height = 0;
    std::vector<V3> pixels;
};
struct Material {
    V3 kd { 0, 0, 0 }, ks { 0, 0, 0 };
    float shininess = 1.0f, transparency = 1.0f;
    int hasTexture = 0;
    std::shared_ptr<Image> kdTexture; // Material::kdTexture (framework/include/framework/mesh.h:33)
};
struct AABB {
    V3 lower { 0, 0, 0 }, upper { 1, 1, 1 };
};
struct Node {
    AABB aabb;
    uint32_t data[2];
    static constexpr uint32_t LeafBit = 1u << 31;
    bool isLeaf() const { return (data[0] & LeafBit) == LeafBit; }
    uint32_t primitiveOffset() const { return data[0] & ~LeafBit; }
    uint32_t primitiveCount() const { return data[1]; }
    uint32_t leftChild() const { return data[0]; }
    uint32_t rightChild() const { return data[1]; }
};
static_assert(sizeof(Node) == 32, "Node must match BVHInterface::Node");
struct Sphere {
    V3 center;
    float radius;
    Material material;
};
struct Light {
    int type; // 0 point, 1 segment, 2 parallelogram
    float f[21];
};
struct Ray {
    V3 origin { 0, 0, 0 }, direction { 0, 0, -1 };
    float t = FLT_MAX;
};
struct HitInfo {
    V3 normal { 0, 0, 0 }, barycentricCoord { 0, 0, 0 };
    V2 texCoord { 0, 0 };
    Material material;
    uint32_t primOrig = 0xffffffffu, meshID = 0;
};

// src/sampler.h:13-46
struct Sampler {
    uint32_t state;
    float next_1d()
    {
        state = state * 747796405u + 2891336453u;
        uint32_t v = state;
        v ^= v >> ((v >> 28u) + 4u);
        v *= 277803737u;
        v ^= v >> 22u;
        return static_cast<float>(v) / 4294967295.f;
    }
};

struct Counters {
    uint64_t c[ORC_STAT_COUNT] = { 0 };
};

} // namespace

struct orc_scene {
    std::vector<Primitive> gather; // input order
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<Sphere> spheres;
    // BVH, reference layout
    std::vector<Node> nodes;
    std::vector<StoredPrim> prims;
    uint32_t numLevels = 0, numLeaves = 0;
    bool sahBinning = false;     // features.extra.enableBvhSahBinning (src/common.h:63)
    bool buildUndefined = false; // the SAH build met a span for which the reference's bin index is undefined
};

namespace {

struct State {
    const orc_scene& scene;
    const orc_features& features;
    Sampler sampler;
    Counters* ctr;
};

// ------------------------------------------------------------------------------------------
// BVH build (src/bvh.cpp:48-95, 111-214, 381-577)
// ------------------------------------------------------------------------------------------
AABB computeSpanAABB(const StoredPrim* p, size_t n) // src/bvh.cpp:125-137
{
    if (n < 1) return {};
    V3 lower = v3(FLT_MAX), upper = v3(-FLT_MAX);
    for (size_t i = 0; i < n; i++) {
        const Primitive& q = p[i].p;
        lower = gmin(lower, gmin(gmin(q.v0.position, q.v1.position), q.v2.position));
        upper = gmax(upper, gmax(gmax(q.v0.position, q.v1.position), q.v2.position));
    }
    return { lower, upper };
}
inline V3 computePrimitiveCentroid(const Primitive& q) // src/bvh.cpp:144-152
{
    return (q.v0.position + q.v1.position + q.v2.position) / 3.0f;
}
uint32_t computeAABBLongestAxis(const AABB& b) // src/bvh.cpp:160-169
{
    V3 d = b.upper - b.lower;
    V3 axes { std::fabs(d.x), std::fabs(d.y), std::fabs(d.z) };
    if (axes.x >= axes.y && axes.x >= axes.z) return 0;
    else if (axes.y >= axes.x && axes.y >= axes.z) return 1;
    else return 2;
}
size_t splitPrimitivesByMedian(uint32_t axis, StoredPrim* p, size_t n) // src/bvh.cpp:181-214
{
    if (axis == 0)
        std::sort(p, p + n, [](const StoredPrim& a, const StoredPrim& b) { return computePrimitiveCentroid(a.p).x < computePrimitiveCentroid(b.p).x; });
    else if (axis == 1)
        std::sort(p, p + n, [](const StoredPrim& a, const StoredPrim& b) { return computePrimitiveCentroid(a.p).y < computePrimitiveCentroid(b.p).y; });
    else
        std::sort(p, p + n, [](const StoredPrim& a, const StoredPrim& b) { return computePrimitiveCentroid(a.p).z < computePrimitiveCentroid(b.p).z; });
    return (n + 1) / 2;
}
// Bin of primitive i of a span sorted along `axis` (src/extra.cpp:582-607).  The reference converts the float
// quotient to size_t; for a span whose centroids coincide on the axis that quotient is 0/0 and the conversion
// is undefined: reported through `undefined`.
size_t determineBucketIndex(size_t i, size_t nBins, uint32_t axis, const StoredPrim* p, size_t n, bool& undefined)
{
    const V3 sub = computePrimitiveCentroid(p[i].p) - computePrimitiveCentroid(p[0].p);
    const V3 len = computePrimitiveCentroid(p[n - 1].p) - computePrimitiveCentroid(p[0].p);
    float q;
    if (axis == 0) q = float(nBins) * sub.x / len.x;
    else if (axis == 1) q = float(nBins) * sub.y / len.y;
    else q = float(nBins) * sub.z / len.z;
    if (!(q >= 0.0f && q < 1.0e9f)) {
        undefined = true;
        return 0;
    }
    size_t idx = size_t(q);
    if (idx == nBins) idx--;
    return idx;
}
float calculateAABBSurfaceArea(const AABB& b) // src/extra.cpp:609-613
{
    const V3 d = b.upper - b.lower;
    const V3 axes { std::fabs(d.x), std::fabs(d.y), std::fabs(d.z) };
    return 2 * (axes.x * axes.y + axes.x * axes.z + axes.y * axes.z);
}
constexpr size_t kNoSplit = size_t(-1);
// src/extra.cpp:623-699: sort along the axis, 50 bins (fewer for spans of fewer primitives) over the centroid
// interval, the cheapest of the splits after each bin by traversal cost 1.5 + (area x count of either side) /
// area of the span's box; no split when that is not below the count.
size_t splitPrimitivesBySAHBin(const AABB& aabb, uint32_t axis, StoredPrim* p, size_t N, bool& undefined)
{
    size_t nBins = 50;
    if (N < nBins) nBins = N;
    splitPrimitivesByMedian(axis, p, N);
    std::vector<size_t> binSize(nBins, 0), binStart(nBins, 0);
    for (size_t i = 0; i < N; i++) {
        const size_t b = determineBucketIndex(i, nBins, axis, p, N, undefined);
        if (undefined || b >= nBins) {
            undefined = true;
            return kNoSplit;
        }
        binSize[b]++;
    }
    size_t relStart = 0;
    for (size_t i = 0; i < nBins; i++) {
        binStart[i] = relStart;
        relStart += binSize[i];
    }
    const float outerSurfaceArea = calculateAABBSurfaceArea(aabb);
    float minCost = std::numeric_limits<float>::max();
    size_t minIndex = kNoSplit;
    const float baseCost = float(N);
    const float travCost = 1.5f;
    for (size_t i = 0; i < nBins - 1; i++) {
        const size_t leftSize = binStart[i + 1];
        const float leftSurfaceArea = calculateAABBSurfaceArea(computeSpanAABB(p, leftSize));
        const size_t rightSize = N - leftSize;
        const float rightSurfaceArea = calculateAABBSurfaceArea(computeSpanAABB(p + leftSize, rightSize));
        const float cost = travCost + (leftSurfaceArea * float(leftSize) + rightSurfaceArea * float(rightSize)) / outerSurfaceArea;
        if (cost < minCost) {
            minCost = cost;
            minIndex = leftSize;
        }
    }
    if (minCost >= baseCost) return kNoSplit;
    return minIndex;
}
void buildRecursive(orc_scene& s, StoredPrim* p, size_t n, uint32_t nodeIndex) // src/bvh.cpp:429-512
{
    const AABB aabb = computeSpanAABB(p, n);
    size_t split = 0;
    bool leaf = n <= 4; // BVH::LeafSize, src/bvh.h:49
    if (!leaf && s.sahBinning) { // src/bvh.cpp:472-495: a span the heuristic does not split becomes a leaf of any size
        split = splitPrimitivesBySAHBin(aabb, computeAABBLongestAxis(aabb), p, n, s.buildUndefined);
        leaf = split == kNoSplit;
    }
    if (leaf) {
        // buildLeafData appends the primitives and THEN records the offset (src/bvh.cpp:387-392) ...
        s.prims.insert(s.prims.end(), p, p + n);
        Node leaf;
        leaf.aabb = aabb;
        leaf.data[0] = uint32_t(s.prims.size() | Node::LeafBit);
        leaf.data[1] = uint32_t(n);
        s.nodes[nodeIndex] = leaf;
        // ... and buildRecursive appends them a second time (src/bvh.cpp:458): the offset points at this copy.
        s.prims.insert(s.prims.end(), p, p + n);
    } else {
        if (!s.sahBinning) split = splitPrimitivesByMedian(computeAABBLongestAxis(aabb), p, n);
        const uint32_t left = uint32_t(s.nodes.size());
        s.nodes.emplace_back();
        const uint32_t right = uint32_t(s.nodes.size());
        s.nodes.emplace_back();
        Node inner;
        inner.aabb = aabb;
        inner.data[0] = left;
        inner.data[1] = right;
        s.nodes[nodeIndex] = inner;
        buildRecursive(s, p, split, left);
        buildRecursive(s, p + split, n - split, right);
    }
}
void buildBVH(orc_scene& s) // src/bvh.cpp:48-95
{
    std::vector<StoredPrim> work(s.gather.size());
    for (size_t i = 0; i < s.gather.size(); i++) work[i] = { s.gather[i], uint32_t(i) };
    s.nodes.clear();
    s.prims.clear();
    Node dflt;
    dflt.aabb = AABB {};
    dflt.data[0] = dflt.data[1] = 0;
    s.nodes.push_back(dflt); // root
    s.nodes.push_back(dflt); // dummy, never referenced (src/bvh.cpp:82)
    buildRecursive(s, work.data(), work.size(), 0);
    // buildNumLevels (src/bvh.cpp:517-549): BFS level count
    {
        std::list<uint32_t> queue { 0u };
        uint32_t level = 0;
        while (!queue.empty()) {
            size_t nAt = queue.size();
            while (nAt > 0) {
                const Node& cur = s.nodes[queue.front()];
                if (!cur.isLeaf()) {
                    queue.push_back(cur.leftChild());
                    queue.push_back(cur.rightChild());
                }
                nAt--;
                queue.pop_front();
            }
            level++;
        }
        s.numLevels = level;
    }
    // buildNumLeaves (src/bvh.cpp:553-577)
    {
        std::list<uint32_t> queue { 0u };
        uint32_t leaves = 0;
        while (!queue.empty()) {
            const Node& cur = s.nodes[queue.front()];
            if (cur.isLeaf()) leaves++;
            else {
                queue.push_back(cur.leftChild());
                queue.push_back(cur.rightChild());
            }
            queue.pop_front();
        }
        s.numLeaves = leaves;
    }
}

// ------------------------------------------------------------------------------------------
// Intersection arithmetic: bodies exist only in prebuilt/libIntersect_linux_x64.a; restated from
// its disassembly listing (symbols trianglePlane @0x2c0, intersectRayWithTriangle @0x420,
// pointInTriangle @0x0, intersectRayWithShape(AxisAlignedBox) @0xcb0, (Sphere) @0x860);
// declarations src/intersect.h:5-16.  Comparisons are written so a NaN takes the reject branch.
// ------------------------------------------------------------------------------------------
inline float mn(float a, float b) { return a < b ? a : b; } // minss a,b
inline float mx(float a, float b) { return a > b ? a : b; } // maxss a,b

bool intersectAABB(const AABB& box, Ray& ray)
{
    const V3 o = ray.origin, d = ray.direction;
    const float FMIN = 1.17549435e-38f; // 0x00800000: the archive substitutes FLT_MIN, not -FLT_MAX
    float tUx = (d.x == 0.0f) ? FLT_MAX : (box.upper.x - o.x) / d.x;
    float tUy = (d.y == 0.0f) ? FLT_MAX : (box.upper.y - o.y) / d.y;
    float tUz = (d.z == 0.0f) ? FLT_MAX : (box.upper.z - o.z) / d.z;
    float tLx = (d.x == 0.0f) ? FMIN : (box.lower.x - o.x) / d.x;
    float tLy = (d.y == 0.0f) ? FMIN : (box.lower.y - o.y) / d.y;
    float tLz = (d.z == 0.0f) ? FMIN : (box.lower.z - o.z) / d.z;
    float tin = mx(mn(tUz, tLz), mx(mn(tUy, tLy), mn(tUx, tLx)));
    float tout = mn(mx(tUz, tLz), mn(mx(tUy, tLy), mx(tUx, tLx)));
    if (0.0f > tin) { // origin inside (or box behind): report the exit distance instead
        if (!(tout > 0.0f)) return false;
        tin = tout;
        tout = FLT_MAX;
    }
    if (tin > tout) return false;
    if (0.0f > tin) return false;
    if (tin > ray.t) return false;
    ray.t = tin;
    return true;
}

bool intersectTriangle(V3 v0, V3 v1, V3 v2, Ray& ray)
{
    // trianglePlane
    V3 n = cross(v1 - v0, v2 - v0);
    n = n * (1.0f / std::sqrt(dot(n, n)));
    const float D = dot(v0, n);
    // intersectRayWithPlane (inlined in the archive)
    const float t = (D - dot(ray.origin, n)) / dot(ray.direction, n);
    if (!(t >= 0.0f)) return false;
    if (!(ray.t >= t)) return false; // equal t is accepted: later candidates win ties
    const V3 p = ray.direction * t + ray.origin;
    // pointInTriangle(v0, v1, v2, n, p): edges inclusive
    if (!(dot(cross(v2 - v0, n), p - v0) >= 0.0f)) return false;
    if (!(dot(cross(v0 - v1, n), p - v1) >= 0.0f)) return false;
    if (!(dot(cross(v1 - v2, n), p - v2) >= 0.0f)) return false;
    ray.t = t;
    return true;
}

bool intersectSphere(const Sphere& sp, Ray& ray, HitInfo& hit, uint32_t index)
{
    const V3 L = ray.origin - sp.center;
    const float h = dot(ray.direction, L);
    const float b = h + h;
    const float cc = dot(L, L) - sp.radius * sp.radius;
    const float disc = b * b - 4.0f * cc;
    if (disc < 0.0f) return false;
    float t0, t1;
    if (disc == 0.0f) {
        t0 = t1 = 0.5f * (-b);
    } else {
        const float s = std::sqrt(disc);
        t0 = (-b - s) * 0.5f;
        t1 = (s - b) * 0.5f;
    }
    float t = FLT_MAX;
    if (t0 >= 0.0f && t0 < t) t = t0;
    if (t1 >= 0.0f && t1 < t) t = t1;
    if (t < 0.0f || t >= ray.t) return false; // strict, unlike the triangle test
    hit.material = sp.material;
    hit.normal = normalize((ray.direction * t + ray.origin) - sp.center);
    hit.primOrig = 0xfffffffeu;
    hit.meshID = index;
    ray.t = t;
    return true;
}

// ------------------------------------------------------------------------------------------
// Hit record (src/bvh.cpp:18-44, src/interpolate.cpp:13-51)
// ------------------------------------------------------------------------------------------
V3 computeBarycentricCoord(V3 v0, V3 v1, V3 v2, V3 p)
{
    V3 first = v1 - v0, second = v2 - v0, point = p - v0;
    float beta = (dot(second, second) * dot(first, point) - dot(first, second) * dot(second, point))
        / (dot(first, first) * dot(second, second) - dot(first, second) * dot(first, second));
    float gamma = (dot(first, first) * dot(second, point) - dot(first, second) * dot(first, point))
        / (dot(first, first) * dot(second, second) - dot(first, second) * dot(first, second));
    return { 1.f - beta - gamma, beta, gamma };
}
inline V3 interpolateNormal(V3 n0, V3 n1, V3 n2, V3 bc) { return bc.x * n0 + bc.y * n1 + bc.z * n2; }
inline V2 interpolateTexCoord(V2 t0, V2 t1, V2 t2, V3 bc) { return bc.x * t0 + bc.y * t1 + bc.z * t2; }

void updateHitInfo(const State& st, const StoredPrim& sp, const Ray& ray, HitInfo& hit)
{
    const Primitive& pr = sp.p;
    const V3 n = normalize(cross(pr.v1.position - pr.v0.position, pr.v2.position - pr.v0.position));
    const V3 p = ray.origin + ray.t * ray.direction;
    hit.material = st.scene.materials[pr.meshID];
    hit.normal = n;
    hit.barycentricCoord = computeBarycentricCoord(pr.v0.position, pr.v1.position, pr.v2.position, p);
    if (st.features.enableNormalInterp)
        hit.normal = interpolateNormal(pr.v0.normal, pr.v1.normal, pr.v2.normal, hit.barycentricCoord);
    if (st.features.enableTextureMapping)
        hit.texCoord = interpolateTexCoord(pr.v0.texCoord, pr.v1.texCoord, pr.v2.texCoord, hit.barycentricCoord);
    if (dot(ray.direction, n) > 0.0f) hit.normal = -hit.normal;
    hit.primOrig = sp.orig;
    hit.meshID = pr.meshID;
}

// ------------------------------------------------------------------------------------------
// Traversal (src/bvh.cpp:235-370)
// ------------------------------------------------------------------------------------------
bool intersectRayWithBVH(State& st, Ray& ray, HitInfo& hit)
{
    const orc_scene& s = st.scene;
    bool is_hit = false;
    Counters* c = st.ctr;
    if (st.features.enableAccelStructure) {
        float prevT = ray.t;
        if (!intersectAABB(s.nodes[0].aabb, ray)) return false; // before the sphere loop (src/bvh.cpp:264-268)
        ray.t = prevT;
        std::vector<uint32_t> stack;
        stack.reserve(size_t(std::log2(float(s.nodes.size() + 1))) + 1);
        stack.push_back(0);
        while (!stack.empty()) {
            const Node& cur = s.nodes[stack.back()];
            stack.pop_back();
            if (!cur.isLeaf()) {
                const uint32_t iLeft = cur.leftChild(), iRight = cur.rightChild();
                if (c) c->c[ORC_STAT_NODE_PAIRS]++;
                prevT = ray.t;
                ray.t = FLT_MAX; // children are never culled against the closest hit so far
                const bool hitL = intersectAABB(s.nodes[iLeft].aabb, ray);
                const float leftT = ray.t;
                ray.t = FLT_MAX;
                const bool hitR = intersectAABB(s.nodes[iRight].aabb, ray);
                const float rightT = ray.t;
                ray.t = prevT;
                if (hitL && hitR) {
                    if (leftT < rightT) {
                        stack.push_back(iRight);
                        stack.push_back(iLeft);
                    } else { // equal distances: right child is visited first
                        stack.push_back(iLeft);
                        stack.push_back(iRight);
                    }
                } else if (hitL) {
                    stack.push_back(iLeft);
                } else if (hitR) {
                    stack.push_back(iRight);
                }
            } else {
                if (c) c->c[ORC_STAT_LEAVES]++;
                const uint32_t start = cur.primitiveOffset(), end = start + cur.primitiveCount();
                for (uint32_t i = start; i < end; i++) {
                    const StoredPrim& sp = s.prims[i];
                    if (c) c->c[ORC_STAT_TRI_TESTS]++;
                    if (intersectTriangle(sp.p.v0.position, sp.p.v1.position, sp.p.v2.position, ray)) {
                        updateHitInfo(st, sp, ray, hit);
                        is_hit = true;
                    }
                }
            }
        }
    } else {
        for (const StoredPrim& sp : s.prims) { // the doubled array: each triangle is tested twice
            if (c) c->c[ORC_STAT_TRI_TESTS]++;
            if (intersectTriangle(sp.p.v0.position, sp.p.v1.position, sp.p.v2.position, ray)) {
                updateHitInfo(st, sp, ray, hit);
                is_hit = true;
            }
        }
    }
    for (size_t i = 0; i < s.spheres.size(); i++) is_hit |= intersectSphere(s.spheres[i], ray, hit, uint32_t(i));
    return is_hit;
}

// ------------------------------------------------------------------------------------------
// Textures (src/texture.cpp:5-114)
// ------------------------------------------------------------------------------------------
inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
V3 getPixelFromCoordinates(const Image& image, int xCoord, int yCoord) // src/texture.cpp:5-11: clamped, row-flipped
{
    xCoord = clampi(xCoord, 0, image.width - 1);
    yCoord = clampi(yCoord, 0, image.height - 1);
    return image.pixels[size_t(image.height - yCoord - 1) * image.width + xCoord];
}
V3 sampleTextureNearest(const Image& image, V2 texCoord) // src/texture.cpp:20-34
{
    int xCoord = int(floorf(texCoord.x * float(image.width)));
    int yCoord = int(floorf(texCoord.y * float(image.height)));
    return getPixelFromCoordinates(image, xCoord, yCoord);
}
V3 sampleTextureBilinear(const Image& image, V2 texCoord) // src/texture.cpp:43-114
{
    int xTL = int(floorf(texCoord.x * float(image.width)));
    if (xTL >= image.width) xTL = image.width - 1;
    else if (texCoord.x * float(image.width) - float(xTL) < 0.5f) --xTL;
    if (xTL < 0) xTL = 0;
    int xTR = xTL + 1;
    if (xTR >= image.width) xTR = xTL;
    if (xTR < 0) xTR = 0;
    int yTL = int(floorf(texCoord.y * float(image.height)));
    if (yTL >= image.height) yTL = image.height - 1;
    else if (texCoord.y * float(image.height) - float(yTL) < 0.5f) --yTL;
    if (yTL < 0) yTL = 0;
    int yBL = yTL + 1;
    if (yBL >= image.height) yBL = image.height - 1;
    float xCoord = texCoord.x * float(image.width);
    if (xCoord >= float(image.width)) xCoord = float(image.width - 1);
    float yCoord = texCoord.y * float(image.height);
    if (yCoord >= float(image.height)) yCoord = float(image.height - 1);
    const float alpha = xCoord - float(xTL) - 0.5f, beta = yCoord - float(yTL) - 0.5f;
    const float coefTL = (1.f - alpha) * (1.f - beta), coefTR = alpha * (1.f - beta);
    const float coefBL = (1.f - alpha) * beta, coefBR = alpha * beta;
    V3 result = getPixelFromCoordinates(image, xTL, yTL) * coefTL;
    result = result + getPixelFromCoordinates(image, xTR, yTL) * coefTR;
    result = result + getPixelFromCoordinates(image, xTL, yBL) * coefBL;
    result = result + getPixelFromCoordinates(image, xTR, yBL) * coefBR;
    return result;
}

// ------------------------------------------------------------------------------------------
// Shading (src/shading.cpp:13-233)
// ------------------------------------------------------------------------------------------
V3 sampleMaterialKd(const State& st, const HitInfo& h) // src/shading.cpp:13-24
{
    if (st.features.enableTextureMapping && h.material.kdTexture)
        return st.features.enableBilinearTextureFiltering ? sampleTextureBilinear(*h.material.kdTexture, h.texCoord)
                                                          : sampleTextureNearest(*h.material.kdTexture, h.texCoord);
    return h.material.kd;
}

V3 computeLambertianModel(const State& st, V3, V3 lightDirection, V3 lightColor, const HitInfo& h)
{
    V3 l = normalize(lightDirection), n = normalize(h.normal);
    float d = dot(l, n);
    if (st.features.enableTransparency && h.material.transparency < 1.0f && d < 0) d = -d;
    if (d <= 0) return v3(0.0f);
    return sampleMaterialKd(st, h) * lightColor * d;
}
V3 computePhongModel(const State& st, V3 cameraDirection, V3 lightDirection, V3 lightColor, const HitInfo& h)
{
    V3 diffuse = computeLambertianModel(st, cameraDirection, lightDirection, lightColor, h);
    V3 d = normalize(-lightDirection), n = normalize(h.normal);
    float dt = dot(-d, n);
    if (st.features.enableTransparency && h.material.transparency < 1.0f && dt < 0) dt = -dt;
    if (dt <= 0) return diffuse;
    V3 v = normalize(cameraDirection);
    V3 r = normalize(d - 2.0f * dot(d, n) * n);
    float VdotR = dot(v, r);
    if (VdotR <= 0) return diffuse;
    float pw = std::pow(VdotR, h.material.shininess);
    V3 specular = pw * h.material.ks * lightColor;
    return diffuse + specular;
}
V3 computeBlinnPhongModel(const State& st, V3 cameraDirection, V3 lightDirection, V3 lightColor, const HitInfo& h)
{
    V3 diffuse = computeLambertianModel(st, cameraDirection, lightDirection, lightColor, h);
    V3 d = normalize(lightDirection), n = normalize(h.normal);
    float dt = dot(n, d);
    if (dt <= 0) return diffuse;
    V3 hv = normalize(d + cameraDirection);
    float NdotH = dot(n, hv);
    if (NdotH <= 0) return diffuse;
    float pw = std::pow(NdotH, h.material.shininess);
    V3 specular = pw * h.material.ks * lightColor;
    return diffuse + specular;
}
V3 gradientSample(float ti) // LinearGradient::sample (src/shading.cpp:185-211) over the hard-coded gradient (:32-40)
{
    struct C { float t; V3 color; };
    static const C comps[5] = {
        { 0.1f, { 215.f / 256.f, 210.f / 256.f, 203.f / 256.f } },
        { 0.22f, { 250.f / 256.f, 250.f / 256.f, 240.f / 256.f } },
        { 0.5f, { 145.f / 256.f, 170.f / 256.f, 175.f / 256.f } },
        { 0.78f, { 255.f / 256.f, 250.f / 256.f, 205.f / 256.f } },
        { 0.9f, { 170.f / 256.f, 170.f / 256.f, 170.f / 256.f } },
    }; // already ascending in t, so the reference's sort is the identity
    if (comps[0].t > ti) return comps[0].color;
    for (int i = 0; i < 4; i++) {
        if (std::fabs(comps[i].t - ti) <= 1e-6f) return comps[i].color;
        if (comps[i].t < ti && comps[i + 1].t > ti) {
            float tL = comps[i].t, tR = comps[i + 1].t;
            float alpha = (ti - tL) / (tR - tL);
            return (1.0f - alpha) * comps[i].color + alpha * comps[i + 1].color;
        }
    }
    return comps[4].color;
}
V3 computeShading(const State& st, V3 cameraDirection, V3 lightDirection, V3 lightColor, const HitInfo& h)
{
    if (st.features.enableShading) {
        switch (st.features.shadingModel) {
        case 0: return computeLambertianModel(st, cameraDirection, lightDirection, lightColor, h);
        case 1: return computePhongModel(st, cameraDirection, lightDirection, lightColor, h);
        case 2: return computeBlinnPhongModel(st, cameraDirection, lightDirection, lightColor, h);
        case 3: {
            float c = dot(normalize(lightDirection), normalize(h.normal));
            return lightColor * gradientSample(c);
        }
        }
    }
    return lightColor * sampleMaterialKd(st, h);
}

// ------------------------------------------------------------------------------------------
// Lights (src/light.cpp:25-276)
// ------------------------------------------------------------------------------------------
typedef bool (*ShadowQuery)(void* user, State& st, Ray& ray, HitInfo& hit);
bool sceneShadowQuery(void*, State& st, Ray& ray, HitInfo& hit)
{
    if (st.ctr) st.ctr->c[ORC_STAT_SHADOW]++;
    return intersectRayWithBVH(st, ray, hit);
}

void sampleSegmentLight(float sample, const float* L, V3& position, V3& color)
{
    V3 e0 = v3(L), e1 = v3(L + 3), c0 = v3(L + 6), c1 = v3(L + 9);
    position = e0 + sample * (e1 - e0);
    color = c0 + sample * (c1 - c0);
}
void sampleParallelogramLight(V2 sample, const float* L, V3& position, V3& color)
{
    V3 p0 = v3(L), e01 = v3(L + 3), e02 = v3(L + 6), c0 = v3(L + 9), c1 = v3(L + 12), c2 = v3(L + 15), c3 = v3(L + 18);
    position = p0 + sample.x * e01 + sample.y * e02;
    float a0 = (1.0f - sample.x) * (1.0f - sample.y);
    float a1 = (1.0f - sample.x) * sample.y;
    float a2 = sample.x * (1.0f - sample.y);
    float a3 = sample.x * sample.y;
    color = a0 * c0 + a1 * c1 + a2 * c2 + a3 * c3;
}
bool visibilityBinary(State& st, V3 lightPosition, const Ray& ray, ShadowQuery q, void* user) // :66-90
{
    if (!st.features.enableShadows) return true;
    V3 ip = ray.origin + ray.t * ray.direction;
    Ray lightRay;
    lightRay.origin = lightPosition;
    lightRay.direction = normalize(ip - lightPosition);
    HitInfo lh;
    bool intersects = q(user, st, lightRay, lh);
    bool expectedRayT = std::fabs(lightRay.t - length(ip - lightPosition)) <= 5e-4f;
    return intersects && expectedRayT;
}
// The reference's loop has no bound and never ends where the shaded surface is met a little off the shaded point.  Tests
// that render such scenes set a bound (orc_set_shadow_walk_cap; 0 = none, the reference's behaviour): the number of
// surfaces after which the walk stops, as the library under test documents for itself.
static int g_shadowWalkCap = 0;
V3 visibilityTransparency(State& st, V3 lightPosition, V3 lightColor, const Ray& ray, ShadowQuery q, void* user) // :107-136
{
    V3 incoming = lightColor, prevLight = lightColor;
    V3 ip = ray.origin + ray.t * ray.direction;
    bool intersects = true, finalRayT = false;
    V3 lightRayOrigin = lightPosition;
    int met = 0;
    while (intersects && !finalRayT && (g_shadowWalkCap == 0 || met++ < g_shadowWalkCap)) {
        Ray lightRay;
        lightRay.origin = lightRayOrigin;
        lightRay.direction = normalize(ip - lightRayOrigin);
        HitInfo lh;
        intersects = q(user, st, lightRay, lh);
        finalRayT = std::fabs(lightRay.t - length(ip - lightRay.origin)) <= 1e-5f;
        if (intersects) {
            prevLight = incoming;
            incoming = incoming * lh.material.kd * (1.0f - lh.material.transparency);
            lightRayOrigin = lightRay.origin + (lightRay.t + 1e-6f) * lightRay.direction;
        }
    }
    return prevLight;
}
V3 visibilityOfLightSample(State& st, V3 lightPosition, V3 lightColor, const Ray& ray, ShadowQuery q, void* user) // :247-259
{
    if (!st.features.enableShadows) return lightColor;
    else if (!st.features.enableTransparency) return visibilityBinary(st, lightPosition, ray, q, user) ? lightColor : v3(0.0f);
    else return visibilityTransparency(st, lightPosition, lightColor, ray, q, user);
}
V3 contributionPointLight(State& st, V3 lpos, V3 lcol, const Ray& ray, const HitInfo& h, ShadowQuery q, void* user) // :151-159
{
    V3 visibleLight = visibilityOfLightSample(st, lpos, lcol, ray, q, user);
    V3 p = ray.origin + ray.t * ray.direction;
    V3 l = normalize(lpos - p);
    V3 v = -ray.direction;
    return computeShading(st, v, l, visibleLight, h);
}
V3 computeLightContribution(State& st, const Ray& ray, const HitInfo& h, ShadowQuery query = sceneShadowQuery, void* user = nullptr) // :262-276
{
    V3 Lo = v3(0.0f);
    const uint32_t numSamples = st.features.numShadowSamples;
    for (const Light& L : st.scene.lights) {
        if (L.type == 0) {
            Lo += contributionPointLight(st, v3(L.f), v3(L.f + 3), ray, h, query, user);
        } else if (L.type == 1) { // :178-195 - note the sampled POSITION is passed as the light direction
            V3 contribution = v3(0.0f);
            for (uint32_t i = 0; i < numSamples; i++) {
                V3 position, color;
                sampleSegmentLight(st.sampler.next_1d(), L.f, position, color);
                V3 lightColor = visibilityOfLightSample(st, position, color, ray, query, user);
                contribution += computeShading(st, -ray.direction, position, lightColor, h) / (1.0f * numSamples);
            }
            Lo += contribution;
        } else { // :215-232
            V3 contribution = v3(0.0f);
            for (uint32_t i = 0; i < numSamples; i++) {
                V3 position, color;
                V2 smp;
                smp.x = st.sampler.next_1d();
                smp.y = st.sampler.next_1d();
                sampleParallelogramLight(smp, L.f, position, color);
                V3 lightColor = visibilityOfLightSample(st, position, color, ray, query, user);
                contribution += computeShading(st, -ray.direction, position, lightColor, h) / (1.0f * numSamples);
            }
            Lo += contribution;
        }
    }
    return Lo;
}

// ------------------------------------------------------------------------------------------
// Recursion (src/recursive.cpp:25-163, src/extra.cpp:418-517)
// ------------------------------------------------------------------------------------------
Ray generateReflectionRay(const Ray& ray, const HitInfo& h) // src/recursive.cpp:91-103
{
    Ray r;
    r.origin = ray.origin + ray.t * ray.direction;
    r.direction = ray.direction - 2.f * dot(ray.direction, h.normal) * h.normal;
    r.origin = r.origin + 0.00001f * r.direction;
    return r;
}
Ray generatePassthroughRay(const Ray& ray) // src/recursive.cpp:112-123
{
    Ray r;
    r.origin = ray.origin + ray.t * ray.direction;
    r.direction = ray.direction;
    r.origin = r.origin + 0.00001f * r.direction;
    return r;
}
void constructOrthonormalBasis(V3 r, V3 out[3]) // src/extra.cpp:418-441
{
    V3 w = normalize(r), t;
    V3 a { std::fabs(w.x), std::fabs(w.y), std::fabs(w.z) };
    if (a.x < a.y && a.x < a.z) t = { 1.0f, w.y, w.z };
    else if (a.y < a.x && a.y < a.z) t = { w.x, 1.0f, w.z };
    else t = { w.x, w.y, 1.0f };
    V3 u = normalize(cross(t, w));
    V3 v = cross(w, u);
    out[0] = w; out[1] = u; out[2] = v;
}
void sampleDisk(Sampler& smp, float radius, float& x, float& y) // src/extra.cpp:443-450
{
    const float r = radius * std::sqrt(smp.next_1d());
    const float angle = smp.next_1d() * 2 * 3.14159265358979323846264338327950288f;
    x = r * std::cos(angle);
    y = r * std::sin(angle);
}

V3 renderRay(State& st, Ray ray, int rayDepth, int rayClass);

void renderRayGlossyComponent(State& st, const Ray& ray, const HitInfo& h, V3& hitColor, int rayDepth) // src/extra.cpp:464-517
{
    const uint32_t numSamples = st.features.numGlossySamples;
    if (numSamples <= 0) return;
    const float radius = 0.5f / h.material.shininess;
    Ray r = generateReflectionRay(ray, h);
    V3 basis[3];
    constructOrthonormalBasis(r.direction, basis);
    V3 acc = v3(0.0f);
    for (uint32_t i = 0; i < numSamples; i++) {
        float px, py;
        sampleDisk(st.sampler, radius, px, py);
        Ray pr;
        pr.direction = normalize(basis[0] + px * basis[1] + py * basis[2]);
        if (dot(pr.direction, h.normal) <= 0) continue; // the two draws are consumed either way
        pr.origin = r.origin + 0.0001f * pr.direction;
        pr.t = FLT_MAX;
        acc += h.material.ks * renderRay(st, pr, rayDepth + 1, ORC_STAT_SECONDARY);
    }
    hitColor += acc / float(numSamples);
    if (st.features.glossyDebugRay) { // "visual debug" traversal whose result is never used (:510-511)
        HitInfo copy = h;
        if (st.ctr) st.ctr->c[ORC_STAT_DEBUG]++;
        intersectRayWithBVH(st, r, copy);
    }
}

V3 renderRay(State& st, Ray ray, int rayDepth, int rayClass) // src/recursive.cpp:25-83
{
    HitInfo h;
    if (st.ctr) st.ctr->c[rayClass]++;
    if (!intersectRayWithBVH(st, ray, h)) return v3(0.0f); // environment map is not restated: black
    V3 Lo = computeLightContribution(st, ray, h);
    if (rayDepth < 6) {
        bool isReflective = (h.material.ks.x != 0.0f) || (h.material.ks.y != 0.0f) || (h.material.ks.z != 0.0f);
        bool isTransparent = h.material.transparency != 1.f;
        if (st.features.enableReflections && !st.features.enableGlossyReflection && isReflective) {
            Ray r = generateReflectionRay(ray, h); // renderRaySpecularComponent, :135-143
            V3 spec = renderRay(st, r, rayDepth + 1, ORC_STAT_SECONDARY);
            Lo += spec * h.material.ks;
        }
        if (st.features.enableReflections && st.features.enableGlossyReflection && isReflective)
            renderRayGlossyComponent(st, ray, h, Lo, rayDepth);
        if (st.features.enableTransparency && isTransparent) { // renderRayTransparentComponent, :155-163
            Ray r = generatePassthroughRay(ray);
            V3 pass = renderRay(st, r, rayDepth + 1, ORC_STAT_SECONDARY);
            Lo = Lo * (1.f - h.material.transparency) + pass * h.material.transparency;
        }
    }
    return Lo;
}

// ------------------------------------------------------------------------------------------
// Camera (framework/src/trackball.cpp:71-74, 101-110; glm type_quat.inl:208-217, 347-354)
// ------------------------------------------------------------------------------------------
inline V3 quatRotate(const float* q, V3 v) // q = w,x,y,z
{
    V3 qv { q[1], q[2], q[3] };
    V3 uv = cross(qv, v);
    V3 uuv = cross(qv, uv);
    return v + ((uv * q[0]) + uuv) * 2.0f;
}
Ray cameraRay(const orc_camera& cam, float ndcX, float ndcY)
{
    V3 cs = normalize(V3 { -ndcX * cam.halfWidth, ndcY * cam.halfHeight, 1.0f });
    Ray r;
    r.origin = v3(cam.position);
    r.direction = quatRotate(cam.quat, cs);
    r.t = FLT_MAX;
    return r;
}

// src/render.cpp:56-123
void generatePixelRays(State& st, const orc_camera& cam, int px, int py, int W, int H, std::vector<Ray>& rays)
{
    rays.clear();
    const uint32_t n = st.features.numPixelSamples;
    if (n > 1) {
        if (st.features.enableJitteredSampling) {
            const uint32_t ns = uint32_t(std::round(std::sqrt(float(n))));
            for (uint32_t i = 0; i < ns; i++)
                for (uint32_t j = 0; j < ns; j++) {
                    float ox = st.sampler.next_1d();
                    float oy = st.sampler.next_1d();
                    ox /= ns;
                    oy /= ns;
                    // cellOffset = {i / numSamples, j / numSamples} is an integer division: always 0
                    float cx = float(i / ns), cy = float(j / ns);
                    float x = (float(px) + cx + ox) / float(W) * 2.f - 1.f;
                    float y = (float(py) + cy + oy) / float(H) * 2.f - 1.f;
                    rays.push_back(cameraRay(cam, x, y));
                }
        } else {
            for (uint32_t i = 0; i < n; i++) {
                float ox = st.sampler.next_1d();
                float oy = st.sampler.next_1d();
                float x = (float(px) + ox) / float(W) * 2.f - 1.f;
                float y = (float(py) + oy) / float(H) * 2.f - 1.f;
                rays.push_back(cameraRay(cam, x, y));
            }
        }
    } else {
        float x = (float(px) + 0.5f) / float(W) * 2.f - 1.f;
        float y = (float(py) + 0.5f) / float(H) * 2.f - 1.f;
        rays.push_back(cameraRay(cam, x, y));
    }
}

// src/extra.cpp:258-292
uint64_t binomial(uint32_t n, uint32_t k)
{
    uint32_t k_min = k > n - k ? n - k : k;
    uint64_t coef = 1;
    for (uint32_t i = 1; i <= k_min; i++) coef = (coef * (n - i + 1)) / i;
    return coef;
}
std::vector<float> gaussianFilter(uint32_t filterSize)
{
    const uint32_t radius = filterSize * 2 + 1;
    std::vector<float> f;
    float sum = 0.0f;
    for (uint32_t i = 0; i < radius; i++) {
        float c = float(binomial(radius - 1, i));
        f.push_back(c);
        sum += c;
    }
    for (uint32_t i = 0; i < radius; i++) f[i] /= sum;
    return f;
}
inline float perceivedLuminance(V3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } // src/extra.cpp:177-184

orc_features defaultFeatures()
{
    orc_features f;
    memset(&f, 0, sizeof(f));
    f.numPixelSamples = 1;
    f.numShadowSamples = 4;
    f.bloomFilterSize = 15;
    f.bloomFilterThreshold = 0.5f;
    f.bloomFilterIntensity = 0.8f;
    f.numGlossySamples = 1;
    return f;
}

Material matFrom(const float* m)
{
    Material r;
    r.kd = v3(m);
    r.ks = v3(m + 3);
    r.shininess = m[6];
    r.transparency = m[7];
    r.hasTexture = m[8] != 0.0f;
    return r;
}

} // namespace

// ==========================================================================================
// C entry points
// ==========================================================================================
extern "C" {

orc_scene* orc_scene_create(const uint32_t* prims25, uint32_t nPrims, const float* materials9, uint32_t nMaterials,
                            const float* lights22, uint32_t nLights, const float* spheres12, uint32_t nSpheres)
{
    orc_scene* s = new orc_scene();
    s->gather.resize(nPrims);
    if (nPrims) memcpy(s->gather.data(), prims25, size_t(nPrims) * 100);
    for (uint32_t i = 0; i < nMaterials; i++) s->materials.push_back(matFrom(materials9 + 9 * i));
    for (uint32_t i = 0; i < nPrims; i++)
        if (s->gather[i].meshID >= nMaterials) {
            delete s;
            return nullptr;
        }
    for (uint32_t i = 0; i < nLights; i++) {
        Light L;
        L.type = int(lights22[22 * i]);
        memcpy(L.f, lights22 + 22 * i + 1, sizeof(L.f));
        s->lights.push_back(L);
    }
    for (uint32_t i = 0; i < nSpheres; i++) {
        const float* p = spheres12 + 12 * i;
        Sphere sp;
        sp.center = v3(p);
        sp.radius = p[3];
        sp.material.kd = v3(p + 4);
        sp.material.ks = v3(p + 7);
        sp.material.shininess = p[10];
        sp.material.transparency = p[11];
        s->spheres.push_back(sp);
    }
    buildBVH(*s);
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
int orc_scene_rebuild_bvh(orc_scene* s, int sahBinning)
{
    s->sahBinning = sahBinning != 0;
    s->buildUndefined = false;
    buildBVH(*s);
    return s->buildUndefined ? 1 : 0;
}
void orc_set_shadow_walk_cap(int surfaces) { g_shadowWalkCap = surfaces; }
int orc_scene_set_texture(orc_scene* s, uint32_t material, const float* rgb, int width, int height)
{
    if (!s || material >= s->materials.size()) return 1;
    if (!rgb) {
        s->materials[material].kdTexture.reset();
        return 0;
    }
    if (width <= 0 || height <= 0) return 1;
    auto img = std::make_shared<Image>();
    img->width = width, img->height = height;
    img->pixels.resize(size_t(width) * height);
    for (size_t i = 0; i < img->pixels.size(); i++) img->pixels[i] = v3(rgb + 3 * i);
    s->materials[material].kdTexture = img;
    return 0;
}
void orc_sample_texture(const float* rgb, int width, int height, int bilinear, const float* uv2, float* out3)
{
    Image img;
    img.width = width, img.height = height;
    img.pixels.resize(size_t(width) * height);
    for (size_t i = 0; i < img.pixels.size(); i++) img.pixels[i] = v3(rgb + 3 * i);
    V2 uv;
    uv.x = uv2[0], uv.y = uv2[1];
    put(out3, bilinear ? sampleTextureBilinear(img, uv) : sampleTextureNearest(img, uv));
}
void orc_shading_textured(int enableShading, int model, int enableTextureMapping, int bilinear, const float* rgb, int width, int height,
                          const float* uv2, const float* cam3, const float* light3, const float* lightColor3, const float* normal3,
                          const float* kd3, const float* ks3, float shininess, float* out3)
{
    orc_scene dummy;
    orc_features f = defaultFeatures();
    f.enableShading = enableShading;
    f.shadingModel = model;
    f.enableTextureMapping = enableTextureMapping;
    f.enableBilinearTextureFiltering = bilinear;
    State st { dummy, f, { 0 }, nullptr };
    HitInfo h;
    h.normal = v3(normal3);
    h.texCoord.x = uv2[0], h.texCoord.y = uv2[1];
    h.material.kd = v3(kd3);
    h.material.ks = v3(ks3);
    h.material.shininess = shininess;
    if (rgb) {
        auto img = std::make_shared<Image>();
        img->width = width, img->height = height;
        img->pixels.resize(size_t(width) * height);
        for (size_t i = 0; i < img->pixels.size(); i++) img->pixels[i] = v3(rgb + 3 * i);
        h.material.kdTexture = img;
    }
    put(out3, computeShading(st, v3(cam3), v3(light3), v3(lightColor3), h));
}
uint32_t orc_bvh_num_nodes(const orc_scene* s) { return uint32_t(s->nodes.size()); }
uint32_t orc_bvh_num_prims(const orc_scene* s) { return uint32_t(s->prims.size()); }
uint32_t orc_bvh_num_levels(const orc_scene* s) { return s->numLevels; }
uint32_t orc_bvh_num_leaves(const orc_scene* s) { return s->numLeaves; }
void orc_bvh_copy(const orc_scene* s, uint32_t* nodes8, uint32_t* prims25, uint32_t* origIndex)
{
    if (nodes8) memcpy(nodes8, s->nodes.data(), s->nodes.size() * 32);
    for (size_t i = 0; i < s->prims.size(); i++) {
        if (prims25) memcpy(prims25 + 25 * i, &s->prims[i].p, 100);
        if (origIndex) origIndex[i] = s->prims[i].orig;
    }
}

int orc_intersect(const orc_scene* s, const orc_features* f, float* ray7, float* hit12)
{
    State st { *s, *f, { 0 }, nullptr };
    Ray r;
    r.origin = v3(ray7);
    r.direction = v3(ray7 + 3);
    r.t = ray7[6];
    HitInfo h;
    bool ok = intersectRayWithBVH(st, r, h);
    ray7[6] = r.t;
    put(hit12, h.normal);
    put(hit12 + 3, h.barycentricCoord);
    hit12[6] = h.texCoord.x;
    hit12[7] = h.texCoord.y;
    uint32_t ids[4] = { ok ? h.primOrig : 0xffffffffu, h.meshID, 0, 0 };
    memcpy(hit12 + 8, ids, 16);
    return ok ? 1 : 0;
}

int orc_render(const orc_scene* s, const orc_camera* cam, const orc_features* f, int W, int H, int startDepth,
               int y0, int y1, int yStep, float* rgb, uint32_t* primId, float* primT, uint64_t* stats, int nThreads)
{
    if (!s || !cam || !f || !rgb || W <= 0 || H <= 0) return -1;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    if (yStep < 1) yStep = 1;
    Counters total;
#ifdef _OPENMP
    if (nThreads <= 0) nThreads = omp_get_max_threads();
#else
    nThreads = 1;
#endif
    // src/render.cpp:27-44 - rows in parallel, schedule(guided); one RenderState + seeded sampler per pixel
#pragma omp parallel num_threads(nThreads)
    {
        Counters local;
        std::vector<Ray> rays;
#pragma omp for schedule(guided)
        for (int y = y0; y < y1; y += yStep) {
            for (int x = 0; x != W; x++) {
                State st { *s, *f, { static_cast<uint32_t>(H * x + y) }, stats ? &local : nullptr };
                generatePixelRays(st, *cam, x, y, W, H, rays);
                if (primId || primT) { // primary hit record of the first pixel ray (extra query, not counted)
                    Counters* keep = st.ctr;
                    st.ctr = nullptr;
                    Ray pr = rays[0];
                    HitInfo ph;
                    bool ok = intersectRayWithBVH(st, pr, ph);
                    st.ctr = keep;
                    if (primId) primId[size_t(y) * W + x] = ok ? ph.primOrig : 0xffffffffu;
                    if (primT) primT[size_t(y) * W + x] = ok ? pr.t : FLT_MAX;
                }
                V3 L = v3(0.0f); // renderRays, src/recursive.cpp:10-17
                for (const Ray& r : rays) L += renderRay(st, r, startDepth, ORC_STAT_PRIMARY);
                L = L / static_cast<float>(rays.size());
                put(rgb + 3 * (size_t(H - 1 - y) * W + x), L); // Screen::setPixel, src/screen.cpp:36-42
            }
        }
#pragma omp critical
        for (int i = 0; i < ORC_STAT_COUNT; i++) total.c[i] += local.c[i];
    }
    if (stats)
        for (int i = 0; i < ORC_STAT_COUNT; i++) stats[i] += total.c[i];
    return 0;
}

int orc_bloom(float* rgb, int W, int H, uint32_t F, float threshold, float intensity)
{
    // postprocessImageWithBloom, src/extra.cpp:330-416, on Screen-indexed buffers
    if (!rgb || W <= 0 || H <= 0 || F > 30) return -1;
    const int PW = W + 2 * int(F), PH = H + 2 * int(F);
    auto idx = [](int w, int h, int x, int y) { return size_t(h - 1 - y) * w + x; }; // Screen::indexAt
    V3* img = reinterpret_cast<V3*>(rgb);
    std::vector<V3> padded(size_t(PW) * PH, v3(0.0f));
    for (int x = int(F); x < PW - int(F); x++) // padBorders, :236-256
        for (int y = int(F); y < PH - int(F); y++) padded[idx(PW, PH, x, y)] = img[idx(W, H, x - int(F), y - int(F))];
    for (V3& p : padded) // applyThreshold, :186-196
        if (perceivedLuminance(p) < threshold) p = v3(0.0f);
    std::vector<V3> copy = padded;
    std::vector<float> filter = gaussianFilter(F);
    for (int y = 0; y < H; y++) // rows, :377-383
        for (int x = 0; x < W; x++) {
            const int X = x + int(F), Y = y + int(F);
            V3 updated = v3(0.0f);
            for (int col = X - int(F); col <= X + int(F); col++) updated += padded[idx(PW, PH, col, Y)] * v3(filter[col - X + int(F)]);
            copy[idx(PW, PH, X, Y)] = updated;
        }
    padded = copy;
    for (int x = 0; x < W; x++) // columns, :388-394
        for (int y = 0; y < H; y++) {
            const int X = x + int(F), Y = y + int(F);
            V3 updated = v3(0.0f);
            for (int row = Y - int(F); row <= Y + int(F); row++) updated += padded[idx(PW, PH, X, row)] * v3(filter[row - Y + int(F)]);
            copy[idx(PW, PH, X, Y)] = updated;
        }
    for (int x = 0; x < W; x++) // combine, :403-413
        for (int y = 0; y < H; y++) img[idx(W, H, x, y)] = img[idx(W, H, x, y)] + copy[idx(PW, PH, x + int(F), y + int(F))] * intensity;
    return 0;
}

// ---- unit entry points ----
void orc_sampler_draws(uint32_t seed, int n, float* out)
{
    Sampler s { seed };
    for (int i = 0; i < n; i++) out[i] = s.next_1d();
}
void orc_barycentric(const float* v0, const float* v1, const float* v2, const float* p, float* out3) { put(out3, computeBarycentricCoord(v3(v0), v3(v1), v3(v2), v3(p))); }
void orc_interp_normal(const float* n0, const float* n1, const float* n2, const float* bc, float* out3) { put(out3, interpolateNormal(v3(n0), v3(n1), v3(n2), v3(bc))); }
void orc_interp_uv(const float* t0, const float* t1, const float* t2, const float* bc, float* out2)
{
    V2 r = interpolateTexCoord({ t0[0], t0[1] }, { t1[0], t1[1] }, { t2[0], t2[1] }, v3(bc));
    out2[0] = r.x;
    out2[1] = r.y;
}
void orc_shading(int enableShading, int model, const float* cam3, const float* light3, const float* lightColor3,
                 const float* normal3, const float* kd3, const float* ks3, float shininess, float* out3)
{
    orc_scene dummy;
    orc_features f = defaultFeatures();
    f.enableShading = enableShading;
    f.shadingModel = model;
    State st { dummy, f, { 0 }, nullptr };
    HitInfo h;
    h.normal = v3(normal3);
    h.material.kd = v3(kd3);
    h.material.ks = v3(ks3);
    h.material.shininess = shininess;
    put(out3, computeShading(st, v3(cam3), v3(light3), v3(lightColor3), h));
}
void orc_sample_segment(const float* light12, float sample, float* pos3, float* col3)
{
    V3 p, c;
    sampleSegmentLight(sample, light12, p, c);
    put(pos3, p);
    put(col3, c);
}
void orc_sample_parallelogram(const float* light21, const float* sample2, float* pos3, float* col3)
{
    V3 p, c;
    sampleParallelogramLight({ sample2[0], sample2[1] }, light21, p, c);
    put(pos3, p);
    put(col3, c);
}
namespace {
struct Script {
    int hit;
    float t;
    Ray last;
    int calls;
};
bool scriptedQuery(void* user, State&, Ray& ray, HitInfo&)
{
    Script* s = static_cast<Script*>(user);
    s->last = ray;
    s->calls++;
    if (s->hit) ray.t = s->t;
    return s->hit != 0;
}
}
void orc_point_light(int enableShadows, int hit, float tHit, const float* lightPos3, const float* lightCol3,
                     const float* ray7, const float* normal3, const float* kd3, const float* ks3, float shininess,
                     float* out3, float* shadowRay6, int* nQueries)
{
    orc_scene dummy;
    orc_features f = defaultFeatures();
    f.enableShading = 1;
    f.enableShadows = enableShadows;
    f.shadingModel = 1;
    State st { dummy, f, { 0 }, nullptr };
    Ray r;
    r.origin = v3(ray7);
    r.direction = v3(ray7 + 3);
    r.t = ray7[6];
    HitInfo h;
    h.normal = v3(normal3);
    h.material.kd = v3(kd3);
    h.material.ks = v3(ks3);
    h.material.shininess = shininess;
    Script sc { hit, tHit, Ray {}, 0 };
    put(out3, contributionPointLight(st, v3(lightPos3), v3(lightCol3), r, h, scriptedQuery, &sc));
    put(shadowRay6, sc.calls ? sc.last.origin : v3(0.0f));
    put(shadowRay6 + 3, sc.calls ? sc.last.direction : v3(0.0f));
    *nQueries = sc.calls;
}
namespace {
// A scripted SEQUENCE of answers (7 floats each: hit, relative, value, kd3, transparency): the reported distance is
// `value`, or - with `relative` - the distance from the ray's origin to `target` plus `value`; past the script's end
// every query misses.  Every ray asked for is kept.
struct ScriptSeq {
    const float* script;
    int n;
    V3 target;
    std::vector<Ray> rays;
};
bool sequenceQuery(void* user, State&, Ray& ray, HitInfo& hit)
{
    ScriptSeq* s = static_cast<ScriptSeq*>(user);
    const int k = (int)s->rays.size();
    s->rays.push_back(ray);
    if (k >= s->n) return false;
    const float* a = s->script + 7 * k;
    if (a[0] == 0.0f) return false;
    ray.t = (a[1] != 0.0f) ? length(s->target - ray.origin) + a[2] : a[2];
    hit.material.kd = v3(a + 3);
    hit.material.transparency = a[6];
    return true;
}
void putRays(const ScriptSeq& sc, int maxRays, float* rays6, int* nQueries)
{
    *nQueries = (int)sc.rays.size();
    for (int k = 0; k < maxRays; k++) {
        put(rays6 + 6 * k, k < (int)sc.rays.size() ? sc.rays[k].origin : v3(0.0f));
        put(rays6 + 6 * k + 3, k < (int)sc.rays.size() ? sc.rays[k].direction : v3(0.0f));
    }
}
}
// computeLightContribution (src/light.cpp:262-276) over `nLights` lights (22 floats each: type, then the struct's floats)
// with every visibility query answered by the script.  out3: the contribution; nextDraw: the sampler's next draw after
// the call; rays6: the first maxRays rays asked for (origin, direction).
void orc_light_contribution(int enableShading, int enableShadows, int enableTransparency, int model, uint32_t numShadowSamples,
                            const float* lights22, int nLights, uint32_t seed, const float* ray7, const float* normal3, const float* kd3,
                            const float* ks3, float shininess, float transparency, const float* script7, int nScript, float* out3,
                            float* nextDraw, float* rays6, int maxRays, int* nQueries)
{
    orc_scene scene;
    for (int l = 0; l < nLights; l++) {
        Light L;
        L.type = (int)lights22[22 * l];
        memcpy(L.f, lights22 + 22 * l + 1, sizeof(L.f));
        scene.lights.push_back(L);
    }
    orc_features f = defaultFeatures();
    f.enableShading = enableShading, f.enableShadows = enableShadows, f.enableTransparency = enableTransparency;
    f.shadingModel = model, f.numShadowSamples = numShadowSamples;
    State st { scene, f, { seed }, nullptr };
    Ray r;
    r.origin = v3(ray7), r.direction = v3(ray7 + 3), r.t = ray7[6];
    HitInfo h;
    h.normal = v3(normal3);
    h.material.kd = v3(kd3), h.material.ks = v3(ks3);
    h.material.shininess = shininess, h.material.transparency = transparency;
    ScriptSeq sc { script7, nScript, r.origin + r.t * r.direction, {} };
    put(out3, computeLightContribution(st, r, h, sequenceQuery, &sc));
    *nextDraw = st.sampler.next_1d();
    putRays(sc, maxRays, rays6, nQueries);
}
// visibilityOfLightSampleTransparency (src/light.cpp:107-136) with the occluders scripted
void orc_visibility_transparency(const float* lightPos3, const float* lightCol3, const float* ray7, const float* script7, int nScript,
                                 float* out3, float* rays6, int maxRays, int* nQueries)
{
    orc_scene scene;
    orc_features f = defaultFeatures();
    f.enableShadows = 1, f.enableTransparency = 1;
    State st { scene, f, { 7u }, nullptr };
    Ray r;
    r.origin = v3(ray7), r.direction = v3(ray7 + 3), r.t = ray7[6];
    ScriptSeq sc { script7, nScript, r.origin + r.t * r.direction, {} };
    put(out3, visibilityTransparency(st, v3(lightPos3), v3(lightCol3), r, sequenceQuery, &sc));
    putRays(sc, maxRays, rays6, nQueries);
}
void orc_reflect(const float* ray7, const float* normal3, float* outRay7)
{
    Ray r;
    r.origin = v3(ray7);
    r.direction = v3(ray7 + 3);
    r.t = ray7[6];
    HitInfo h;
    h.normal = v3(normal3);
    Ray o = generateReflectionRay(r, h);
    put(outRay7, o.origin);
    put(outRay7 + 3, o.direction);
    outRay7[6] = o.t;
}
void orc_passthrough(const float* ray7, float* outRay7)
{
    Ray r;
    r.origin = v3(ray7);
    r.direction = v3(ray7 + 3);
    r.t = ray7[6];
    Ray o = generatePassthroughRay(r);
    put(outRay7, o.origin);
    put(outRay7 + 3, o.direction);
    outRay7[6] = o.t;
}
void orc_basis(const float* r3, float* out9)
{
    V3 b[3];
    constructOrthonormalBasis(v3(r3), b);
    put(out9, b[0]);
    put(out9 + 3, b[1]);
    put(out9 + 6, b[2]);
}
void orc_sample_disk(uint32_t seed, float radius, int n, float* out2n)
{
    Sampler s { seed };
    for (int i = 0; i < n; i++) sampleDisk(s, radius, out2n[2 * i], out2n[2 * i + 1]);
}
int orc_gaussian_filter(uint32_t filterSize, float* outTaps)
{
    std::vector<float> f = gaussianFilter(filterSize);
    memcpy(outTaps, f.data(), f.size() * 4);
    return int(f.size());
}
uint64_t orc_binomial(uint32_t n, uint32_t k) { return binomial(n, k); }
float orc_luminance(const float* rgb3) { return perceivedLuminance(v3(rgb3)); }

void orc_camera_make(const float* e, const float* lookAt3, float dist, float fovyRad, float aspect, orc_camera* out)
{
    // glm::quat(eulerAngles), type_quat.inl:208-217
    float cx = std::cos(e[0] * 0.5f), cy = std::cos(e[1] * 0.5f), cz = std::cos(e[2] * 0.5f);
    float sx = std::sin(e[0] * 0.5f), sy = std::sin(e[1] * 0.5f), sz = std::sin(e[2] * 0.5f);
    out->quat[0] = cx * cy * cz + sx * sy * sz;
    out->quat[1] = sx * cy * cz - cx * sy * sz;
    out->quat[2] = cx * sy * cz + sx * cy * sz;
    out->quat[3] = cx * cy * sz - sx * sy * cz;
    put(out->position, v3(lookAt3) + quatRotate(out->quat, V3 { 0, 0, -dist })); // Trackball::position
    out->halfHeight = std::tan(fovyRad / 2.0f);                                  // trackball.cpp:26-27
    out->halfWidth = aspect * out->halfHeight;
}
void orc_camera_ray(const orc_camera* cam, float ndcX, float ndcY, float* ray7)
{
    Ray r = cameraRay(*cam, ndcX, ndcY);
    put(ray7, r.origin);
    put(ray7 + 3, r.direction);
    ray7[6] = r.t;
}
void orc_aabb_test(const float* box6, float* ray7, int* hit)
{
    AABB b { v3(box6), v3(box6 + 3) };
    Ray r;
    r.origin = v3(ray7);
    r.direction = v3(ray7 + 3);
    r.t = ray7[6];
    *hit = intersectAABB(b, r) ? 1 : 0;
    ray7[6] = r.t;
}
int orc_triangle_test(const float* v0, const float* v1, const float* v2, float* ray7)
{
    Ray r;
    r.origin = v3(ray7);
    r.direction = v3(ray7 + 3);
    r.t = ray7[6];
    bool ok = intersectTriangle(v3(v0), v3(v1), v3(v2), r);
    ray7[6] = r.t;
    return ok ? 1 : 0;
}

} // extern "C"
