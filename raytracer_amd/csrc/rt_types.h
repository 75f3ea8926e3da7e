// Device-side scene layout (what rt_scene_create uploads once to HBM) and the host-side BVH
// container.  Internal to librt_amd.so.
#pragma once
#include "../../include/rt_amd.h"
#include "rt_math.h"
#include <vector>

namespace rt {

// ---- reference-layout records crossing the C ABI (src/bvh_interface.h:14-54) ----
struct RefVertex {
    float position[3], normal[3], texCoord[2];
};
struct RefPrimitive {
    uint32_t meshID;
    RefVertex v0, v1, v2;
};
static_assert(sizeof(RefPrimitive) == 100, "BVHInterface::Primitive is 100 bytes");
struct RefNode {
    float lower[3], upper[3];
    uint32_t data[2];
};
static_assert(sizeof(RefNode) == 32, "BVHInterface::Node is 32 bytes");
constexpr uint32_t kLeafBit = 1u << 31;

// ---- flattened GPU layout ----
// A child reference: leaf  = kLeafBit | count << 25 | first triangle (compact index)
//                    inner = index of the NodePair holding that node's two children
constexpr uint32_t kRefCountShift = 25;
constexpr uint32_t kRefOffsetMask = (1u << kRefCountShift) - 1;
constexpr uint32_t kMaxLeafCount = 63; // the count field's largest value; in the caller's tree it stands for "63 or more: see leafCounts"
constexpr int kStackDepth = 32; // per-lane traversal stack entries (LDS)

// Two sibling 32-byte child records {box, ref, pad} = one 64-byte, 64-byte-aligned fetch per
// inner-node step (the reference allocates siblings adjacently too, src/bvh.cpp:504-505).
struct NodePair {
    float lo0[3], hi0[3];
    uint32_t ref0, pad0;
    float lo1[3], hi1[3];
    uint32_t ref1, pad1;
};
static_assert(sizeof(NodePair) == 64, "NodePair must be one 64-byte line");

// Per-triangle constants of the reference's triangle test, hoisted out of the ray loop.  The
// reference recomputes them for every (ray, triangle) pair from the same vertices with the same
// IEEE operations, so the hoisted values are bit-identical to what it would compute.
struct TriPlane { // 16 B, read for every candidate triangle
    float n[3];   // normalize(cross(v1 - v0, v2 - v0))
    float D;      // dot(v0, n)
};
struct TriEdges { // 80 B, read only when the plane distance is inside [0, ray.t]
    float v0[3], c0x;
    float c0y, c0z, v1x, v1y;
    float v1z, c1[3];
    float v2[3], c2x;
    float c2y, c2z, pad[2];
    // c0 = cross(v2 - v0, n), c1 = cross(v0 - v1, n), c2 = cross(v1 - v2, n)
};
static_assert(sizeof(TriEdges) == 80, "TriEdges layout");
struct TriAttr { // 96 B, read once per ray for the final hit (updateHitInfo, src/bvh.cpp:18-44)
    uint32_t primId, meshID;
    float d00, d01, d11, denom; // dot(e1,e1), dot(e1,e2), dot(e2,e2), d00*d11 - d01*d01
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad[3];
};
static_assert(sizeof(TriAttr) == 96, "TriAttr layout");

// Inner node of the query tree: up to four children, boxes stored per coordinate so that one lane tests all four with
// vector loads.  An unused slot holds kEmptyRef.  128 B, 128-B aligned.
constexpr uint32_t kEmptyRef = 0xffffffffu;
struct Node4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t ref[4]; // as NodePair::ref, an inner reference being a Node4 index
    uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "Node4 layout");

// The query tree's 4-wide form for kernels that give a ray four lanes: one 32-byte record per child, so lane k of a
// quad fetches child k with two 16-byte loads and the quad reads one 128-byte line.  A leaf reference holds the index
// of a QuadLeaf instead of a first triangle.
struct QuadChild {
    float lo[3], hi[3];
    uint32_t ref, pad;
};
struct QuadNode {
    QuadChild c[4];
};
static_assert(sizeof(QuadNode) == 128, "QuadNode layout");
// The up to four triangles of one query-tree leaf, interleaved so that lane k's j-th 16-byte load of its triangle k
// lies next to the other lanes' (one 64-byte run per load and quad): plane[k], then edge words e[j][k].  Word e[4][k].z
// holds the triangle's index in query-tree order (TriCert).  Unused slots hold zeros (their plane test yields NaN).
struct QuadLeaf {
    float plane[4][4];
    float e[5][4][4];
};
static_assert(sizeof(QuadLeaf) == 384, "QuadLeaf layout");

// Per triangle of the query tree (see QueryTree): where the triangle sits in the reference's tree.
struct TriCert { // 32 B, read once per ray for the closest triangle found
    float leafLo[3];  // box of the reference leaf that holds the triangle
    uint32_t compact; // the triangle's index in reference leaf order (TriPlane / TriEdges / TriAttr of the scene)
    float leafHi[3];
    uint32_t route;   // the leaf's place in the reference's tree: 1, then one bit per level below the root (1 = second child)
};
static_assert(sizeof(TriCert) == 32, "TriCert layout");

struct GpuMaterial { // 32 B
    float kd[3], shininess;
    float ks[3], transparency;
};
struct GpuTexture { // Material::kdTexture of one mesh (16 B); pixels == null: the mesh has none
    const float* pixels; // width * height RGB floats, rows as decoded (Image::pixels, framework/src/image.cpp:33-55)
    int32_t width, height;
};
struct GpuLight { // 22 floats + pad = 96 B
    float type;
    float f[21];
    float pad[2];
};
struct GpuSphere { // 48 B
    float center[3], radius;
    float kd[3], shininess;
    float ks[3], transparency;
};

// Everything a kernel needs to read the scene; passed by value as a kernel argument.
struct SceneView {
    const NodePair* pairs;
    const TriPlane* triPlane;
    const TriEdges* triEdges;
    const TriAttr* triAttr;
    const uint32_t* bruteOrder; // compact triangle indices in primitives() array order
    const uint32_t* triReflects; // one bit per compact triangle: its mesh's material has a non-zero ks (a hit there continues with a reflection)
    const GpuMaterial* materials;
    const GpuTexture* textures; // one per material
    const GpuLight* lights;
    const GpuSphere* spheres;
    float rootLo[3], rootHi[3];
    uint32_t rootRef;
    uint32_t nPairs, nTris, nMaterials, nLights, nSpheres;
    const TriCert* cert; // null in the reference-order view; per query-tree triangle otherwise
    const uint32_t* leafCounts; // caller's tree: leafCounts[first triangle] = size of a leaf whose count field reads kMaxLeafCount (null: no such leaf)
    const Node4* nodes4; // query view: the query tree with four children per node, root = node 0 (`pairs` holds its binary form)
    const QuadNode* quadNodes; // query view: the same nodes, one record per child (null when a leaf holds more than four triangles)
    const QuadLeaf* quadLeaves;
    const NodePair* refPairs; // the reference's tree in either view
    uint32_t refRootRef;
    uint32_t fastBounds; // every box coordinate is 0 or has magnitude in [2^-60, 2^30]: reciprocal-based slab distances are exact
};

// Host-side result of flattening (filled by flatten(), uploaded by rt_scene_create)
struct FlatScene {
    std::vector<NodePair> pairs;
    std::vector<TriPlane> triPlane;
    std::vector<TriEdges> triEdges;
    std::vector<TriAttr> triAttr;
    std::vector<uint32_t> srcPos;     // per compact triangle: its place in the caller's primitives() array ...
    std::vector<uint32_t> srcPrimId;  // ... and the index it reports (TriAttr::primId); the three arrays above may be left empty
                                      // (flatten's `records` = false: the caller computes them on the device from these)
    std::vector<uint32_t> bruteOrder;
    std::vector<uint32_t> leafCounts; // empty, or one entry per compact triangle: the size of the leaf that starts there, for leaves of 63 or more
    float rootLo[3], rootHi[3];
    uint32_t rootRef = 0;
    uint32_t maxDepth = 0;
    bool fastBounds = true;
    // for the query tree
    std::vector<float> leafBox; // 6 floats per compact triangle: lower, upper of its reference leaf
    std::vector<uint32_t> leafRoute; // per compact triangle: TriCert::route of its leaf
    bool nested = true;         // every child box lies inside its parent's, every triangle inside the root box
};

// A second tree over the same triangles, built for traversal cost alone (binned surface-area heuristic).  Queries walk
// it in any order with conservative box tests; what they find is accepted only with a proof that the reference's own
// walk reports the same triangle (see trace_kernel), otherwise the ray is traced again in the reference's tree.
struct QueryTree {
    std::vector<NodePair> pairs; // the tree as built (two children per node): launches with several lanes per ray
    std::vector<Node4> nodes;    // the same tree with up to four children per node: one ray per lane
    std::vector<QuadNode> quadNodes;   // ... and in the layout for four lanes per ray (empty when a leaf holds more than four triangles)
    std::vector<QuadLeaf> quadLeaves;
    uint32_t pairRootRef = 0, pairDepth = 0;
    std::vector<TriPlane> triPlane; // copies in query-tree leaf order
    std::vector<TriEdges> triEdges;
    std::vector<TriCert> cert;
    std::vector<uint32_t> order;    // query-tree place -> compact triangle (what the three arrays above are gathered by; they stay empty
                                    // when the caller gathers on the device: build_query_tree's hostLayouts = false)
    uint32_t rootRef = 0;
    uint32_t maxDepth = 0;
    bool usable = false;
    bool builtOnDevice = false;  // the hierarchy came from device_build_query_hierarchy
    float hierarchyMs = 0.0f;    // wall time of the hierarchy build (host: the recursive build; device: upload, launches, download)
    float deviceKernelMs = 0.0f; // device: the launches alone
    float layoutMs = 0.0f;       // wall time of writing the traversal layouts from the hierarchy (host)
};
// onDevice: try the device builder first (the host builder takes over if it cannot be used)
// hostLayouts: also gather the triangle records and certificates into query-tree order on the host (the self-tests read them)
// prims: the caller's primitives() (needed when `flat` holds no triangle records); dLo4 / dHi4: the triangles' boxes already on the
// current device (4 floats per compact triangle), for the device builder
void build_query_tree(const FlatScene& flat, QueryTree& out, bool onDevice = false, bool hostLayouts = true, const RefPrimitive* prims = nullptr,
                      const float* dLo4 = nullptr, const float* dHi4 = nullptr);

// One node of a hierarchy built on the device: a child is a node index or a leaf reference (kLeafBit | count << 25 | first place in the
// triangle order).
struct DeviceBuildNode {
    float lo[3], hi[3];
    uint32_t child[2];
};
bool device_build_query_hierarchy(const float* lo4, const float* hi4, bool resident, uint32_t n, uint32_t leafMax, std::vector<DeviceBuildNode>& tree,
                                  std::vector<uint32_t>& order, uint32_t& depth, float* kernelMs);

// One span of the reference's build as settled on the device (device_build_reference_spans): its box, its place in the primitive
// order, the level it was decided at, and what became of it - split == 0: a leaf; split == kRefBuildHost: the host builds the
// subtree from the order the span was given (see ref_build.hip); otherwise the left child (record `left`) holds the first `split`
// primitives and the right child (record `left + 1`) the rest.
constexpr uint32_t kRefBuildHost = 0xffffffffu;
struct RefBuildNode {
    float lo[3], hi[3];
    uint32_t begin, count, split, left, level, pad;
};
bool device_build_reference_spans(const RefPrimitive* prims, uint32_t n, bool sah, std::vector<RefBuildNode>& tree, std::vector<uint32_t>& order,
                                  uint32_t info[4], float* kernelMs);

// Returns RT_OK or an error status with `err` describing it.
int flatten(const RefNode* nodes, uint32_t nNodes, const RefPrimitive* prims, uint32_t nPrims, const uint32_t* primIds,
            uint32_t nMaterials, FlatScene& out, const char** err, bool records = true);

} // namespace rt

// Host-side BVH in the reference layout
struct rt_bvh {
    std::vector<rt::RefNode> nodes;
    std::vector<rt::RefPrimitive> prims;
    std::vector<uint32_t> primIds;
    uint32_t numLevels = 0, numLeaves = 0;
};

namespace rt {
void set_error(const char* fmt, ...);
}
