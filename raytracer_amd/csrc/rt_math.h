// fp32 vector arithmetic with the operation order of the reference's vendored glm 0.9.9
// (framework/third_party/glm/glm/detail/func_geometric.inl:48-89).  Float ties decide which
// triangle a ray reports, so nothing here may be contracted into FMAs or reassociated:
// every translation unit including this header is built with -ffp-contract=off.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct vec3 {
    float x, y, z;
};
struct vec2 {
    float x, y;
};

RT_HD vec3 make3(float x, float y, float z) { return vec3 { x, y, z }; }
RT_HD vec3 splat(float a) { return vec3 { a, a, a }; }
RT_HD vec3 load3(const float* p) { return vec3 { p[0], p[1], p[2] }; }
RT_HD vec3 operator+(vec3 a, vec3 b) { return vec3 { a.x + b.x, a.y + b.y, a.z + b.z }; }
RT_HD vec3 operator-(vec3 a, vec3 b) { return vec3 { a.x - b.x, a.y - b.y, a.z - b.z }; }
RT_HD vec3 operator-(vec3 a) { return vec3 { -a.x, -a.y, -a.z }; }
RT_HD vec3 operator*(vec3 a, vec3 b) { return vec3 { a.x * b.x, a.y * b.y, a.z * b.z }; }
RT_HD vec3 operator*(float s, vec3 v) { return vec3 { s * v.x, s * v.y, s * v.z }; }
RT_HD vec3 operator*(vec3 v, float s) { return vec3 { v.x * s, v.y * s, v.z * s }; }
RT_HD vec3 operator/(vec3 v, float s) { return vec3 { v.x / s, v.y / s, v.z / s }; }
RT_HD vec2 operator+(vec2 a, vec2 b) { return vec2 { a.x + b.x, a.y + b.y }; }
RT_HD vec2 operator*(float s, vec2 v) { return vec2 { s * v.x, s * v.y }; }
// glm::dot: tmp = a*b; tmp.x + tmp.y + tmp.z
RT_HD float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD vec3 cross(vec3 x, vec3 y) { return vec3 { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y }; }
RT_HD float length(vec3 v) { return sqrtf(dot(v, v)); }
// glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
RT_HD vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// glm::min / glm::max
RT_HD float gmin(float x, float y) { return (y < x) ? y : x; }
RT_HD float gmax(float x, float y) { return (x < y) ? y : x; }
RT_HD vec3 gmin(vec3 a, vec3 b) { return vec3 { gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z) }; }
RT_HD vec3 gmax(vec3 a, vec3 b) { return vec3 { gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z) }; }
// x86 minss / maxss operand semantics used by the reference's box test
RT_HD float sse_min(float a, float b) { return a < b ? a : b; }
RT_HD float sse_max(float a, float b) { return a > b ? a : b; }

// glm quat * vec3 (type_quat.inl:347-354); q = {w, x, y, z}
RT_HD vec3 quat_rotate(const float* q, vec3 v)
{
    vec3 qv = make3(q[1], q[2], q[3]);
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q[0]) + uuv) * 2.0f;
}

} // namespace rt
