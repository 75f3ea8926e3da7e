// The per-triangle constants of the traversal and shading kernels, from a reference-layout primitive: one function for the
// host (flatten) and the device (rt_scene_create's kernels), so that both write the same bits.
#pragma once
#include "rt_types.h"

namespace rt {

RT_HD void precompute_triangle(const RefPrimitive& p, const uint32_t primId, TriPlane& pl, TriEdges& ed, TriAttr& at)
{
    const vec3 v0 = load3(p.v0.position), v1 = load3(p.v1.position), v2 = load3(p.v2.position);
    // trianglePlane of the reference's intersect archive == the geometric normal of updateHitInfo
    vec3 n = cross(v1 - v0, v2 - v0);
    n = n * (1.0f / sqrtf(dot(n, n)));
    pl.n[0] = n.x, pl.n[1] = n.y, pl.n[2] = n.z;
    pl.D = dot(v0, n);
    // pointInTriangle edge vectors
    const vec3 c0 = cross(v2 - v0, n), c1 = cross(v0 - v1, n), c2 = cross(v1 - v2, n);
    ed.v0[0] = v0.x, ed.v0[1] = v0.y, ed.v0[2] = v0.z, ed.c0x = c0.x;
    ed.c0y = c0.y, ed.c0z = c0.z, ed.v1x = v1.x, ed.v1y = v1.y;
    ed.v1z = v1.z, ed.c1[0] = c1.x, ed.c1[1] = c1.y, ed.c1[2] = c1.z;
    ed.v2[0] = v2.x, ed.v2[1] = v2.y, ed.v2[2] = v2.z, ed.c2x = c2.x;
    ed.c2y = c2.y, ed.c2z = c2.z, ed.pad[0] = ed.pad[1] = 0.0f;
    // computeBarycentricCoord constants (src/interpolate.cpp:13-23)
    const vec3 first = v1 - v0, second = v2 - v0;
    at.primId = primId;
    at.meshID = p.meshID;
    at.d00 = dot(first, first);
    at.d01 = dot(first, second);
    at.d11 = dot(second, second);
    at.denom = at.d00 * at.d11 - at.d01 * at.d01;
    for (int k = 0; k < 3; k++) at.n0[k] = p.v0.normal[k], at.n1[k] = p.v1.normal[k], at.n2[k] = p.v2.normal[k];
    for (int k = 0; k < 2; k++) at.uv0[k] = p.v0.texCoord[k], at.uv1[k] = p.v1.texCoord[k], at.uv2[k] = p.v2.texCoord[k];
    at.pad[0] = at.pad[1] = at.pad[2] = 0.0f;
}

} // namespace rt
