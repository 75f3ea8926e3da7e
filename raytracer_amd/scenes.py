"""Scene inputs for tests and bench.py.

Preset scenes (Cornell box, monkey, teapot, ...) are the output of the reference's own loader
(loadScenePrebuilt, src/scene.cpp:5-112, incl. centerAndScaleToUnitMesh, framework/src/mesh.cpp:151-176)
captured in tests/golden/scenes.npz by oracle/make_golden.py.  `dragon_standin` is a deterministic
procedural mesh used where the reference names data/dragon.obj, which its tree does not ship.

A scene is a dict of arrays in the layout the C ABI takes:
  prims      (N, 25) uint32  BVHInterface::Primitive records in gather order
  materials  (M, 9)  float32 kd3 ks3 shininess transparency hasTexture
  lights     (L, 22) float32 type + 21 floats
  spheres    (S, 12) float32
"""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES_NPZ = os.path.join(ROOT, "tests", "golden", "scenes.npz")

_cache = None


def preset(name: str) -> dict:
    global _cache
    if _cache is None:
        _cache = np.load(SCENES_NPZ)
    key = f"scene_{name}_prims"
    if key not in _cache.files:
        raise KeyError(f"unknown preset scene {name!r}")
    return {k: _cache[f"scene_{name}_{k}"].copy() for k in ("prims", "materials", "lights", "spheres")}


def point_light(pos, color=(1.0, 1.0, 1.0)) -> np.ndarray:
    rec = np.zeros(22, np.float32)
    rec[1:4] = pos
    rec[4:7] = color
    return rec


def pack_prims(mesh_ids, positions, normals, uvs) -> np.ndarray:
    """positions/normals (N,3,3), uvs (N,3,2), mesh_ids (N,) -> (N,25) uint32 Primitive records."""
    n = len(positions)
    rec = np.zeros((n, 25), np.float32)
    for k in range(3):
        rec[:, 1 + 8 * k:4 + 8 * k] = positions[:, k]
        rec[:, 4 + 8 * k:7 + 8 * k] = normals[:, k]
        rec[:, 7 + 8 * k:9 + 8 * k] = uvs[:, k]
    out = rec.view(np.uint32).copy()
    out[:, 0] = np.asarray(mesh_ids, np.uint32)
    return out


def unpack_positions(prims: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(prims, np.uint32).view(np.float32).reshape(-1, 25)
    return np.stack([f[:, 1:4], f[:, 9:12], f[:, 17:20]], axis=1)


def center_and_scale_unit(vertices: np.ndarray) -> np.ndarray:
    """centerAndScaleToUnitMesh (framework/src/mesh.cpp:151-176) on an (V,3) float32 vertex array:
    centre = sequential float32 sum / count, scale = 1 / max distance to the centre."""
    v = np.ascontiguousarray(vertices, np.float32)
    center = np.cumsum(v, axis=0, dtype=np.float32)[-1] / np.float32(len(v))
    d = v - center
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32)
    return (d / np.float32(dist.max())).astype(np.float32)


def dragon_standin(n_u: int = 725, n_v: int = 60, seed: int = 7) -> dict:
    """Stand-in for data/dragon.obj (absent from the reference tree; its assignment text sizes the
    dragon at 87K triangles): a (2,3) torus knot tube with deterministic bumps, n_u*n_v*2 = 87,000
    triangles by default, smooth vertex normals, centred and scaled like loadMesh(..., true), with
    the `default` material of data/default.mtl (Kd 1, Ks 0.2, Ns 10) and the Dragon preset's light
    (src/scene.cpp:88-93)."""
    u = np.arange(n_u, dtype=np.float64) * (2.0 * np.pi / n_u)
    v = np.arange(n_v, dtype=np.float64) * (2.0 * np.pi / n_v)
    p, q = 2.0, 3.0
    # knot centre line and a Frenet-like frame
    r = 1.0 + 0.45 * np.cos(q * u)
    c = np.stack([r * np.cos(p * u), r * np.sin(p * u), 0.55 * np.sin(q * u)], axis=1)
    du = 1e-4
    r2 = 1.0 + 0.45 * np.cos(q * (u + du))
    c2 = np.stack([r2 * np.cos(p * (u + du)), r2 * np.sin(p * (u + du)), 0.55 * np.sin(q * (u + du))], axis=1)
    t = c2 - c
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    b = np.cross(t, up)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    n = np.cross(b, t)
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    rad = 0.17 * (1.0 + 0.25 * np.sin(9 * uu + ph[0]) * np.sin(4 * vv + ph[1]) + 0.12 * np.sin(31 * uu + ph[2]) * np.cos(7 * vv + ph[3]))
    pos = c[:, None, :] + rad[..., None] * (np.cos(vv)[..., None] * n[:, None, :] + np.sin(vv)[..., None] * b[:, None, :])
    verts = pos.reshape(-1, 3).astype(np.float32)
    verts = center_and_scale_unit(verts)
    i = np.arange(n_u)[:, None]
    j = np.arange(n_v)[None, :]
    a00 = (i * n_v + j).ravel()
    a10 = (((i + 1) % n_u) * n_v + j).ravel()
    a01 = (i * n_v + (j + 1) % n_v).ravel()
    a11 = (((i + 1) % n_u) * n_v + (j + 1) % n_v).ravel()
    tris = np.stack([np.stack([a00, a10, a11], axis=1), np.stack([a00, a11, a01], axis=1)], axis=1).reshape(-1, 3)
    # smooth normals: area-weighted face normals accumulated per vertex
    P = verts[tris].astype(np.float64)
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    vn = np.zeros((len(verts), 3), np.float64)
    for k in range(3):
        np.add.at(vn, tris[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-30)
    vn = vn.astype(np.float32)
    positions = verts[tris]
    normals = vn[tris]
    uvs = np.zeros((len(tris), 3, 2), np.float32)
    prims = pack_prims(np.zeros(len(tris), np.uint32), positions, normals, uvs)
    materials = np.array([[1, 1, 1, 0.2, 0.2, 0.2, 10.0, 1.0, 0.0]], np.float32)
    lights = point_light((-1.0, 1.0, -1.0))[None, :]
    return dict(prims=prims, materials=materials, lights=lights, spheres=np.zeros((0, 12), np.float32))


def merge(*scenes: dict) -> dict:
    """Concatenates scenes (mesh ids re-based, lights and spheres united): the 'teapot + monkey' configuration."""
    prims, mats, lights, spheres = [], [], [], []
    base = 0
    for s in scenes:
        p = s["prims"].copy()
        p[:, 0] += np.uint32(base)
        base += len(s["materials"])
        prims.append(p)
        mats.append(s["materials"])
        lights.append(s["lights"])
        spheres.append(s["spheres"])
    lights = np.concatenate(lights)
    _, first = np.unique(lights, axis=0, return_index=True)  # a light named by several scenes is kept once
    return dict(prims=np.concatenate(prims), materials=np.concatenate(mats), lights=lights[np.sort(first)],
                spheres=np.concatenate(spheres))


def mesh_hash(scene: dict) -> str:
    return hashlib.sha256(np.ascontiguousarray(scene["prims"]).tobytes()).hexdigest()[:16]
