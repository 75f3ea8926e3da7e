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
  textures   {mesh index: (H, W, 3) float32}  Material::kdTexture as Image::pixels (rows as decoded); optional
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
    sc = {k: _cache[f"scene_{name}_{k}"].copy() for k in ("prims", "materials", "lights", "spheres")}
    # every preset material with a texture names data/default.png (data/default.mtl:17): the reference's own decode of it
    tex = [m for m in range(len(sc["materials"])) if sc["materials"][m, 8] != 0.0]
    if tex:
        default_png = np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))["texture_default"]
        sc["textures"] = {m: default_png.copy() for m in tex}
    return sc


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
    textures = {}
    base = 0
    for s in scenes:
        p = s["prims"].copy()
        p[:, 0] += np.uint32(base)
        textures.update({base + m: img for m, img in s.get("textures", {}).items()})
        base += len(s["materials"])
        prims.append(p)
        mats.append(s["materials"])
        lights.append(s["lights"])
        spheres.append(s["spheres"])
    lights = np.concatenate(lights)
    _, first = np.unique(lights, axis=0, return_index=True)  # a light named by several scenes is kept once
    out = dict(prims=np.concatenate(prims), materials=np.concatenate(mats), lights=lights[np.sort(first)],
               spheres=np.concatenate(spheres))
    if textures:
        out["textures"] = textures
    return out


def mesh_hash(scene: dict) -> str:
    return hashlib.sha256(np.ascontiguousarray(scene["prims"]).tobytes()).hexdigest()[:16]


def _parse_mtl(path: str) -> dict:
    mats, cur = {}, None
    if not os.path.exists(path):
        return mats
    for line in open(path, errors="replace"):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "newmtl":
            # tinyobj's material defaults: diffuse/specular 0, shininess 1, dissolve 1
            cur = mats.setdefault(" ".join(t[1:]), dict(kd=[0, 0, 0], ks=[0, 0, 0], ns=1.0, d=1.0, tex=0.0))
        elif cur is not None:
            if t[0] == "Kd":
                cur["kd"] = [float(x) for x in t[1:4]]
            elif t[0] == "Ks":
                cur["ks"] = [float(x) for x in t[1:4]]
            elif t[0] == "Ns":
                cur["ns"] = float(t[1])
            elif t[0] == "d":
                cur["d"] = float(t[1])
            elif t[0] == "Tr":
                cur["d"] = 1.0 - float(t[1])
            elif t[0] == "map_Kd":
                cur["tex"] = 1.0
                cur["map_kd"] = os.path.join(os.path.dirname(os.path.abspath(path)), " ".join(t[1:]))
    return mats


def load_image(path: str) -> np.ndarray:
    """Image::Image (framework/src/image.cpp:33-55): 8-bit RGB / 255.0f, rows as decoded."""
    from PIL import Image  # decoding only (PNG decodes are exact; JPEG decoders may differ from stb_image in the last bit)
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), np.uint8)
    return (a.astype(np.float32) / np.float32(255.0)).astype(np.float32)


def load_obj(path: str, normalize: bool = True, light=(-1.0, 1.0, -1.0)) -> dict:
    """loadMesh(file, normalize) (framework/src/mesh.cpp:52-149) for a Wavefront OBJ: one mesh per run of faces
    with the same material inside a shape, quads split as tinyobjloader does, a vertex without a
    normal gets its triangle's geometric normal, and - with `normalize` - the vertices (deduplicated per mesh
    on position+normal+texcoord, as the loader's vertex cache does) are centred on their mean and scaled by
    the largest distance to it (mesh.cpp:151-176).  The light defaults to the Dragon preset's (src/scene.cpp:88-93)."""
    V, VN, VT = [], [], []
    meshes = []  # (material name, [((vi, ti, ni) * 3)])
    mtl = {}
    cur_mat, cur_faces, shape_open = None, None, False

    def flush():
        nonlocal cur_faces
        if cur_faces:
            meshes.append((cur_mat, cur_faces))
        cur_faces = []

    def idx(tok, n):
        i = int(tok)
        return i - 1 if i > 0 else n + i

    base = os.path.dirname(os.path.abspath(path))
    cur_faces = []
    for line in open(path, errors="replace"):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        if t[0] == "v":
            V.append([float(x) for x in t[1:4]])
        elif t[0] == "vn":
            VN.append([float(x) for x in t[1:4]])
        elif t[0] == "vt":
            VT.append([float(x) for x in t[1:3]])
        elif t[0] == "mtllib":
            mtl.update(_parse_mtl(os.path.join(base, " ".join(t[1:]))))
        elif t[0] == "usemtl":
            flush()
            cur_mat = " ".join(t[1:])
        elif t[0] in ("o", "g"):
            flush()
        elif t[0] == "f":
            corners = []
            for c in t[1:]:
                p = c.split("/")
                vi = idx(p[0], len(V))
                ti = idx(p[1], len(VT)) if len(p) > 1 and p[1] else -1
                ni = idx(p[2], len(VN)) if len(p) > 2 and p[2] else -1
                corners.append((vi, ti, ni))
            if len(corners) == 4:
                # tinyobjloader splits a quad along its shorter diagonal (tiny_obj_loader.h:1463-1524), in float32
                q = np.asarray([V[c[0]] for c in corners], np.float32)
                e02, e13 = q[2] - q[0], q[3] - q[1]
                s02 = (e02[0] * e02[0] + e02[1] * e02[1]) + e02[2] * e02[2]
                s13 = (e13[0] * e13[0] + e13[1] * e13[1]) + e13[2] * e13[2]
                if s02 < s13:
                    cur_faces += [(corners[0], corners[1], corners[2]), (corners[0], corners[2], corners[3])]
                else:
                    cur_faces += [(corners[0], corners[1], corners[3]), (corners[1], corners[2], corners[3])]
            else:  # triangles; larger polygons as a fan (tinyobjloader clips ears: only convex polygons agree)
                for k in range(1, len(corners) - 1):
                    cur_faces.append((corners[0], corners[k], corners[k + 1]))
    flush()
    if not meshes:
        raise ValueError(f"{path}: no faces")
    V = np.asarray(V, np.float32)
    VN = np.asarray(VN, np.float32).reshape(-1, 3)
    VT = np.asarray(VT, np.float32).reshape(-1, 2)
    out_pos, out_nrm, out_uv, out_id, materials, dedup = [], [], [], [], [], []
    textures = {}
    for m, (name, faces) in enumerate(meshes):
        f = np.asarray(faces, np.int64)  # (T, 3, 3)
        pos = V[f[:, :, 0]]
        geo = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]).astype(np.float32)
        geo = (geo * (np.float32(1) / np.sqrt((geo[:, 0] * geo[:, 0] + geo[:, 1] * geo[:, 1]) + geo[:, 2] * geo[:, 2], dtype=np.float32))[:, None]).astype(np.float32)
        nrm = np.repeat(geo[:, None, :], 3, axis=1)
        has_n = f[:, :, 2] >= 0
        if len(VN) and has_n.any():
            nrm[has_n] = VN[f[:, :, 2][has_n]]
        uv = np.zeros((len(f), 3, 2), np.float32)
        has_t = f[:, :, 1] >= 0
        if len(VT) and has_t.any():
            uv[has_t] = VT[f[:, :, 1][has_t]]
        out_pos.append(pos), out_nrm.append(nrm), out_uv.append(uv), out_id.append(np.full(len(f), m, np.uint32))
        # the loader's vertex cache: first occurrence order of distinct (position, normal, texcoord)
        key = np.concatenate([pos.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2)], axis=1).view(np.uint32)
        _, first = np.unique(key, axis=0, return_index=True)
        dedup.append(pos.reshape(-1, 3)[np.sort(first)])
        mm = mtl.get(name)
        materials.append([1, 1, 1, 0, 0, 0, 1.0, 1.0, 0.0] if mm is None else [*mm["kd"], *mm["ks"], mm["ns"], mm["d"], mm["tex"]])
        if mm is not None and mm.get("map_kd") and os.path.exists(mm["map_kd"]):
            textures[m] = load_image(mm["map_kd"])
    pos = np.concatenate(out_pos)
    if normalize:
        allv = np.concatenate(dedup).astype(np.float32)
        center = np.cumsum(allv, axis=0, dtype=np.float32)[-1] / np.float32(len(allv))
        d = allv - center
        maxd = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], dtype=np.float32).max()
        pos = ((pos - center) / np.float32(maxd)).astype(np.float32)
    prims = pack_prims(np.concatenate(out_id), pos, np.concatenate(out_nrm), np.concatenate(out_uv))
    out = dict(prims=prims, materials=np.asarray(materials, np.float32), lights=point_light(light)[None, :],
               spheres=np.zeros((0, 12), np.float32))
    if textures:
        out["textures"] = textures
    return out
