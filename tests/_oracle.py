"""ctypes access to oracle/liboracle.so - the CPU checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)

STAT_NAMES = ("primary", "shadow", "secondary", "debug", "node_pairs", "tri_tests", "leaves")


class OrcFeatures(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "enableShading", "enableReflections", "enableShadows", "enableNormalInterp", "enableTextureMapping",
        "enableAccelStructure", "enableBilinearTextureFiltering", "enableTransparency", "enableJitteredSampling",
        "shadingModel")] + [
        ("numPixelSamples", C.c_uint32), ("numShadowSamples", C.c_uint32),
        ("enableBloomEffect", C.c_int32), ("enableGlossyReflection", C.c_int32),
        ("bloomFilterSize", C.c_uint32), ("bloomFilterThreshold", C.c_float), ("bloomFilterIntensity", C.c_float),
        ("numGlossySamples", C.c_uint32), ("glossyDebugRay", C.c_int32), ("reserved", C.c_int32)]


class OrcCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("quat", C.c_float * 4), ("halfWidth", C.c_float), ("halfHeight", C.c_float)]


def make_features(**kw) -> OrcFeatures:
    f = OrcFeatures()
    f.numPixelSamples, f.numShadowSamples = 1, 4
    f.bloomFilterSize, f.bloomFilterThreshold, f.bloomFilterIntensity = 15, 0.5, 0.8
    f.numGlossySamples = 1
    for k, v in kw.items():
        if not hasattr(f, k):
            raise AttributeError(k)
        setattr(f, k, v)
    return f


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = [os.path.join(ORACLE_DIR, n) for n in ("oracle.cpp", "oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
    L = C.CDLL(so)
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [u32p, C.c_uint32, f32p, C.c_uint32, f32p, C.c_uint32, f32p, C.c_uint32]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_set_shadow_walk_cap.argtypes = [C.c_int]
    L.orc_scene_rebuild_bvh.argtypes = [C.c_void_p, C.c_int]
    for n in ("orc_bvh_num_nodes", "orc_bvh_num_prims", "orc_bvh_num_levels", "orc_bvh_num_leaves"):
        getattr(L, n).restype = C.c_uint32
        getattr(L, n).argtypes = [C.c_void_p]
    L.orc_bvh_copy.argtypes = [C.c_void_p, u32p, u32p, u32p]
    L.orc_intersect.argtypes = [C.c_void_p, C.POINTER(OrcFeatures), f32p, f32p]
    L.orc_render.argtypes = [C.c_void_p, C.POINTER(OrcCamera), C.POINTER(OrcFeatures), C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_int, f32p, u32p, f32p, u64p, C.c_int]
    L.orc_bloom.argtypes = [f32p, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float]
    L.orc_sampler_draws.argtypes = [C.c_uint32, C.c_int, f32p]
    L.orc_barycentric.argtypes = [f32p] * 5
    L.orc_interp_normal.argtypes = [f32p] * 5
    L.orc_interp_uv.argtypes = [f32p] * 5
    L.orc_shading.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, C.c_float, f32p]
    L.orc_scene_set_texture.argtypes = [C.c_void_p, C.c_uint32, f32p, C.c_int, C.c_int]
    L.orc_sample_texture.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p]
    L.orc_shading_textured.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p,
                                       f32p, C.c_float, f32p]
    L.orc_sample_segment.argtypes = [f32p, C.c_float, f32p, f32p]
    L.orc_sample_parallelogram.argtypes = [f32p, f32p, f32p, f32p]
    L.orc_light_contribution.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, f32p, C.c_int, C.c_uint32, f32p, f32p, f32p, f32p,
                                         C.c_float, C.c_float, f32p, C.c_int, f32p, f32p, f32p, C.c_int, C.POINTER(C.c_int)]
    L.orc_visibility_transparency.argtypes = [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, C.POINTER(C.c_int)]
    L.orc_point_light.argtypes = [C.c_int, C.c_int, C.c_float, f32p, f32p, f32p, f32p, f32p, f32p, C.c_float, f32p, f32p,
                                  C.POINTER(C.c_int)]
    L.orc_reflect.argtypes = [f32p, f32p, f32p]
    L.orc_passthrough.argtypes = [f32p, f32p]
    L.orc_basis.argtypes = [f32p, f32p]
    L.orc_sample_disk.argtypes = [C.c_uint32, C.c_float, C.c_int, f32p]
    L.orc_gaussian_filter.argtypes = [C.c_uint32, f32p]
    L.orc_binomial.restype = C.c_uint64
    L.orc_binomial.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_luminance.restype = C.c_float
    L.orc_luminance.argtypes = [f32p]
    L.orc_camera_make.argtypes = [f32p, f32p, C.c_float, C.c_float, C.c_float, C.POINTER(OrcCamera)]
    L.orc_camera_ray.argtypes = [C.POINTER(OrcCamera), C.c_float, C.c_float, f32p]
    L.orc_aabb_test.argtypes = [f32p, f32p, C.POINTER(C.c_int)]
    L.orc_triangle_test.restype = C.c_int
    L.orc_triangle_test.argtypes = [f32p, f32p, f32p, f32p]
    _lib = L
    return L


def fa(x):
    """contiguous float32 array + pointer"""
    a = np.ascontiguousarray(x, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


DEG2RAD = np.float32(0.01745329251994329576923690768489)  # glm::radians


def make_camera(W, H, fov_deg=50.0, dist=3.0, look_at=(0, 0, 0), rot_deg=(20.0, 20.0, 0.0)) -> OrcCamera:
    """Default CameraConfig (reference src/config.h:18-23) through Trackball's formulas."""
    cam = OrcCamera()
    e, ep = fa(np.asarray(rot_deg, np.float32) * DEG2RAD)
    la, lap = fa(look_at)
    fovy = np.float32(fov_deg) * DEG2RAD
    aspect = np.float32(W) / np.float32(H)
    lib().orc_camera_make(ep, lap, C.c_float(dist), C.c_float(float(fovy)), C.c_float(float(aspect)), C.byref(cam))
    return cam


class OracleScene:
    def __init__(self, prims, materials, lights, spheres=None, textures=None, sah_binning=False):
        L = lib()
        self.prims = np.ascontiguousarray(prims, np.uint32).reshape(-1, 25)
        self.materials = np.ascontiguousarray(materials, np.float32).reshape(-1, 9)
        self.lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 22)
        self.spheres = np.ascontiguousarray(spheres if spheres is not None else np.zeros((0, 12)), np.float32).reshape(-1, 12)
        self.h = L.orc_scene_create(_ptr(self.prims, u32p), len(self.prims), _ptr(self.materials, f32p), len(self.materials),
                                    _ptr(self.lights, f32p), len(self.lights), _ptr(self.spheres, f32p), len(self.spheres))
        if not self.h:
            raise RuntimeError("orc_scene_create failed")
        if sah_binning and L.orc_scene_rebuild_bvh(self.h, 1) != 0:
            raise RuntimeError("orc_scene_rebuild_bvh: a span's centroids coincide on its split axis (undefined in the reference)")
        for m, img in (textures or {}).items():  # {mesh index: (H, W, 3) float32, rows as decoded}
            a = np.ascontiguousarray(img, np.float32)
            if L.orc_scene_set_texture(self.h, int(m), _ptr(a, f32p), a.shape[1], a.shape[0]) != 0:
                raise RuntimeError("orc_scene_set_texture failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def bvh(self):
        L = lib()
        nn, npr = L.orc_bvh_num_nodes(self.h), L.orc_bvh_num_prims(self.h)
        nodes = np.zeros((nn, 8), np.uint32)
        prims = np.zeros((npr, 25), np.uint32)
        orig = np.zeros(npr, np.uint32)
        L.orc_bvh_copy(self.h, _ptr(nodes, u32p), _ptr(prims, u32p), _ptr(orig, u32p))
        return dict(nodes=nodes, prims=prims, orig=orig, levels=L.orc_bvh_num_levels(self.h), leaves=L.orc_bvh_num_leaves(self.h))

    def intersect(self, features, origin, direction, t=np.finfo(np.float32).max):
        ray = np.array(list(origin) + list(direction) + [t], np.float32)
        hit = np.zeros(12, np.float32)
        ok = lib().orc_intersect(self.h, C.byref(features), _ptr(ray, f32p), _ptr(hit, f32p))
        ids = hit[8:].view(np.uint32)
        return bool(ok), float(ray[6]), dict(normal=hit[0:3].copy(), bary=hit[3:6].copy(), uv=hit[6:8].copy(),
                                             prim=int(ids[0]), mesh=int(ids[1]))

    def render(self, cam, features, W, H, start_depth=0, rows=None, want_ids=True, threads=0, row_step=1):
        rgb = np.zeros((H, W, 3), np.float32)
        pid = np.full((H, W), 0xFFFFFFFF, np.uint32) if want_ids else None
        pt = np.full((H, W), np.finfo(np.float32).max, np.float32) if want_ids else None
        stats = np.zeros(len(STAT_NAMES), np.uint64)
        y0, y1 = rows if rows else (0, H)
        rc = lib().orc_render(self.h, C.byref(cam), C.byref(features), W, H, start_depth, y0, y1, row_step, _ptr(rgb, f32p),
                              _ptr(pid, u32p), _ptr(pt, f32p), _ptr(stats, u64p), threads)
        if rc != 0:
            raise RuntimeError("orc_render failed")
        return dict(rgb=rgb, prim=pid, t=pt, stats=dict(zip(STAT_NAMES, (int(v) for v in stats))))


def bloom(rgb, F=15, threshold=0.5, intensity=0.8):
    out = np.ascontiguousarray(rgb, np.float32).copy()
    H, W, _ = out.shape
    rc = lib().orc_bloom(_ptr(out, f32p), W, H, F, threshold, intensity)
    if rc != 0:
        raise RuntimeError("orc_bloom failed")
    return out


_scenes = None


def load_scene_arrays(name):
    """Reference-loader output committed under tests/golden/scenes.npz."""
    global _scenes
    if _scenes is None:
        _scenes = np.load(os.path.join(GOLDEN_DIR, "scenes.npz"))
    return tuple(_scenes[f"scene_{name}_{k}"] for k in ("prims", "materials", "lights", "spheres"))


def oracle_scene(name) -> OracleScene:
    return OracleScene(*load_scene_arrays(name))
