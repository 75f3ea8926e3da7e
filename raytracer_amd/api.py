"""ctypes binding of librt_amd.so (include/rt_amd.h) - harness plumbing for tests and bench.py.

The library is the product; this module only marshals arrays.  There is no CPU fallback: if the
shared library is missing, loading raises."""
import ctypes as C
import weakref
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "lib", "librt_amd.so")  # RT_AMD_LIB: an experimental build

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)

RT_OK, RT_ERR_INVALID, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_NOMEM, RT_ERR_LIMIT, RT_ERR_AGAIN = range(7)
RT_RENDER_COUNT_WORK, RT_RENDER_NO_CULL, RT_RENDER_SERIAL, RT_RENDER_REFERENCE_TREE, RT_RENDER_OVERLAPPED = 1, 2, 4, 8, 16
RT_PIPELINE_DEPTH_FIRST, RT_PIPELINE_ROUNDS, RT_PIPELINE_PERSISTENT = 0, 1, 2
RT_RENDER_COUNT_SCHEDULE = 32
RT_BLOOM_STATIC_MAX_PIXELS = 1 << 21
RT_MAX_FRAMES = 16
NO_HIT = 0xFFFFFFFF
FLT_MAX = float(np.finfo(np.float32).max)


class Features(C.Structure):
    """rt_features: POD mirror of the reference's Features/ExtraFeatures."""
    _fields_ = [(n, C.c_int32) for n in (
        "enableShading", "enableReflections", "enableShadows", "enableNormalInterp", "enableTextureMapping",
        "enableAccelStructure", "enableBilinearTextureFiltering", "enableTransparency", "enableJitteredSampling",
        "shadingModel")] + [
        ("numPixelSamples", C.c_uint32), ("numShadowSamples", C.c_uint32),
        ("enableBloomEffect", C.c_int32), ("enableGlossyReflection", C.c_int32),
        ("bloomFilterSize", C.c_uint32), ("bloomFilterThreshold", C.c_float), ("bloomFilterIntensity", C.c_float),
        ("numGlossySamples", C.c_uint32), ("reserved", C.c_int32 * 2)]

    def __init__(self, **kw):
        super().__init__()
        # defaults of the reference structs (src/common.h:62-115)
        self.numPixelSamples, self.numShadowSamples = 1, 4
        self.bloomFilterSize, self.bloomFilterThreshold, self.bloomFilterIntensity = 15, 0.5, 0.8
        self.numGlossySamples = 1
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("quat", C.c_float * 4), ("halfWidth", C.c_float), ("halfHeight", C.c_float)]


class Region(C.Structure):
    _fields_ = [("stripeRows", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primaryRays", "shadowRays", "secondaryRays", "nodePairSteps", "triangleTests",
                                          "pixels", "stackOverflows", "referenceTreeQueries", "pipeline", "queryNodeSteps")]

    @property
    def rays(self):
        return self.primaryRays + self.shadowRays + self.secondaryRays

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt status {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Loads librt_amd.so; raises if it has not been built (python raytracer_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(-1, f"{LIB_PATH} is missing: build it with `python raytracer_amd/build.py` (no CPU fallback exists)")
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (same SONAME as the
    # system one).  Whichever is loaded first serves both, but only if torch comes first - loaded
    # second it brings a second runtime and then sees no GPU.  So a process that may use torch
    # (tests, bench.py) imports it before this library unless RT_AMD_NO_TORCH is set.
    if "torch" not in sys.modules and not os.environ.get("RT_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_last_error.restype = C.c_char_p
    L.rt_bvh_build.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.rt_bvh_build_sah_binning.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.rt_bvh_build_device.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint32)]
    L.rt_bvh_destroy.argtypes = [vp]
    for n in ("rt_bvh_num_nodes", "rt_bvh_num_primitives", "rt_bvh_num_levels", "rt_bvh_num_leaves"):
        getattr(L, n).restype = C.c_uint32
        getattr(L, n).argtypes = [vp]
    for n in ("rt_bvh_nodes", "rt_bvh_primitives", "rt_bvh_primitive_ids"):
        getattr(L, n).restype = vp
        getattr(L, n).argtypes = [vp]
    L.rt_camera_from_trackball.argtypes = [C.c_float, C.c_float, f32p, f32p, C.c_float, C.POINTER(Camera)]
    L.rt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.rt_set_device.argtypes = [C.c_int]
    L.rt_scene_create.argtypes = [vp, C.c_uint32, vp, C.c_uint32, u32p, f32p, C.c_uint32, f32p, C.c_uint32, f32p, C.c_uint32,
                                  C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_trim.argtypes = [vp]
    L.rt_scene_set_texture.argtypes = [vp, C.c_uint32, f32p, C.c_int, C.c_int]
    L.rt_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Features), C.c_int, C.c_int, C.c_int, C.POINTER(Region), C.c_uint32,
                            vp, vp, vp, vp]
    L.rt_render_frames.argtypes = [vp, C.POINTER(Camera), C.c_int, C.POINTER(Features), C.c_int, C.c_int, C.c_int, C.POINTER(Region),
                                   C.c_uint32, vp, vp, vp, vp]
    L.rt_bloom_scratch_bytes.restype = C.c_size_t
    L.rt_bloom_scratch_bytes.argtypes = [C.c_int, C.c_int]
    L.rt_bloom.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float, C.POINTER(Region), vp]
    L.rt_bloom_static.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float, vp]
    L.rt_sync.argtypes = [vp, vp, C.POINTER(Stats)]
    L.rt_render_image.argtypes = [vp, C.POINTER(Camera), C.POINTER(Features), C.c_int, C.c_int, C.c_int, C.c_uint32, f32p, u32p,
                                  f32p, C.POINTER(Stats)]
    L.rt_render_images.argtypes = [vp, C.POINTER(Camera), C.c_int, C.POINTER(Features), C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(f32p),
                                   C.POINTER(Stats)]
    L.rt_render_images_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Camera), C.c_int, C.POINTER(Features), C.c_int, C.c_int, C.c_int,
                                         C.c_uint32, C.POINTER(f32p), C.POINTER(Stats)]
    L.rt_scene_clone.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rt_scene_device.argtypes = [vp, C.POINTER(C.c_int)]
    L.rt_scene_stream.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rt_stripe_owner.argtypes = [C.c_int, C.c_int, C.c_int]
    L.rt_stripe_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.rt_intersect.argtypes = [vp, C.POINTER(Features), f32p, C.c_uint32, vp]
    L.rt_selftest_division.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rt_selftest_query_tree.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.rt_selftest_query_tree_built.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), f32p]
    _lib = L
    return L


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode(errors="replace"))


def camera_from_trackball(width, height, fov_deg=50.0, dist=3.0, look_at=(0.0, 0.0, 0.0), rot_deg=(20.0, 20.0, 0.0)) -> Camera:
    """CameraConfig defaults of the reference (src/config.h:18-23)."""
    cam = Camera()
    la = (C.c_float * 3)(*look_at)
    rot = (C.c_float * 3)(*rot_deg)
    aspect = float(np.float32(width) / np.float32(height))
    check(lib().rt_camera_from_trackball(fov_deg, dist, la, rot, aspect, C.byref(cam)))
    return cam


class Bvh:
    """Host BVH in the reference's nodes()/primitives() layout (rt_bvh_build; rt_bvh_build_sah_binning with sah_binning;
    rt_bvh_build_device with device=True: the same arrays, spans settled on the GPU, `device_info` = its four counters)."""

    def __init__(self, prims25, sah_binning=False, device=False):
        self.input = np.ascontiguousarray(prims25, np.uint32).reshape(-1, 25)
        h = C.c_void_p()
        self.device_info = None
        if device:
            info = (C.c_uint32 * 4)()
            check(lib().rt_bvh_build_device(self.input.ctypes.data, len(self.input), 1 if sah_binning else 0, C.byref(h), info))
            self.device_info = dict(device_spans=info[0], host_spans=info[1], levels=info[2], host_primitives=info[3])
        else:
            build = lib().rt_bvh_build_sah_binning if sah_binning else lib().rt_bvh_build
            check(build(self.input.ctypes.data, len(self.input), C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:  # at interpreter exit the module may already be torn down
            _lib.rt_bvh_destroy(self.h)
            self.h = None

    @property
    def num_nodes(self):
        return lib().rt_bvh_num_nodes(self.h)

    @property
    def num_primitives(self):
        return lib().rt_bvh_num_primitives(self.h)

    @property
    def num_levels(self):
        return lib().rt_bvh_num_levels(self.h)

    @property
    def num_leaves(self):
        return lib().rt_bvh_num_leaves(self.h)

    def nodes(self):
        n = self.num_nodes
        return np.ctypeslib.as_array(C.cast(lib().rt_bvh_nodes(self.h), u32p), (n, 8)).copy()

    def primitives(self):
        n = self.num_primitives
        if n == 0:
            return np.zeros((0, 25), np.uint32)
        return np.ctypeslib.as_array(C.cast(lib().rt_bvh_primitives(self.h), u32p), (n, 25)).copy()

    def primitive_ids(self):
        n = self.num_primitives
        if n == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(C.cast(lib().rt_bvh_primitive_ids(self.h), u32p), (n,)).copy()


def _p(a, t=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(t) if t is not C.c_void_p else C.c_void_p(a.ctypes.data)


_live_scenes = weakref.WeakSet()  # every Scene alive in this process (trim_all)


def trim_all():
    """rt_scene_trim on every live Scene of this process: for hosts that keep many scenes on one GPU."""
    for s in list(_live_scenes):
        if getattr(s, "h", None):
            s.trim()


class Scene:
    """Device-resident scene (rt_scene_create) built from reference-layout arrays."""

    def __init__(self, nodes, prims, prim_ids, materials, lights, spheres=None, textures=None):
        self.nodes = np.ascontiguousarray(nodes, np.uint32).reshape(-1, 8)
        self.prims = np.ascontiguousarray(prims, np.uint32).reshape(-1, 25)
        self.prim_ids = None if prim_ids is None else np.ascontiguousarray(prim_ids, np.uint32)
        m = np.ascontiguousarray(materials, np.float32)
        m = m.reshape(len(m), -1) if m.size else np.zeros((0, 8), np.float32)
        self.materials = np.ascontiguousarray(m[:, :8])
        self.lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 22)
        self.spheres = np.ascontiguousarray(spheres if spheres is not None else np.zeros((0, 12)), np.float32).reshape(-1, 12)
        h = C.c_void_p()
        check(lib().rt_scene_create(_p(self.nodes), len(self.nodes), _p(self.prims), len(self.prims), _p(self.prim_ids, u32p),
                                    _p(self.materials, f32p), len(self.materials), _p(self.lights, f32p), len(self.lights),
                                    _p(self.spheres, f32p), len(self.spheres), C.byref(h)))
        self.h = h
        _live_scenes.add(self)
        for m, img in (textures or {}).items():
            self.set_texture(m, img)

    @classmethod
    def from_triangles(cls, prims25, materials, lights, spheres=None, textures=None, sah_binning=False, device_build=True):
        """Scene triangles in gather order -> rt_bvh_build_device (device_build=False: rt_bvh_build or rt_bvh_build_sah_binning on
        the host; the same arrays) -> rt_scene_create."""
        bvh = Bvh(prims25, sah_binning, device=device_build)
        s = cls(bvh.nodes(), bvh.primitives(), bvh.primitive_ids(), materials, lights, spheres, textures)
        s.bvh = bvh
        return s

    def set_texture(self, material, image):
        """Material::kdTexture of mesh `material`: (H, W, 3) float32, rows as decoded (top row first); None removes it."""
        if image is None:
            check(lib().rt_scene_set_texture(self.h, int(material), None, 0, 0))
            return
        a = np.ascontiguousarray(image, np.float32)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("texture must be (H, W, 3)")
        check(lib().rt_scene_set_texture(self.h, int(material), _p(a, f32p), a.shape[1], a.shape[0]))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.rt_scene_destroy(self.h)
            self.h = None

    def trim(self):
        """rt_scene_trim: give back the scene's working buffers (the next render allocates them again)."""
        check(lib().rt_scene_trim(self.h))

    def render_image(self, cam, features, W, H, start_depth=0, flags=0, want_ids=True):
        """rt_render_image: the reference's renderImage, host buffers in and out."""
        rgb = np.zeros((H, W, 3), np.float32)
        pid = np.zeros((H, W), np.uint32) if want_ids else None
        pt = np.zeros((H, W), np.float32) if want_ids else None
        st = Stats()
        check(lib().rt_render_image(self.h, C.byref(cam), C.byref(features), W, H, start_depth, flags, _p(rgb, f32p), _p(pid, u32p),
                                    _p(pt, f32p), C.byref(st)))
        return dict(rgb=rgb, prim=pid, t=pt, stats=st)

    def render_images(self, cams, features, W, H, start_depth=0, flags=0):
        """rt_render_images: one host image per camera (the reference's loop over cameras), rendered several frames at a time."""
        rgb = np.zeros((len(cams), H, W, 3), np.float32)
        ptrs = (f32p * len(cams))(*[rgb[k].ctypes.data_as(f32p) for k in range(len(cams))])
        st = Stats()
        check(lib().rt_render_images(self.h, (Camera * len(cams))(*cams), len(cams), C.byref(features), W, H, start_depth, flags, ptrs, C.byref(st)))
        return dict(rgb=rgb, stats=st)

    def clone(self, device=None):
        """rt_scene_clone: the same scene once more, on GPU `device` (default: the one it is on)."""
        if device is None:
            device = self.device()
        h = C.c_void_p()
        check(lib().rt_scene_clone(self.h, device, C.byref(h)))
        other = object.__new__(Scene)
        other.h = h
        _live_scenes.add(other)
        for name in ("nodes", "prims", "prim_ids", "materials", "lights", "spheres", "bvh"):
            if hasattr(self, name):
                setattr(other, name, getattr(self, name))
        return other

    def device(self):
        d = C.c_int()
        check(lib().rt_scene_device(self.h, C.byref(d)))
        return d.value

    def render_device(self, cam, features, W, H, start_depth, d_rgb, d_prim=None, d_t=None, region=None, flags=0, stream=None):
        """rt_render on caller-owned device memory (integers = device addresses). Asynchronous."""
        check(lib().rt_render(self.h, C.byref(cam), C.byref(features), W, H, start_depth, C.byref(region) if region else None, flags,
                              d_rgb, d_prim, d_t, stream))

    def render_frames_device(self, cams, features, W, H, start_depth, d_rgb, d_prim=None, d_t=None, region=None, flags=0, stream=None):
        """rt_render_frames: one frame per camera in one launch sequence; frame f at d_rgb + f * W*H*3 floats. Asynchronous."""
        arr = (Camera * len(cams))(*cams)
        check(lib().rt_render_frames(self.h, arr, len(cams), C.byref(features), W, H, start_depth, C.byref(region) if region else None,
                                     flags, d_rgb, d_prim, d_t, stream))

    def sync(self, stream=None) -> Stats:
        st = Stats()
        check(lib().rt_sync(self.h, stream, C.byref(st)))
        return st

    def intersect(self, features, rays7):
        rays = np.ascontiguousarray(rays7, np.float32).reshape(-1, 7).copy()
        hits = np.zeros((len(rays), 12), np.uint32)
        check(lib().rt_intersect(self.h, C.byref(features), _p(rays, f32p), len(rays), _p(hits)))
        return dict(t=rays[:, 6].copy(), normal=hits[:, 0:3].view(np.float32).copy(), bary=hits[:, 3:6].view(np.float32).copy(),
                    uv=hits[:, 6:8].view(np.float32).copy(), prim=hits[:, 8].copy(), mesh=hits[:, 9].copy(),
                    hit=hits[:, 10].astype(bool))


def render_images_multi(scenes, cams, features, W, H, start_depth=0, flags=0):
    """rt_render_images_multi: the loop over cameras on several participants (one Scene object each; Scene.clone makes them)."""
    rgb = np.zeros((len(cams), H, W, 3), np.float32)
    ptrs = (f32p * len(cams))(*[rgb[k].ctypes.data_as(f32p) for k in range(len(cams))])
    handles = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
    st = Stats()
    check(lib().rt_render_images_multi(handles, len(scenes), (Camera * len(cams))(*cams), len(cams), C.byref(features), W, H, start_depth, flags,
                                       ptrs, C.byref(st)))
    return dict(rgb=rgb, stats=st)


def stripe_plan(height, participants, participant):
    """rt_stripe_plan: the (first memory row, rows) runs of one participant's stripes in a Screen-order frame."""
    n = lib().rt_stripe_plan(height, participants, participant, None, None, 0)
    if n < 0:
        raise ValueError("bad stripe plan arguments")
    first, rows = (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))()
    lib().rt_stripe_plan(height, participants, participant, first, rows, n)
    return [(first[i], rows[i]) for i in range(n)]


def bloom_device(d_rgb, d_scratch, W, H, F=15, threshold=0.5, intensity=0.8, region=None, stream=None):
    check(lib().rt_bloom(d_rgb, d_scratch, W, H, F, threshold, intensity, C.byref(region) if region else None, stream))


def bloom_static(d_rgb, W, H, F=15, threshold=0.5, intensity=0.8, stream=None):
    """rt_bloom_static: the whole frame, intermediate image in the library's static device array."""
    check(lib().rt_bloom_static(d_rgb, W, H, F, threshold, intensity, stream))
