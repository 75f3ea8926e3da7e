"""CPU-only checks: the oracle against the reference's measured ray counts, and the parts of
librt_amd.so that need no GPU (exports, host BVH build, camera, argument validation)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE = dict(enableShading=1, enableShadows=1, enableReflections=1, enableNormalInterp=1, enableAccelStructure=1, shadingModel=1)


# Rays per frame of the unmodified reference binary (default camera, Phong + shadows + reflections +
# normal interpolation, 1 spp; "B bounces" = renderRays started at rayDepth 6 - B), measured in the
# build container with a counter around bvh.intersect.  The glossy figure includes the reference's
# unused debug query (src/extra.cpp:510-511), so the oracle is run with that query switched on.
KNOWN_RAY_COUNTS = [
    ("cornell", 512, 512, 4, dict(), 332046),
    ("cornell", 512, 512, 2, dict(enableGlossyReflection=1, numGlossySamples=4), 362845),
    ("monkey", 1920, 1080, 2, dict(), 2797725),
    ("teapot", 1920, 1080, 2, dict(), 2392554),
    ("teapot", 1920, 1080, 0, dict(enableReflections=0), 2215563),
]


@pytest.mark.parametrize("scene,W,H,start,extra,want", KNOWN_RAY_COUNTS)
def test_oracle_reproduces_reference_ray_counts(scene, W, H, start, extra, want):
    kw = dict(BASE)
    kw.update(extra)
    f = O.make_features(glossyDebugRay=1, **kw)
    s = O.oracle_scene(scene)
    st = s.render(O.make_camera(W, H), f, W, H, start_depth=start, want_ids=False)["stats"]
    assert st["primary"] == W * H
    assert st["primary"] + st["shadow"] + st["secondary"] + st["debug"] == want


# ---------------------------------------------------------------------------------------------
# librt_amd.so without a GPU
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def rt():
    from raytracer_amd import api
    return api


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    names = declared_functions()
    assert len(names) >= 20, names
    lib = rt.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


@pytest.mark.parametrize("name", ["triangle", "cube", "cornell", "monkey", "teapot", "spheres"])
def test_host_bvh_build_is_the_reference_tree(rt, name):
    prims, mats, lights, spheres = O.load_scene_arrays(name)
    want = O.OracleScene(prims, mats, lights, spheres).bvh()
    got = rt.Bvh(prims)
    assert got.num_levels == want["levels"] and got.num_leaves == want["leaves"]
    assert np.array_equal(got.nodes().view(np.uint32).reshape(-1, 8), want["nodes"])
    assert np.array_equal(got.primitives().view(np.uint32).reshape(-1, 25), want["prims"])
    assert np.array_equal(got.primitive_ids(), want["orig"])


def test_dragon_standin_is_stable(rt):
    from raytracer_amd import scenes
    sc = scenes.dragon_standin()
    assert len(sc["prims"]) == 87000
    assert scenes.mesh_hash(sc) == "462c18b55bb47549"
    b = rt.Bvh(sc["prims"])
    assert (b.num_nodes, b.num_levels, b.num_leaves) == (65536, 16, 32768)


@pytest.mark.parametrize("name", ["cube", "cornell", "monkey", "teapot", "knot", "dragon"])
def test_query_tree_is_a_sound_index_of_the_reference_tree(rt, name):
    """The library's own traversal tree (host build): every triangle of the reference's leaves in exactly one leaf, boxes
    nested and around their triangles, the binary and the 4-wide form holding the same leaves, and every triangle's
    record naming the box and the route of its reference leaf - what the GPU's proof of equality relies on."""
    from raytracer_amd import scenes
    if name in ("knot", "dragon"):
        sc = scenes.dragon_standin(n_u=120, n_v=24) if name == "knot" else scenes.dragon_standin()
        prims, mats = sc["prims"], sc["materials"]
    else:
        prims, mats, _, _ = O.load_scene_arrays(name)
    b = rt.Bvh(prims)
    info = (C.c_uint32 * 8)()
    rc = rt.lib().rt_selftest_query_tree(b.nodes().ctypes.data_as(C.c_void_p), b.num_nodes, b.primitives().ctypes.data_as(C.c_void_p),
                                         b.num_primitives, len(mats), info)
    assert rc == rt.RT_OK, rt.lib().rt_last_error().decode()
    usable, tris, pairs, pdepth, nodes4, depth4, rpairs, rdepth = list(info)
    print(name, list(info))
    assert usable == 1 and tris == len(prims)
    assert pairs >= tris // 4 and nodes4 < pairs and depth4 <= (pdepth + 1) // 2 + 1
    if name == "dragon":
        assert (rpairs, rdepth) == (32767, 16)


def test_query_tree_is_not_used_for_unnested_boxes(rt):
    """A caller's tree whose child boxes stick out of their parents' cannot carry the proof: the library must say so."""
    prims, mats, _, _ = O.load_scene_arrays("monkey")
    b = rt.Bvh(prims)
    nodes = b.nodes().view(np.uint32).reshape(-1, 8).copy()
    f = nodes.view(np.float32)
    inner = [i for i in range(2, len(nodes)) if not (nodes[i, 6] >> 31)]
    f[inner[3], 3:6] += 0.25  # an inner node's upper corner beyond its parent's
    info = (C.c_uint32 * 8)()
    rc = rt.lib().rt_selftest_query_tree(nodes.ctypes.data_as(C.c_void_p), len(nodes), b.primitives().ctypes.data_as(C.c_void_p),
                                         b.num_primitives, len(mats), info)
    assert rc == rt.RT_OK and info[0] == 0


def test_camera_matches_oracle(rt):
    for W, H in ((1920, 1080), (512, 512), (97, 61)):
        got = rt.camera_from_trackball(W, H)
        want = O.make_camera(W, H)
        assert np.array_equal(np.array(got.position[:], np.float32).view(np.uint32), np.array(want.position[:], np.float32).view(np.uint32))
        assert np.array_equal(np.array(got.quat[:], np.float32).view(np.uint32), np.array(want.quat[:], np.float32).view(np.uint32))
        assert np.float32(got.halfWidth) == np.float32(want.halfWidth) and np.float32(got.halfHeight) == np.float32(want.halfHeight)


def test_argument_validation_needs_no_gpu(rt):
    lib = rt.lib()
    out = C.c_void_p()
    assert lib.rt_bvh_build(None, 3, C.byref(out)) == rt.RT_ERR_INVALID
    assert b"rt_bvh_build" in lib.rt_last_error()
    f = rt.Features()
    cam = rt.camera_from_trackball(64, 64)
    assert lib.rt_render(None, C.byref(cam), C.byref(f), 64, 64, 0, None, 0, None, None, None, None) == rt.RT_ERR_INVALID
    assert lib.rt_bloom(None, None, 64, 64, 15, C.c_float(0.5), C.c_float(0.8), None, None) == rt.RT_ERR_INVALID
    assert lib.rt_bloom_scratch_bytes(64, 32) == 64 * 32 * 3 * 4
    assert lib.rt_sync(None, None, None) == rt.RT_ERR_INVALID


def test_flatten_rejects_malformed_trees(rt):
    """rt_scene_create validates the node array before it touches the device."""
    prims, mats, lights, spheres = O.load_scene_arrays("cube")
    b = rt.Bvh(prims)
    nodes = b.nodes().view(np.uint32).reshape(-1, 8).copy()
    lib = rt.lib()

    def create(n, p=b.primitives(), nm=len(mats)):
        out = C.c_void_p()
        m8 = np.ascontiguousarray(np.asarray(mats, np.float32).reshape(-1, 9)[:, :8])
        rc = lib.rt_scene_create(n.ctypes.data_as(C.c_void_p), len(n), p.ctypes.data_as(C.c_void_p), len(p), None,
                                 m8.ctypes.data_as(C.POINTER(C.c_float)), nm, None, 0, None, 0, C.byref(out))
        return rc, lib.rt_last_error().decode()

    bad = nodes.copy()
    bad[0, 6] = len(nodes) + 5  # child index out of range
    rc, msg = create(bad)
    assert rc == rt.RT_ERR_INVALID and "out of range" in msg
    inner = [i for i in range(len(nodes)) if not (nodes[i, 6] >> 31) and i != 1]
    if len(inner) > 1:
        bad = nodes.copy()
        bad[inner[1], 6] = 0  # a cycle back to the root
        rc, msg = create(bad)
        assert rc == rt.RT_ERR_INVALID and "twice" in msg
    rc, msg = create(nodes, nm=0)  # primitives refer to mesh 0 but no material is given
    assert rc == rt.RT_ERR_INVALID and "material" in msg
