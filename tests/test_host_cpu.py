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


def _scene_triangles(name):
    from raytracer_amd import scenes
    if name in ("knot", "dragon"):
        sc = scenes.dragon_standin(n_u=120, n_v=24) if name == "knot" else scenes.dragon_standin()
        return sc["prims"], sc["materials"], sc["lights"], sc["spheres"]
    return O.load_scene_arrays(name)


@pytest.mark.parametrize("name", ["triangle", "cube", "cornell", "monkey", "teapot", "spheres", "knot", "dragon"])
def test_host_bvh_build_with_sah_binning_is_the_reference_tree(rt, name):
    """features.extra.enableBvhSahBinning (reference src/bvh.cpp:472-495, src/extra.cpp:582-699): the library's build (boxes
    grown from either end of the sorted span) against the oracle's (both boxes recomputed for each of the 49 splits)."""
    prims, mats, lights, spheres = _scene_triangles(name)
    want = O.OracleScene(prims, mats, lights, spheres, sah_binning=True).bvh()
    got = rt.Bvh(prims, sah_binning=True)
    assert got.num_levels == want["levels"] and got.num_leaves == want["leaves"]
    assert np.array_equal(got.nodes().view(np.uint32).reshape(-1, 8), want["nodes"])
    assert np.array_equal(got.primitives().view(np.uint32).reshape(-1, 25), want["prims"])
    assert np.array_equal(got.primitive_ids(), want["orig"])


def _area(lower, upper):  # calculateAABBSurfaceArea, src/extra.cpp:609-613, in float32
    a = np.abs(upper - lower).astype(np.float32)
    return np.float32(2) * ((a[0] * a[1] + a[0] * a[2]) + a[1] * a[2])


def _sah_split(pos, keys):
    """splitPrimitivesBySAHBin (src/extra.cpp:623-699) restated with numpy float32 on a span given as vertex positions
    (n, 3, 3) and centroid coordinates on the split axis, both sorted by that coordinate.  Returns the left size or None."""
    n = len(keys)
    bins = min(50, n)
    q = (np.float32(bins) * (keys - keys[0])) / (keys[-1] - keys[0])
    idx = np.minimum(q.astype(np.int64), bins - 1)
    start = np.searchsorted(idx, np.arange(bins), side="left")  # first primitive of each bin
    outer = _area(pos.min(axis=(0, 1)), pos.max(axis=(0, 1)))
    best, best_cost = None, np.finfo(np.float32).max
    for b in range(bins - 1):
        left = int(start[b + 1])
        la = _area(pos[:left].min(axis=(0, 1)), pos[:left].max(axis=(0, 1)))
        ra = _area(pos[left:].min(axis=(0, 1)), pos[left:].max(axis=(0, 1)))
        cost = np.float32(1.5) + (la * np.float32(left) + ra * np.float32(n - left)) / outer
        if cost < best_cost:
            best, best_cost = left, cost
    return None if best_cost >= np.float32(n) else best


@pytest.mark.parametrize("name", ["cornell", "monkey", "knot"])
def test_sah_binning_splits_follow_the_reference_formula(rt, name):
    """Every node of the library's SAH-binned tree against a numpy restatement of the split rule that shares no code with
    the library or the oracle: an inner node's left child holds exactly the `split` primitives with the smallest centroid
    coordinates on the node's longest axis, and a leaf of more than four primitives is a span the rule does not split."""
    prims, mats, lights, spheres = _scene_triangles(name)
    b = rt.Bvh(prims, sah_binning=True)
    nodes, stored, ids = b.nodes().view(np.uint32).reshape(-1, 8), b.primitives().view(np.uint32).reshape(-1, 25), b.primitive_ids()
    boxes = nodes.view(np.float32)[:, :6]
    allpos = np.ascontiguousarray(prims, np.uint32).reshape(-1, 25).view(np.float32)[:, 1:].reshape(-1, 3, 8)[:, :, :3]
    cent = ((allpos[:, 0] + allpos[:, 1]) + allpos[:, 2]) / np.float32(3)
    big_leaves = inner = 0

    def members(i):
        if nodes[i, 6] >> 31:
            off, cnt = int(nodes[i, 6] & 0x7FFFFFFF), int(nodes[i, 7])
            return list(ids[off:off + cnt])
        return members(int(nodes[i, 6])) + members(int(nodes[i, 7]))

    def check(i):
        nonlocal big_leaves, inner
        m = np.array(members(i), np.int64)
        assert np.array_equal(boxes[i, :3], allpos[m].min(axis=(0, 1))) and np.array_equal(boxes[i, 3:], allpos[m].max(axis=(0, 1)))
        if len(m) <= 4:
            assert nodes[i, 6] >> 31
            return
        ext = np.abs(boxes[i, 3:] - boxes[i, :3])
        axis = 0 if (ext[0] >= ext[1] and ext[0] >= ext[2]) else (1 if (ext[1] >= ext[0] and ext[1] >= ext[2]) else 2)
        order = np.argsort(cent[m, axis], kind="stable")
        split = _sah_split(allpos[m[order]], cent[m[order], axis])
        if nodes[i, 6] >> 31:
            assert split is None, (i, len(m), split)
            big_leaves += 1
            return
        inner += 1
        left = np.array(members(int(nodes[i, 6])), np.int64)
        assert split == len(left), (i, len(m), split, len(left))
        assert cent[left, axis].max() <= np.sort(cent[m, axis])[split]  # the left child took the smallest keys
        check(int(nodes[i, 6]))
        check(int(nodes[i, 7]))

    check(0)
    print(name, "inner nodes", inner, "leaves of more than four", big_leaves)
    assert inner > 0 and (big_leaves > 0 or name == "knot")


def test_sah_binning_reports_input_the_reference_is_undefined_on(rt):
    """Five triangles with one centroid: the reference's bin index is 0/0 converted to size_t (src/extra.cpp:590-598)."""
    prims, mats, lights, spheres = O.load_scene_arrays("triangle")
    five = np.repeat(np.ascontiguousarray(prims, np.uint32).reshape(-1, 25)[:1], 5, axis=0)
    out = C.c_void_p()
    assert rt.lib().rt_bvh_build_sah_binning(five.ctypes.data_as(C.c_void_p), 5, C.byref(out)) == rt.RT_ERR_INVALID
    assert b"undefined" in rt.lib().rt_last_error()
    with pytest.raises(RuntimeError):
        O.OracleScene(five, mats, lights, spheres, sah_binning=True)
    assert rt.Bvh(five).num_leaves == 2  # the median build takes them


def test_dragon_standin_is_stable(rt):
    from raytracer_amd import scenes
    sc = scenes.dragon_standin()
    assert len(sc["prims"]) == 87000
    assert scenes.mesh_hash(sc) == "462c18b55bb47549"
    b = rt.Bvh(sc["prims"])
    assert (b.num_nodes, b.num_levels, b.num_leaves) == (65536, 16, 32768)


@pytest.mark.parametrize("sah_binning", [False, True])
@pytest.mark.parametrize("name", ["cube", "cornell", "monkey", "teapot", "knot", "dragon"])
def test_query_tree_is_a_sound_index_of_the_reference_tree(rt, name, sah_binning):
    """The library's own traversal tree (host build): every triangle of the reference's leaves in exactly one leaf, boxes
    nested and around their triangles, the binary and the 4-wide form holding the same leaves, and every triangle's
    record naming the box and the route of its reference leaf - what the GPU's proof of equality relies on."""
    from raytracer_amd import scenes
    if name in ("knot", "dragon"):
        sc = scenes.dragon_standin(n_u=120, n_v=24) if name == "knot" else scenes.dragon_standin()
        prims, mats = sc["prims"], sc["materials"]
    else:
        prims, mats, _, _ = O.load_scene_arrays(name)
    b = rt.Bvh(prims, sah_binning)  # the caller's tree: the reference's median build, or its SAH-binned one (leaves of any size)
    info = (C.c_uint32 * 8)()
    rc = rt.lib().rt_selftest_query_tree(b.nodes().ctypes.data_as(C.c_void_p), b.num_nodes, b.primitives().ctypes.data_as(C.c_void_p),
                                         b.num_primitives, len(mats), info)
    assert rc == rt.RT_OK, rt.lib().rt_last_error().decode()
    usable, tris, pairs, pdepth, nodes4, depth4, rpairs, rdepth = list(info)
    print(name, list(info))
    if name == "cube" and sah_binning:  # the binned build leaves the cube's 12 triangles in the root: nothing to index
        assert b.num_nodes == 2 and usable == 0
        return
    assert usable == 1 and tris == len(prims)
    assert pairs >= tris // 4 and nodes4 < pairs and depth4 <= (pdepth + 1) // 2 + 1
    if name == "dragon" and not sah_binning:
        assert (rpairs, rdepth) == (32767, 16)


def test_caller_trees_with_leaves_of_hundreds_of_triangles_are_accepted(rt):
    """The SAH-binned build leaves a span it cannot split cheaply whole; on overlapping triangles that is most of the scene in one
    leaf.  The flattened tree and the library's index of it must hold up (a leaf reference's count field ends at 63)."""
    import fuzz_parity as F
    sc = F.soup(0)
    b = rt.Bvh(sc["prims"], sah_binning=True)
    nodes = b.nodes()
    assert int(nodes[(nodes[:, 6] >> 31) == 1][:, 7].max()) > 63
    info = (C.c_uint32 * 8)()
    rc = rt.lib().rt_selftest_query_tree(nodes.ctypes.data_as(C.c_void_p), b.num_nodes, b.primitives().ctypes.data_as(C.c_void_p),
                                         b.num_primitives, len(sc["materials"]), info)
    assert rc == rt.RT_OK, rt.lib().rt_last_error().decode()
    assert info[0] == 1 and info[1] == len(sc["prims"])


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


def test_device_tree_build_without_a_gpu_says_so(rt):
    """rt_bvh_build_device never falls back to the host build by itself: without a usable GPU it returns RT_ERR_HIP and names
    rt_bvh_build; spans of at most four primitives need no device (the root is a leaf)."""
    lib = rt.lib()
    count = C.c_int(0)
    if lib.rt_device_count(C.byref(count)) == rt.RT_OK and count.value > 0:
        pytest.skip("a GPU is present: tests/test_device_tree_build.py covers the device build")
    prims, _, _, _ = O.load_scene_arrays("cornell")
    with pytest.raises(rt.RtError) as e:
        rt.Bvh(prims, device=True)
    assert e.value.code == rt.RT_ERR_HIP and "rt_bvh_build" in str(e.value)
    small = rt.Bvh(prims[:3], device=True)
    host = rt.Bvh(prims[:3])
    assert np.array_equal(small.nodes(), host.nodes()) and np.array_equal(small.primitives(), host.primitives())
