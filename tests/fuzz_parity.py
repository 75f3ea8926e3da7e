"""Randomised differential test of the HIP path against the CPU oracle: scenes, trees, cameras (also inside the model),
feature combinations, frame sizes, start depths and flags drawn from a seed; every case is held to the bar of
test_gpu_parity.py.  Used by test_gpu_fuzz.py (a fixed set of seeds) and from the command line for longer runs:

    python tests/fuzz_parity.py <first seed> <count> [big|soup]   ->  one line per case, failures at the end (exit status 1)

`big`: frames of 1280x720 and 1920x1080 on the large scenes, up to four per call - the sizes at which the library changes its
schedule (one ray per lane, hand-over, large-sequence settings).  `soup`: a generated scene per seed, triangles on a coarse grid
with duplicates - exact ties of distance, rays through shared edges and vertices.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle as O  # noqa: E402
from raytracer_amd import api, scenes  # noqa: E402

RMSE_TOL = 1e-3
SCENES = ["cornell", "cornell_plight", "cube", "monkey", "teapot", "spheres", "knot", "teapot+monkey", "monkey+spheres"]
_cache = {}


def soup(seed):
    """A scene made to produce exact ties: triangles with vertices on a coarse grid (shared edges and vertices, coplanar and
    crossing faces), some of them stored twice under different meshes (two hits at one distance: the reference reports the one
    its walk tests last), lights and look-at points on the same grid, a sphere or two."""
    r = np.random.default_rng(1_000_003 * seed + 17)
    n = int(r.choice([1, 2, 5, 12, 40, 150, 400]))
    step = float(r.choice([0.125, 0.25, 0.5]))
    cells = int(round(1.0 / step))
    pos = np.zeros((0, 3, 3), np.float32)
    while len(pos) < n:
        v = (r.integers(-cells, cells + 1, size=(2 * n + 8, 3, 3)) * step).astype(np.float32)
        e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
        area2 = np.cross(e1, e2)
        keep = (np.abs(area2).sum(axis=1) > 0) & (np.abs(e1).max(axis=1) <= 4 * step) & (np.abs(e2).max(axis=1) <= 4 * step)
        pos = np.concatenate([pos, v[keep]])[:n]
    dup = r.random(n) < 0.15
    pos = np.concatenate([pos, pos[dup]])
    if r.random() < 0.3:  # a few triangles without area (two corners in one place, or three on a line): they can never be hit
        flat = r.random(len(pos)) < 0.1
        pos[flat, 2] = np.where(r.random((int(flat.sum()), 1)) < 0.5, pos[flat, 1], 2 * pos[flat, 1] - pos[flat, 0])
    n_all = len(pos)
    n_mesh = int(r.integers(1, 5))
    ids = r.integers(0, n_mesh, n_all).astype(np.uint32)
    geo = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0]).astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        geo = geo / np.sqrt((geo * geo).sum(axis=1, dtype=np.float32))[:, None]
    geo[~np.isfinite(geo).all(axis=1)] = (0.0, 1.0, 0.0)  # (vertex normals of the triangles without area)
    nrm = np.repeat(geo[:, None, :], 3, axis=1).astype(np.float32)
    bend = r.random(n_all) < 0.5  # vertex normals that differ from the face's
    nrm[bend] = nrm[bend] + r.uniform(-0.3, 0.3, (int(bend.sum()), 3, 3)).astype(np.float32)
    uv = r.uniform(-0.5, 1.5, (n_all, 3, 2)).astype(np.float32)  # also outside the image
    mats = np.zeros((n_mesh, 9), np.float32)
    mats[:, 0:3] = r.uniform(0.1, 1.0, (n_mesh, 3))
    mats[:, 3:6] = r.uniform(0.0, 0.9, (n_mesh, 3)) * (r.random((n_mesh, 1)) < 0.7)
    mats[:, 6] = r.choice([1.0, 5.0, 20.0, 80.0], n_mesh)
    mats[:, 7] = 1.0
    def light():
        # near grid points, not on them: a light in the plane of a triangle makes every point of it a coin toss between lit and unlit
        # (dot(normal, to the light) is 0 up to rounding), which a last-ulp difference in a glossy ray's hit point then flips
        at = lambda: (r.integers(-2 * cells, 2 * cells + 1, 3) * step + r.uniform(0.01, 0.04, 3) * r.choice([-1.0, 1.0], 3)).astype(np.float32)
        kind = int(r.choice([0, 0, 1, 2]))
        rec = np.zeros(22, np.float32)
        rec[0] = kind
        if kind == 0:    # point: position, colour
            rec[1:4], rec[4:7] = at(), r.uniform(0.3, 1.0, 3)
        elif kind == 1:  # segment: two end points, their colours (src/common.h:50-53)
            rec[1:4], rec[4:7], rec[7:10], rec[10:13] = at(), at(), r.uniform(0.2, 1.0, 3), r.uniform(0.2, 1.0, 3)
        else:            # parallelogram: corner, two edges, four colours (src/common.h:55-60)
            rec[1:4], rec[4:7], rec[7:10] = at(), r.integers(-2, 3, 3) * step, r.integers(-2, 3, 3) * step
            rec[10:22] = r.uniform(0.0, 1.0, 12)
        return rec
    lights = np.stack([light() for _ in range(int(r.integers(1, 3)))])
    spheres = np.zeros((int(r.integers(0, 3)), 12), np.float32)
    for sp in spheres:
        sp[0:3] = r.integers(-cells, cells + 1, 3) * step
        sp[3] = float(r.choice([step, 2 * step, 0.3]))
        sp[4:7], sp[7:10], sp[10], sp[11] = r.uniform(0.1, 1, 3), r.uniform(0, 0.8, 3), 10.0, 1.0
    textures = {}
    for m in range(n_mesh):
        if r.random() < 0.5:  # Material::kdTexture: a small random image, odd sizes included
            mats[m, 8] = 1.0
            textures[m] = r.uniform(0.0, 1.0, (int(r.choice([1, 3, 8])), int(r.choice([2, 5, 8])), 3)).astype(np.float32)
    return dict(prims=scenes.pack_prims(ids, pos, nrm, uv), materials=mats, lights=lights, spheres=spheres, textures=textures)


def scene_arrays(name):
    if name.startswith("soup:"):
        return soup(int(name[5:]))
    if name == "knot":
        return scenes.dragon_standin(n_u=120, n_v=24)
    if name == "dragon":
        return scenes.dragon_standin()
    if "+" in name:
        return scenes.merge(*[scenes.preset(n) for n in name.split("+")])
    return scenes.preset(name)


def scene_pair(name, sah, transparent):
    key = (name, sah, transparent)
    if key not in _cache:
        if len(_cache) > (2 if name in ("dragon", "knot", "teapot+monkey") else 12):  # (a scene keeps its working buffers: gigabytes per stream at 1080p)
            _cache.clear()
        sc = scene_arrays(name)
        mats = sc["materials"].copy()
        if transparent:
            mats[::2, 7] = 0.6  # every other mesh lets 60 % through
            mats[1::3, 7] = 0.25
        try:
            o = O.OracleScene(sc["prims"], mats, sc["lights"], sc["spheres"], sc.get("textures"), sah_binning=sah)
        except RuntimeError:  # input the reference's SAH-binned build is undefined on (centroids that coincide): the median build
            sah = False
            o = O.OracleScene(sc["prims"], mats, sc["lights"], sc["spheres"], sc.get("textures"))
        g = api.Scene.from_triangles(sc["prims"], mats, sc["lights"], sc["spheres"], sc.get("textures"), sah_binning=sah)
        _cache[key] = (g, o)
    return _cache[key]


def draw_case(seed, big=False, soups=False):
    r = np.random.default_rng(seed)
    c = dict(seed=seed)
    c["scene"] = SCENES[r.integers(len(SCENES))]
    if soups:
        c["scene"] = f"soup:{seed}"
    c["sah"] = bool(r.random() < 0.3) and c["scene"] != "spheres"
    c["W"], c["H"] = int(r.integers(48, 200)), int(r.integers(32, 140))
    if big:
        c["scene"] = str(r.choice(["dragon", "dragon", "knot", "teapot+monkey"]))
        c["W"], c["H"] = (1920, 1080) if r.random() < 0.6 else (1280, 720)
    c["cam"] = dict(fov_deg=float(r.choice([25.0, 50.0, 80.0, 110.0])), dist=float(r.choice([0.05, 0.4, 1.2, 3.0, 6.0])),
                    look_at=tuple(float(x) for x in r.uniform(-0.4, 0.4, 3)) if r.random() < 0.5 else (0.0, 0.0, 0.0),
                    rot_deg=(float(r.uniform(-89, 89)), float(r.uniform(-180, 180)), 0.0))
    f = dict(enableShading=int(r.random() < 0.9), shadingModel=int(r.integers(4)), enableShadows=int(r.random() < 0.7),
             enableReflections=int(r.random() < 0.75), enableNormalInterp=int(r.random() < 0.7), enableAccelStructure=1,
             enableTextureMapping=int(r.random() < 0.3), enableBilinearTextureFiltering=int(r.random() < 0.5),
             numShadowSamples=int(r.integers(1, 6)))
    if r.random() < 0.1 and c["scene"] in ("cube", "cornell", "cornell_plight", "spheres"):
        f["enableAccelStructure"] = 0  # every triangle for every ray: small scenes only
    if r.random() < 0.5:
        f["enableGlossyReflection"], f["numGlossySamples"] = 1, int(r.choice([1, 2, 3, 4, 5, 6]))
    if r.random() < 0.15:
        f["numPixelSamples"], f["enableJitteredSampling"] = int(r.choice([2, 4, 5])), int(r.random() < 0.5)
    c["transparent"] = bool(r.random() < 0.2)
    if c["transparent"]:  # (the reference's walk of a light sample through transparent occluders need not end: shadows off)
        f["enableTransparency"], f["enableShadows"] = 1, 0
    if r.random() < 0.3:
        f["enableBloomEffect"], f["bloomFilterSize"] = 1, int(r.choice([1, 2, 7, 15, 30]))
        f["bloomFilterThreshold"], f["bloomFilterIntensity"] = float(r.choice([0.2, 0.5, 0.9])), float(r.choice([0.3, 0.8, 1.5]))
    c["features"] = f
    c["start_depth"] = int(r.choice([0, 2, 3, 4, 5, 6]))
    if f.get("numGlossySamples", 1) >= 4 and c["start_depth"] < 2:
        c["start_depth"] = 2  # keeps the oracle's frame within seconds
    c["flags"] = int(r.choice([0, 0, 0, api.RT_RENDER_NO_CULL, api.RT_RENDER_REFERENCE_TREE, api.RT_RENDER_SERIAL,
                               api.RT_RENDER_NO_CULL | api.RT_RENDER_REFERENCE_TREE]))
    c["frames"] = int(r.choice([1, 1, 1, 2, 5]))  # more than one: rt_render_images, one camera per frame
    if soups:  # views along the grid's axes and diagonals, from grid points: rays through shared edges and vertices
        c["frames"] = 1
        if c["transparent"] and r.random() < 0.5:  # light through transparent surfaces, the walk bounded at 64 surfaces on both sides
            f["enableShadows"], c["walk_cap"] = 1, 64
            if f.get("numGlossySamples", 1) > 2:  # (up to 64 queries per light sample and hit: keep the oracle's frame within seconds)
                f["numGlossySamples"] = 2
            f.pop("numPixelSamples", None), f.pop("enableJitteredSampling", None)
        c["cam"]["rot_deg"] = (float(r.choice([0.0, 45.0, 90.0, -90.0, 30.0])), float(r.choice([0.0, 45.0, 90.0, 180.0, -135.0])), 0.0)
        c["cam"]["look_at"] = tuple(float(x) for x in r.integers(-2, 3, 3) * 0.25)
        c["cam"]["dist"] = float(r.choice([0.5, 1.0, 2.0, 4.0]))
        c["W"], c["H"] = int(r.choice([32, 64, 65, 128])), int(r.choice([32, 64, 63, 96]))
    if big:
        c["frames"] = int(r.choice([1, 1, 2, 4]))
        c["cam"]["dist"] = float(r.choice([0.4, 1.2, 3.0]))
        f.pop("numPixelSamples", None), f.pop("enableJitteredSampling", None)
        if f.get("numGlossySamples", 1) > 4:
            f["numGlossySamples"] = 4
        if c["flags"] & api.RT_RENDER_SERIAL:
            c["flags"] = 0
        # a camera inside the model makes every ray hit: keep a pixel's recursion tree within a few dozen rays, the oracle's frame within a minute
        n = f.get("numGlossySamples", 1) if f.get("enableGlossyReflection") else 1
        c["start_depth"] = max(c["start_depth"], {2: 2, 3: 3, 4: 4}.get(n, 0))
        if n >= 3:
            c["frames"] = min(c["frames"], 2)
    return c


def run_case(c):
    """Returns a list of complaints (empty: the case holds)."""
    g, o = scene_pair(c["scene"], c["sah"], c["transparent"])
    W, H, f = c["W"], c["H"], c["features"]
    fg, fo = api.Features(**f), O.make_features(**f)
    bad = []
    cams = []
    for k in range(c["frames"]):
        kw = dict(c["cam"])
        kw["rot_deg"] = (kw["rot_deg"][0], kw["rot_deg"][1] + 37.0 * k, 0.0)
        cams.append((api.camera_from_trackball(W, H, **kw), O.make_camera(W, H, **kw)))

    def want_of(co):
        O.lib().orc_set_shadow_walk_cap(c.get("walk_cap", 0))
        try:
            ro = o.render(co, fo, W, H, start_depth=c["start_depth"])
        finally:
            O.lib().orc_set_shadow_walk_cap(0)
        rgb = ro["rgb"]
        if f.get("enableBloomEffect"):
            rgb = O.bloom(rgb, f["bloomFilterSize"], f["bloomFilterThreshold"], f["bloomFilterIntensity"])
        return ro, rgb

    def rmse_of(a, b):
        d = a.astype(np.float64) - b.astype(np.float64)
        return np.sqrt((d * d).mean(axis=(0, 1)))

    if c["frames"] == 1:
        n = W * H
        rg = g.render_image(cams[0][0], fg, W, H, start_depth=c["start_depth"], flags=c["flags"])
        ro, want = want_of(cams[0][1])
        id_bad = int(np.count_nonzero(rg["prim"] != ro["prim"]))
        t_bad = int(np.count_nonzero(rg["t"].view(np.uint32) != ro["t"].view(np.uint32)))
        if id_bad > 1e-4 * n or t_bad > 1e-4 * n or ((id_bad or t_bad) and n < 500000):
            bad.append(f"primary hits differ: {id_bad} ids, {t_bad} distances of {n}")
        elif id_bad or t_bad:  # within the bar of 99.99 %: reported, not failed
            print(f"note seed {c['seed']}: {id_bad} primary ids, {t_bad} distances of {n} differ", flush=True)
        got = (rg["stats"].primaryRays, rg["stats"].shadowRays, rg["stats"].secondaryRays)
        exp = (ro["stats"]["primary"], ro["stats"]["shadow"], ro["stats"]["secondary"])
        # (a walk through transparent occluders ends when it meets the shaded surface within 1e-5, src/light.cpp:107-136, and otherwise goes on
        # to the bound: at grazing incidence that is decided by rounding, and each flip is worth up to 64 queries - 1e-3 on their count)
        # Everything else that decides a ray count is integer work or correctly rounded arithmetic (the glossy sample's sin / cos are
        # evaluated as the host libm evaluates them): the counts are equal.
        slack = (0.0, 1e-3 if c.get("walk_cap") else 0.0, 0.0)
        if any(abs(a - b) > k * b for a, b, k in zip(got, exp, slack)):
            bad.append(f"ray counts {got} against {exp}")
        if rg["stats"].stackOverflows:
            bad.append("stack overflows")
        e = rmse_of(rg["rgb"], want)
        if not (e <= RMSE_TOL).all() or not np.isfinite(rg["rgb"]).all():
            bad.append(f"rgb rmse {e}, max abs {np.abs(rg['rgb'] - want).max():.3g}")
    else:
        rs = g.render_images([cg for cg, _ in cams], fg, W, H, start_depth=c["start_depth"], flags=c["flags"])
        for k, (_, co) in enumerate(cams):
            _, want = want_of(co)
            e = rmse_of(rs["rgb"][k], want)
            if not (e <= RMSE_TOL).all() or not np.isfinite(rs["rgb"][k]).all():
                bad.append(f"frame {k}: rgb rmse {e}, max abs {np.abs(rs['rgb'][k] - want).max():.3g}")
    return bad


def describe(c):
    f = c["features"]
    on = [k.replace("enable", "") for k, v in f.items() if k.startswith("enable") and v]
    return (f"seed {c['seed']}: {c['scene']}{' sah' if c['sah'] else ''}{' transparent' if c['transparent'] else ''} {c['W']}x{c['H']} x{c['frames']} depth "
            f"{c['start_depth']} flags {c['flags']} fov {c['cam']['fov_deg']} dist {c['cam']['dist']} model {f['shadingModel']} glossy "
            f"{f.get('numGlossySamples', 0) if f.get('enableGlossyReflection') else 0} spp {f.get('numPixelSamples', 1)} [{' '.join(on)}]")


def main():
    """usage: fuzz_parity.py FIRST COUNT [big|soup]   or   fuzz_parity.py seeds [plain|big|soup]:S1,S2,... ..."""
    cases = []
    if sys.argv[1] == "seeds":
        for group in sys.argv[2:]:
            mode, _, seeds = group.partition(":")
            cases += [(int(x), mode == "big", mode == "soup") for x in seeds.split(",") if x]
    else:
        first, count = int(sys.argv[1]), int(sys.argv[2])
        mode = sys.argv[3] if len(sys.argv) > 3 else ""
        cases = [(seed, mode == "big", mode == "soup") for seed in range(first, first + count)]
    count = len(cases)
    failures = []
    for seed, big, soups in cases:
        c = draw_case(seed, big, soups)
        if big or c.get("walk_cap"):
            print("     " + describe(c), flush=True)  # (the oracle's frame can take a minute)
        try:
            bad = run_case(c)
        except Exception as e:  # an error return is a finding too
            bad = [f"{type(e).__name__}: {e}"]
        print(("FAIL " if bad else "ok   ") + describe(c) + ("  -> " + "; ".join(bad) if bad else ""), flush=True)
        if bad:
            failures.append(seed)
    print(f"{count - len(failures)} of {count} cases hold; failing seeds: {failures}", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
