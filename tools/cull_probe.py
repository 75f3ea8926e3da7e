import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracer_amd import api
sc = bench.build_scene_arrays("dragon")
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
W, H = 1920, 1080
cam = api.camera_from_trackball(W, H)
for name, kw, sd in (("primary only", dict(enableShadows=0, enableReflections=0), 6), ("primary+shadow", dict(enableReflections=0), 6),
                     ("glossy x4", dict(enableGlossyReflection=1, numGlossySamples=4), 2)):
    fkw = dict(bench.FEATURE_BASE); fkw.update(kw)
    feat = api.Features(**fkw)
    for label, flags in (("no cull", api.RT_RENDER_NO_CULL | api.RT_RENDER_COUNT_WORK), ("cull", api.RT_RENDER_COUNT_WORK)):
        st = scene.render_image(cam, feat, W, H, start_depth=sd, flags=flags, want_ids=False)["stats"]
        print("%-15s %-8s rays %9d  node pairs/ray %6.2f  tri tests/ray %6.2f" % (name, label, st.rays, st.nodePairSteps / st.rays, st.triangleTests / st.rays))
