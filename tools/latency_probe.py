"""Tiny frames (64 camera rays, no shadows/reflections): duration of the trace launch vs steps per ray."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api
sc = bench.build_scene_arrays("dragon")
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
feat = api.Features(enableShading=1, enableShadows=0, enableReflections=0, enableNormalInterp=1, enableAccelStructure=1, shadingModel=1)
for W, H, dist in ((8, 8, 3.0), (8, 8, 1.2), (64, 64, 1.2), (512, 512, 1.2)):
    cam = api.camera_from_trackball(W, H, dist=dist)
    for flags in (api.RT_RENDER_COUNT_WORK | api.RT_RENDER_NO_CULL, api.RT_RENDER_COUNT_WORK, 0, 0):
        r = scene.render_image(cam, feat, W, H, start_depth=6, flags=flags, want_ids=False)
    st = scene.render_image(cam, feat, W, H, start_depth=6, flags=api.RT_RENDER_COUNT_WORK, want_ids=False)["stats"]
    print("probe %dx%d dist %.1f: rays %d node pairs %d tri tests %d (culled walk)" % (W, H, dist, st.rays, st.nodePairSteps, st.triangleTests))
