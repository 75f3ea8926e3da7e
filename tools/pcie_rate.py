"""Whole-frame time of rt_render_image (host buffers in and out: render + bloom + device->host copy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracer_amd import api
wl = bench.WORKLOADS["dragon_glossy_bloom"]
W, H = 1920, 1080
sc = bench.build_scene_arrays(wl["scene"])
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
for _ in range(2):
    r = scene.render_image(cam, feat, W, H, start_depth=wl["start_depth"], want_ids=False)
t0 = time.perf_counter()
n = 10
for _ in range(n):
    r = scene.render_image(cam, feat, W, H, start_depth=wl["start_depth"], want_ids=False)
dt = (time.perf_counter() - t0) / n
print("rt_render_image: %.2f ms/frame, %.1f Mrays/s (host buffers, PCIe copy included)" % (dt * 1e3, r["stats"].rays / dt / 1e6))
