"""Render frames of one rank's share of the headline frame (a target for profilers): world rank frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api
world, rank, n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 0), (3, 4)))
wl = bench.WORKLOADS["dragon_glossy_bloom"]
W, H = 1920, 1080
sc = bench.build_scene_arrays(wl["scene"], None)
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
reg = api.Region(8, rank, world)
for _ in range(n):
    scene.render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), region=reg, stream=stream)
torch.cuda.synchronize()
print("rays", scene.sync(stream).rays)
