"""Time of one rank's share of the headline frame for several world sizes (one GPU; no gather)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api
wl = bench.WORKLOADS["dragon_glossy_bloom"]
W, H = 1920, 1080
sc = bench.build_scene_arrays(wl["scene"], None)
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        reg = api.Region(8, rank, world)
        for _ in range(2):
            scene.render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), region=reg, stream=stream)
        scene.sync(stream)  # as a caller reading the frame's counters does: the next frames use this frame's launch plan
        t0 = time.perf_counter()
        for _ in range(5):
            scene.render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), region=reg, stream=stream)
        torch.cuda.synchronize()
        worst = max(worst, (time.perf_counter() - t0) / 5)
    print("world %d: slowest rank %.2f ms per frame" % (world, worst * 1e3))
