"""One headline frame cut into S interleaved sub-frames (rt_region), each rendered by its own scene object on its own stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wl = bench.WORKLOADS["dragon_glossy_bloom"]
W, H = 1920, 1080
sc = bench.build_scene_arrays(wl["scene"], None)
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
scenes = [api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"]) for _ in range(S)]
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(S)]
regs = [api.Region(rows, t, S) for t in range(S)]
def run(k):
    for i in range(k):
        for s in range(S):
            scenes[s].render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), region=regs[s] if S > 1 else None, stream=streams[s].cuda_stream)
        torch.cuda.synchronize()  # a frame ends before the next begins
run(2)
for s in range(S):
    scenes[s].sync(streams[s].cuda_stream)
run(1)
t0 = time.perf_counter()
run(n)
dt = time.perf_counter() - t0
print("%d sub-frames of %d-row stripes: %.2f ms per frame" % (S, rows, dt / n * 1e3))
