"""Render N frames of a bench workload (no counting pass, no timing) - a target for profilers.

usage: frame.py N [workload] [overlapped|alone] [streams] [frames per launch sequence]
With frames per launch sequence K > 1 the N frames are rendered by rt_render_frames, K at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "dragon_glossy_bloom"]
W, H = wl.get("width", 1920), wl.get("height", 1080)
sc = bench.build_scene_arrays(wl["scene"])
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
flags = api.RT_RENDER_OVERLAPPED if (len(sys.argv) > 3 and sys.argv[3] == "overlapped") else 0  # launch sizes of frames that share the GPU
inflight = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # frames queued at once, each on a stream of its own
pooled = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if pooled > 1:
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    bufs = [torch.zeros((pooled, H, W, 3), dtype=torch.float32, device="cuda") for _ in range(inflight)]
    for i, first in enumerate(range(0, n, pooled)):
        k = i % inflight
        scene.render_frames_device([cam] * min(pooled, n - first), feat, W, H, wl["start_depth"], bufs[k].data_ptr(), flags=flags,
                                   stream=streams[k].cuda_stream)
        if inflight == 1:  # one sequence at a time: its status read before the next (which gives the next its launch plan)
            scene.sync(streams[k].cuda_stream)
    stream = streams[0].cuda_stream
elif inflight > 1:
    streams = [torch.cuda.Stream() for _ in range(inflight)]
    frames = [torch.zeros_like(frame) for _ in range(inflight)]
    for k in range(n):
        scene.render_device(cam, feat, W, H, wl["start_depth"], frames[k % inflight].data_ptr(), flags=flags, stream=streams[k % inflight].cuda_stream)
    stream = streams[0].cuda_stream
else:
    for _ in range(n):  # frame after frame, each one's status read before the next (which gives the next its launch plan)
        scene.render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), flags=flags, stream=stream)
        scene.sync(stream)
torch.cuda.synchronize()
print("stats", scene.sync(stream).rays if hasattr(scene.sync(stream), "rays") else "ok")
