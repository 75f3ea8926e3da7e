"""Probe: one rt_render + rt_bloom captured in a HIP graph (through torch) against plain stream launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "dragon_glossy_bloom"]
W, H = wl.get("width", 1920), wl.get("height", 1080)
sc = bench.build_scene_arrays(wl["scene"], None)
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
scratch = torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device="cuda")

def step(stream):
    scene.render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), stream=stream)
    if feat.enableBloomEffect:
        api.bloom_device(frame.data_ptr(), scratch.data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold, feat.bloomFilterIntensity, stream=stream)

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        step(s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(s.cuda_stream)
    torch.cuda.synchronize()
    print("stream launches: %.3f ms/frame" % ((time.perf_counter() - t0) / 10 * 1e3))
    ref = frame.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step(s.cuda_stream)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("graph replay:    %.3f ms/frame; frames equal: %s" % ((time.perf_counter() - t0) / 10 * 1e3, bool(torch.equal(ref, frame))))
