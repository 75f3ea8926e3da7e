"""One rank's share of the headline frame with frames in flight, for several world sizes (one GPU; no gather).

usage: python tools/region_flight_probe.py [frames_in_flight ...]   (default 4 8)
Prints ms per frame of rank 0's stripes (render + bloom + copy to pinned host memory, as a bench step without the
gather), the host time spent queueing one frame, and what the world would deliver if every rank took that long.
"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_PROBE_HW_QUEUES", "8")  # the box exports 4
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api

flights = [int(a) for a in sys.argv[1:]] or [4, 8]
wl = bench.WORKLOADS["dragon_glossy_bloom"]
W, H = 1920, 1080
sc = bench.build_scene_arrays(wl["scene"], None)
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
dev = torch.device("cuda", 0)
FMAX = max(flights)
streams = [torch.cuda.Stream(device=dev) for _ in range(FMAX)]
frames = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in range(FMAX)]
scratch = [torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device=dev) for _ in range(FMAX)]
host = [torch.empty((H, W, 3), dtype=torch.float32, pin_memory=True) for _ in range(FMAX)]
full_rays = None
for world in (1, 2, 4, 8):
    reg = api.Region(8, 0, world)
    for F in flights:
        fl = api.RT_RENDER_OVERLAPPED if F > 1 else 0

        def step(i):
            k = i % F
            st = streams[k]
            with torch.cuda.stream(st):
                scene.render_device(cam, feat, W, H, wl["start_depth"], frames[k].data_ptr(), region=reg, flags=fl, stream=st.cuda_stream)
                api.bloom_device(frames[k].data_ptr(), scratch[k].data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold,
                                 feat.bloomFilterIntensity, stream=st.cuda_stream)
                host[k].copy_(frames[k], non_blocking=True)
        for i in range(2 * F):
            step(i)
        torch.cuda.synchronize()
        stats = scene.sync(streams[0].cuda_stream)
        rays = stats.primaryRays + stats.shadowRays + stats.secondaryRays
        if world == 1:
            full_rays = rays
        for i in range(F):
            step(i)
        torch.cuda.synchronize()
        n = 24
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("world %d, %d frames in flight: rank 0 %.3f ms per frame (host queues a frame in %.3f ms), %d rays -> %.0f Mrays/s for the world"
              % (world, F, dt * 1e3, t_host / n * 1e3, rays, full_rays / dt / 1e6), flush=True)
