"""Throughput of the headline frame rendered K frames per launch sequence (rt_render_frames) on S streams.

usage: python tools/frames_probe.py K:S[:flags] ...   e.g. 1:4 4:1 4:2 8:2   (flags: o = RT_RENDER_OVERLAPPED)
A step is one frame: render + bloom + copy to pinned host memory, as in bench.py.  Environment:
  RT_PROBE_WORLD=N       rank 0's stripes of a world of N (no exchange)      RT_PROBE_WORKLOAD=<bench workload>
  RT_PROBE_NO_COPY / RT_PROBE_NO_BLOOM / RT_PROBE_NO_SHADOWS   leave that part out
  RT_PROBE_LONG=m        m times as many sequences                            RT_PROBE_HW_QUEUES=q (default 8)
  RT_PROBE_PRIORITIES=p0,p1,..  stream k gets priority p[k % len] (-1 = high, 0 = normal)
  RT_PROBE_STAGGER=a,b,c,d  stream k first renders a sequence of that many frames (1 = k + 1): the streams run out of step
Every configuration gets a scene of its own (a scene keeps one buffer set per stream).
"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_PROBE_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import api

wl = bench.WORKLOADS[os.environ.get("RT_PROBE_WORKLOAD", "dragon_glossy_bloom")]
W, H = wl.get("width", 1920), wl.get("height", 1080)
world = int(os.environ.get("RT_PROBE_WORLD", "1"))
sc = bench.build_scene_arrays(wl["scene"], None)
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
if os.environ.get("RT_PROBE_NO_SHADOWS"):
    fkw["enableShadows"] = 0
feat = api.Features(**fkw)
cam = api.camera_from_trackball(W, H)
dev = torch.device("cuda", 0)
reg = api.Region(8, 0, world)
for spec in sys.argv[1:]:
    parts = spec.split(":")
    K, S = int(parts[0]), int(parts[1])
    fl = api.RT_RENDER_OVERLAPPED if (len(parts) > 2 and "o" in parts[2]) else 0
    scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])  # a scene keeps a buffer set per stream: start afresh
    prio = [int(x) for x in os.environ["RT_PROBE_PRIORITIES"].split(",")] if os.environ.get("RT_PROBE_PRIORITIES") else None
    streams = [torch.cuda.Stream(device=dev, priority=prio[k % len(prio)]) if prio else torch.cuda.Stream(device=dev) for k in range(S)]
    frames = [torch.zeros((K, H, W, 3), dtype=torch.float32, device=dev) for _ in range(S)]
    scratch = [torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device=dev) for _ in range(S)]
    host = [torch.empty((K, H, W, 3), dtype=torch.float32, pin_memory=True) for _ in range(S)]

    def batch(i):
        k = i % S
        st = streams[k]
        with torch.cuda.stream(st):
            scene.render_frames_device([cam] * K, feat, W, H, wl["start_depth"], frames[k].data_ptr(), region=reg, flags=fl, stream=st.cuda_stream)
            for f in range(K if not os.environ.get("RT_PROBE_NO_BLOOM") else 0):
                api.bloom_device(frames[k][f].data_ptr(), scratch[k].data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold,
                                 feat.bloomFilterIntensity, stream=st.cuda_stream)
            if not os.environ.get("RT_PROBE_NO_COPY"):
                host[k].copy_(frames[k], non_blocking=True)
    for rep in range(3):
        for i in range(S):
            batch(i)
        torch.cuda.synchronize()
        for k in range(S):
            try:
                stats = scene.sync(streams[k].cuda_stream)
            except api.RtError as e:
                print("  (set-up pass %d: %s)" % (rep, e))
    rays = (stats.primaryRays + stats.shadowRays + stats.secondaryRays) / K
    n = max(2 * S, 24 // K)
    n *= int(os.environ.get("RT_PROBE_LONG", "1"))
    extra = 0
    t0 = time.perf_counter()
    if os.environ.get("RT_PROBE_STAGGER"):  # put the streams out of step: stream k first renders a shorter sequence (counted):
        # RT_PROBE_STAGGER = its frames per stream, comma separated (1 = k + 1 frames on stream k)
        spec = os.environ["RT_PROBE_STAGGER"]
        first = [int(x) for x in spec.split(",")] if "," in spec else [k + 1 for k in range(S)]
        for k in range(S):
            m = min(K, first[k % len(first)])
            if m <= 0:
                continue
            extra += m
            with torch.cuda.stream(streams[k]):
                scene.render_frames_device([cam] * m, feat, W, H, wl["start_depth"], frames[k].data_ptr(), region=reg, flags=fl,
                                           stream=streams[k].cuda_stream)
    for i in range(n):
        batch(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (n * K + extra)
    print("%2d frames per sequence on %d stream(s)%s, world %d: %.3f ms per frame, %.0f Mrays/s per GPU" % (K, S, " (overlapped sizes)" if fl else "", world, dt * 1e3, rays / dt / 1e6), flush=True)
    del streams, frames, scratch, host, scene
    torch.cuda.empty_cache()
