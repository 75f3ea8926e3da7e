"""rt_render_images from host memory to host memory: ms per frame for N cameras of a bench workload (default: the headline frame).

usage: python tools/images_probe.py N [workload]"""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_PROBE_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from raytracer_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "dragon_glossy_bloom"]
W, H = wl.get("width", 1920), wl.get("height", 1080)
sc = bench.build_scene_arrays(wl["scene"])
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
fkw = dict(bench.FEATURE_BASE); fkw.update(wl["feat"])
feat = api.Features(**fkw)
cams = [api.camera_from_trackball(W, H, rot_deg=(20.0, 20.0 + 360.0 * k / n, 0.0)) for k in range(n)]
for rep in range(4):
    t0 = time.perf_counter()
    r = scene.render_images(cams, feat, W, H, wl["start_depth"])
    dt = time.perf_counter() - t0
    print(f"{n} frames {W}x{H}: {dt / n * 1e3:.2f} ms per frame, {r['stats'].rays / dt / 1e6:.0f} Mrays/s (host memory to host memory)", flush=True)
one = scene.render_image(cams[3], feat, W, H, wl["start_depth"], want_ids=False)
print("frame 3 equals rt_render_image's:", bool(np.array_equal(one["rgb"].view(np.uint32), r["rgb"][3].view(np.uint32))))
