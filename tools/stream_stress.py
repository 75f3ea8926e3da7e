"""Frames of the headline configuration queued on F streams at once, R passes: every frame and its ray counts must equal
the frame rendered alone.  usage: python tools/stream_stress.py [F] [R] [overlapped-flag 0/1]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raytracer_amd import api  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
flag = api.RT_RENDER_OVERLAPPED if (len(sys.argv) > 3 and sys.argv[3] == "1") else 0
W, H = 1920, 1080
wl = bench.WORKLOADS["dragon_glossy_bloom"]
sc = bench.build_scene_arrays(wl["scene"])
scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
kw = dict(bench.FEATURE_BASE)
kw.update(wl["feat"])
feat = api.Features(**kw)
cams = [api.camera_from_trackball(W, H, rot_deg=(20.0, 20.0 + 7.0 * k, 0.0)) for k in range(F)]
dev = torch.device("cuda", 0)
want, wstats = [], []
ref = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
for c in cams:
    scene.render_device(c, feat, W, H, wl["start_depth"], ref.data_ptr(), flags=flag, stream=0)
    wstats.append(scene.sync(0).rays)
    want.append(ref.cpu().numpy().copy())
streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
bufs = [torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev) for _ in range(F)]
torch.cuda.synchronize()
bad = 0
for rep in range(R):
    for c, st, b in zip(cams, streams, bufs):
        scene.render_device(c, feat, W, H, wl["start_depth"], b.data_ptr(), flags=flag, stream=st.cuda_stream)
    for k, (st, b) in enumerate(zip(streams, bufs)):
        stats = scene.sync(st.cuda_stream)
        same = np.array_equal(b.cpu().numpy().view(np.uint32), want[k].view(np.uint32))
        if not same or stats.rays != wstats[k]:
            bad += 1
            d = np.count_nonzero(b.cpu().numpy() != want[k])
            print(f"pass {rep} stream {k}: frame equal {same} ({d} values differ), rays {stats.rays} want {wstats[k]}", flush=True)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} F={F} R={R}: {bad} bad frames of {F * R}")
sys.exit(1 if bad else 0)
