"""One rank of the two-process rehearsal of the multi-GPU path (started by test_two_ranks_one_gpu.py through
torch.distributed.run).  Both ranks share GPU 0; the process group is gloo, so stripes travel through host
memory.  Each rank renders its interleaved 8-row stripes of the frame with a real rt_region, the stripes are
gathered with StripePlan, bloom runs on the gathered frame, and rank 0 compares the result bit for bit with
the frame one rt_render call produces.  Writes {"equal":…, "rays":…} as JSON to argv[1]."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes  # noqa: E402
from raytracer_amd.stripes import StripePlan  # noqa: E402


def main():
    out_path = sys.argv[1]
    W, H = int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    api.check(api.lib().rt_set_device(0))
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    sc = scenes.merge(scenes.preset("teapot"), scenes.preset("monkey"))
    scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
    feat = api.Features(enableShading=1, enableShadows=1, enableReflections=1, enableNormalInterp=1, enableAccelStructure=1,
                        shadingModel=1, enableGlossyReflection=1, numGlossySamples=2, enableBloomEffect=1)
    cam = api.camera_from_trackball(W, H)
    scratch = torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device=dev)

    def bloom(frame):
        api.bloom_device(frame.data_ptr(), scratch.data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold,
                         feat.bloomFilterIntensity, stream=0)

    # this rank's stripes, then everyone's
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    scene.render_device(cam, feat, W, H, 3, frame.data_ptr(), region=api.Region(8, rank, world), stream=0)
    st = scene.sync(0)
    mine = frame.cpu()
    touched = (mine != 0).any(dim=2).any(dim=1)                      # memory rows this rank wrote
    rows = torch.arange(H - 1, -1, -1)                               # image row of each memory row
    foreign = touched & (((rows // 8) % world) != rank)
    plan = StripePlan(H, W, rank, world, stripe_rows=8, device="cpu")
    plan.gather(mine, dist)
    frame.copy_(mine)
    bloom(frame)
    torch.cuda.synchronize()

    rays = torch.tensor([st.primaryRays, st.shadowRays, st.secondaryRays, st.pixels], dtype=torch.int64)
    dist.all_reduce(rays)
    result = {"foreign_rows_written": int(foreign.sum())}

    # a launch sequence of three frames (rt_render_frames), three cameras: this rank's stripes of every frame, then one
    # all-to-all that leaves frame f whole on rank f % world, which blooms it; each root compares its frames with the
    # frames one rt_render call per camera produces
    cams = [api.camera_from_trackball(W, H, rot_deg=(20.0, 20.0 + 50.0 * k, 0.0)) for k in range(3)]
    seq = torch.zeros((len(cams), H, W, 3), dtype=torch.float32, device=dev)
    scene.render_frames_device(cams, feat, W, H, 3, seq.data_ptr(), region=api.Region(8, rank, world), stream=0)
    scene.sync(0)
    cpu = seq.cpu()
    roots = plan.assemble(cpu, dist)
    seq.copy_(cpu)
    seq_equal = roots == [f for f in range(len(cams)) if f % world == rank]
    for f in roots:
        bloom(seq[f])
        whole = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        scene.render_device(cams[f], feat, W, H, 3, whole.data_ptr(), stream=0)
        scene.sync(0)
        bloom(whole)
        torch.cuda.synchronize()
        seq_equal = seq_equal and bool(torch.equal(seq[f].view(torch.int32), whole.view(torch.int32))) and bool((whole > 0).sum() > 1000)
    seq_flag = torch.tensor([1 if seq_equal else 0])
    dist.all_reduce(seq_flag, op=dist.ReduceOp.MIN)
    result["sequence_frames_equal"] = bool(seq_flag.item())
    if rank == 0:
        whole = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        scene.render_device(cam, feat, W, H, 3, whole.data_ptr(), region=api.Region(8, 0, 1), stream=0)
        s1 = scene.sync(0)
        bloom(whole)
        torch.cuda.synchronize()
        a, b = frame.cpu().numpy(), whole.cpu().numpy()
        result.update(equal=bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), differing=int((a != b).sum()),
                      rays=rays.tolist(), rays_one_call=[s1.primaryRays, s1.shadowRays, s1.secondaryRays, s1.pixels],
                      lit=int((b > 0).sum()))
    flag = torch.tensor([result["foreign_rows_written"]])
    dist.all_reduce(flag)
    if rank == 0:
        result["foreign_rows_written"] = int(flag.item())
        with open(out_path, "w") as f:
            json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
