#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one frame: rt_render (+ RCCL all-gather of the row stripes when N > 1) + rt_bloom, with the
scene and the framebuffer resident in HBM.  The frame is cut into interleaved 8-row stripes, stripe s
belongs to rank s % N; every rank ends each step holding the full post-bloom frame.  Rank 0 prints one
JSON line.  `value` counts every closest-hit query that contributes to the image (primary + shadow +
reflection rays) over all ranks, divided by the slowest rank's time for the K steps.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

FEATURE_BASE = dict(enableShading=1, enableShadows=1, enableReflections=1, enableNormalInterp=1, enableAccelStructure=1,
                    shadingModel=1)
# name -> (scene builder name, features, start depth, bloom, description)
WORKLOADS = {
    "dragon_glossy_bloom": dict(scene="dragon", feat=dict(enableGlossyReflection=1, numGlossySamples=4, enableBloomEffect=1),
                                start_depth=2, desc="dragon stand-in (87,000 tris) 1080p, Phong + shadows, glossy x4, 4 bounces, bloom F=15"),
    "dragon_primary_shadow": dict(scene="dragon", feat=dict(enableReflections=0), start_depth=0,
                                  desc="dragon stand-in 1080p, primary + shadow rays"),
    "teapot_monkey_phong": dict(scene="teapot+monkey", feat=dict(), start_depth=2,
                                desc="teapot + monkey 1080p, Phong + shadows + mirror reflections, 4 bounces"),
    "cornell_mirror": dict(scene="cornell", feat=dict(), start_depth=4, desc="CornellBox-Mirror-Rotated, 2 bounces"),
    # one step = one frame of an orbit: camera rotation.y = 20 + 360 * k / steps degrees (other fields default)
    "dragon_4k_orbit": dict(scene="dragon", feat=dict(enableGlossyReflection=1, numGlossySamples=4, enableBloomEffect=1),
                            start_depth=2, width=3840, height=2160, orbit=True,
                            desc="dragon stand-in 3840x2160 camera sweep, Phong + shadows, glossy x4, 4 bounces, bloom F=15"),
}


def build_scene_arrays(name, obj=None):
    from raytracer_amd import scenes
    if obj and name == "dragon":  # the real data/dragon.obj (or any other mesh) in place of the stand-in
        return scenes.load_obj(obj)
    if name == "dragon":
        return scenes.dragon_standin()
    if name == "teapot+monkey":
        return scenes.merge(scenes.preset("teapot"), scenes.preset("monkey"))
    return scenes.preset(name)


def cpu_baseline(sc, feat_kw, start_depth, W, H, budget_s):
    """Times the CPU oracle (a port of the reference algorithm incl. its unused glossy debug query) on a
    bounded, evenly spread sample of rows of the same frame, on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    kw = dict(FEATURE_BASE)
    kw.update({k: v for k, v in feat_kw.items() if k != "enableBloomEffect"})
    f = O.make_features(glossyDebugRay=1, **kw)
    cam = O.make_camera(W, H)
    s = O.OracleScene(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    t0 = time.perf_counter()
    probe = s.render(cam, f, W, H, start_depth=start_depth, want_ids=False, threads=threads, row_step=32)
    tp = time.perf_counter() - t0
    step = 1
    while tp * 32 / step > budget_s and step < 32:
        step *= 2
    # a sample of about budget_s seconds: every step-th row of the frame, repeated while time remains
    reps, dt, r = 0, 0.0, probe
    while reps == 0 or (dt + dt / reps <= budget_s and reps < 64):
        t0 = time.perf_counter()
        r = s.render(cam, f, W, H, start_depth=start_depth, want_ids=False, threads=threads, row_step=step)
        dt += time.perf_counter() - t0
        reps += 1
    st = r["stats"]
    rays = st["primary"] + st["shadow"] + st["secondary"]
    return dict(value=rays * reps / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample=f"every {max(step, 1)}th row of the {W}x{H} frame, {reps} time(s) ({rays} rays per pass, {dt:.2f} s in all; "
                       f"{st['debug']} extra debug queries per pass executed but not counted)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="dragon_glossy_bloom", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0, help="default: the workload's (1920 unless it says otherwise)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--obj", default=None, help="Wavefront OBJ to use where a workload names the dragon")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cull", action="store_true", help="traverse exactly as the reference (no distance culling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    api.check(api.lib().rt_set_device(local_rank))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    wl = WORKLOADS[args.workload]
    W, H = args.width or wl.get("width", 1920), args.height or wl.get("height", 1080)
    if H % 8:
        raise SystemExit("height must be a multiple of the 8-row stripe")
    sc = build_scene_arrays(wl["scene"], args.obj)
    from raytracer_amd import scenes as _scenes
    scene = api.Scene.from_triangles(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])
    fkw = dict(FEATURE_BASE)
    fkw.update(wl["feat"])
    feat = api.Features(**fkw)
    cam = api.camera_from_trackball(W, H)
    cams = [api.camera_from_trackball(W, H, rot_deg=(20.0, 20.0 + 360.0 * k / max(args.steps, 1), 0.0)) for k in range(args.steps)] \
        if wl.get("orbit") else None
    region = api.Region(8, rank, world)
    flags = api.RT_RENDER_NO_CULL if args.no_cull else 0
    stream = torch.cuda.current_stream().cuda_stream

    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    scratch = torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device=dev)
    from raytracer_amd.stripes import StripePlan
    plan = StripePlan(H, W, rank, world, stripe_rows=8, device=dev)

    ev_r0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_r1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev_r0[i].record()
        scene.render_device(cams[i] if (cams and i is not None) else cam, feat, W, H, wl["start_depth"], frame.data_ptr(), region=region,
                            flags=flags, stream=stream)
        if i is not None:
            ev_r1[i].record()
        plan.gather(frame, dist)
        if feat.enableBloomEffect:
            api.bloom_device(frame.data_ptr(), scratch.data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold,
                             feat.bloomFilterIntensity, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    stats = scene.sync(stream)  # ray counters of this rank's share of one frame
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0

    if cams:  # every frame of the sweep has its own ray count: recount them outside the timed region
        tot = [0, 0, 0, 0]
        for c in cams:
            scene.render_device(c, feat, W, H, wl["start_depth"], frame.data_ptr(), region=region, flags=flags, stream=stream)
            st = scene.sync(stream)
            for k, v in enumerate((st.primaryRays, st.shadowRays, st.secondaryRays, st.pixels)):
                tot[k] += v / len(cams)
        rays_rank = torch.tensor(tot, dtype=torch.float64, device=dev)
    else:
        rays_rank = torch.tensor([stats.primaryRays, stats.shadowRays, stats.secondaryRays, stats.pixels], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(rays_rank)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    rays_frame = float(rays_rank[:3].sum().item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_r0, ev_r1)]))

    # Algorithmic bytes of rank 0's render launch (one rt_render = a fixed sequence of kernels on one stream,
    # dominated by the trace kernels): per ray 64 B per node-pair step + 36 B per triangle test of the
    # REFERENCE traversal (no culling) + 28 B ray in + 16 B hit out, plus 12 B per pixel written.  The
    # duration is measured with events on the stream the launch runs on.
    scene.render_device(cam, feat, W, H, wl["start_depth"], frame.data_ptr(), region=region,
                        flags=api.RT_RENDER_NO_CULL | api.RT_RENDER_COUNT_WORK, stream=stream)
    work = scene.sync(stream)
    alg_bytes = 64.0 * work.nodePairSteps + 36.0 * work.triangleTests + 44.0 * work.rays + 12.0 * work.pixels
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload if world == 1 else "", None)
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        out = {
            "metric": "Mrays/s",
            "value": rays_frame * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "description": wl["desc"], "width": W, "height": H,
                       "triangles": int(len(sc["prims"])), "mesh_hash": _scenes.mesh_hash(sc), "mesh": args.obj or wl["scene"],
                       "rays_per_frame": int(rays_frame),
                       "partition": f"interleaved 8-row stripes over {world} GPU(s)" + (", RCCL all-gather" if world > 1 else ""),
                       "culling": not args.no_cull},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "rt_render launch (camera, trace/group_trace per stage, path/leaf/shade logic, resolve)",
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "node_pairs_per_ray": work.nodePairSteps / max(work.rays, 1), "tri_tests_per_ray": work.triangleTests / max(work.rays, 1)},
        }
    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(sc, wl["feat"], wl["start_depth"], W, H, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
