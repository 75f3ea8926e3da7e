#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms/frame of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one frame, as the reference's renderImage delivers it: render (+ exchange of the row stripes when N > 1) +
rt_bloom + the copy of the finished frame into a pinned host buffer.  The scene is resident in HBM.  Frames are
independent (the reference renders one per camera, src/main.cpp:515-526), and the path stage of ONE frame is a chain of
short dependent launches that leaves most of the chip idle, so the steps are rendered `--frames-per-sequence` at a time by
rt_render_frames - one launch sequence whose stages trace the rays of all its frames together - and up to
`--frames-in-flight` such sequences are queued on streams of their own and overlap on the GPU.  `ms_per_step` is the
elapsed time of the K steps divided by K, `ms_per_frame_alone` what one frame takes with nothing else queued (rt_render).
With N > 1 every frame is cut into interleaved 8-row stripes, stripe s belongs to rank s % N; the stripes of a sequence
are exchanged with one all-to-all that leaves frame f whole on rank f % N, which blooms it and copies it to the host:
every frame of the job ends in host memory once (`--partition frames` is the comparison: whole frames dealt to the ranks in turn,
nothing exchanged).  Rank 0 prints one JSON line.  `value` counts every closest-hit query
that contributes to the image (primary + shadow + reflection rays) over all ranks, divided by the slowest rank's time.
"""
import argparse
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
    "dragon_glossy1_bloom": dict(scene="dragon", feat=dict(enableGlossyReflection=1, numGlossySamples=1, enableBloomEffect=1),
                                 start_depth=2, desc="dragon stand-in 1080p, Phong + shadows, glossy x1, 4 bounces, bloom F=15"),
    "dragon_glossy16_bloom": dict(scene="dragon", feat=dict(enableGlossyReflection=1, numGlossySamples=16, enableBloomEffect=1),
                                  start_depth=2, desc="dragon stand-in 1080p, Phong + shadows, glossy x16, 4 bounces, bloom F=15"),
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sc, feat_kw, start_depth, W, H, budget_s):
    """Times the CPU oracle (a port of the reference algorithm incl. its unused glossy debug query) on a
    bounded, evenly spread sample of rows of the same frame: on all of this host's cores the harness may use
    (OMP_NUM_THREADS, else the CPU count) for three quarters of the budget, then on one thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    kw = dict(FEATURE_BASE)
    kw.update({k: v for k, v in feat_kw.items() if k != "enableBloomEffect"})
    f = O.make_features(glossyDebugRay=1, **kw)
    cam = O.make_camera(W, H)
    s = O.OracleScene(sc["prims"], sc["materials"], sc["lights"], sc["spheres"])

    def timed(threads, budget):
        t0 = time.perf_counter()
        probe = s.render(cam, f, W, H, start_depth=start_depth, want_ids=False, threads=threads, row_step=64)
        tp = time.perf_counter() - t0
        step = 1
        while tp * 64 / step > budget and step < 64:
            step *= 2
        # a sample of about `budget` seconds: every step-th row of the frame, repeated while time remains
        reps, dt, r = 0, 0.0, probe
        while reps == 0 or (dt + dt / reps <= budget and reps < 64):
            t0 = time.perf_counter()
            r = s.render(cam, f, W, H, start_depth=start_depth, want_ids=False, threads=threads, row_step=step)
            dt += time.perf_counter() - t0
            reps += 1
        st = r["stats"]
        rays = st["primary"] + st["shadow"] + st["secondary"]
        return rays * reps / dt / 1e6, (f"every {max(step, 1)}th row of the {W}x{H} frame, {reps} time(s) ({rays} rays per pass, {dt:.2f} s in all; "
                                        f"{st['debug']} extra debug queries per pass executed but not counted)")

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    v_all, sample = timed(threads, 0.75 * budget_s)
    v_one, sample_one = timed(1, 0.25 * budget_s)
    return dict(value=v_all, unit="Mrays/s", cores=threads, kind="port", sample=sample, cpu_model=cpu_model(),
                host_cpus=os.cpu_count(), value_1_thread=v_one, sample_1_thread=sample_one)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="dragon_glossy_bloom", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0, help="default: the workload's (1920 unless it says otherwise)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--obj", default=None, help="Wavefront OBJ to use where a workload names the dragon")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-cull", action="store_true", help="traverse exactly as the reference (no distance culling)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="launch sequences queued at once, each on its own stream (1..8; default: 4 on one GPU, 2 on several)")
    ap.add_argument("--frames-per-sequence", type=int, default=0,
                    help="frames rendered by one launch sequence (rt_render_frames; 1..16; default: 8 on one GPU, 16 on several); fewer when "
                         "the steps do not fill the streams")
    ap.add_argument("--hw-queues", type=int, default=8, help="GPU_MAX_HW_QUEUES for this process (0: leave the environment's value)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend when N > 1: nccl (= RCCL over xGMI) or gloo (stripes staged through host memory)")
    ap.add_argument("--stream-priorities", default="", help="priority of stream k = entry k modulo the list's length (-1: high, 0: normal; '0': all equal; "
                                                            "default: the first half of the streams high, e.g. -1,-1,0,0)")
    ap.add_argument("--partition", default="stripes", choices=["stripes", "frames"],
                    help="N > 1: 'stripes' = every frame cut into interleaved 8-row stripes over the ranks + one all-to-all per sequence (the "
                         "default); 'frames' = whole frames dealt to the ranks in turn, no exchange (the comparison)")
    ap.add_argument("--keep-pooling", action="store_true", help="pool --frames-per-sequence frames even when a frame renders in under 3 ms")
    ap.add_argument("--sequence-sizes", default="", help="frames of the timed launch sequences, in queue order (comma list; default: see sequences())")
    args = ap.parse_args()

    # Streams map onto the HIP runtime's hardware queues, 4 per process by default: with 4 frame streams next to the null
    # stream and the copy back, two streams share a queue and their launches wait for each other.  Must be set before the
    # runtime starts.
    if args.hw_queues > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)
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
    n_dev = torch.cuda.device_count()
    device_index = local_rank % n_dev  # ranks may share a GPU (rehearsals on a one-GPU box)
    torch.cuda.set_device(device_index)
    api.check(api.lib().rt_set_device(device_index))
    if world > 1:
        import datetime
        patience = datetime.timedelta(seconds=300)  # a collective a rank never joins ends the job with an error instead of hanging it
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=patience)
        else:
            dist.init_process_group("gloo", timeout=patience)
    dev = torch.device("cuda", device_index)

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
    by_frames = world > 1 and args.partition == "frames"  # rank r renders frames r, r + N, ... whole; nothing is exchanged
    region = api.Region(8, 0, 1) if by_frames else api.Region(8, rank, world)
    flags = api.RT_RENDER_NO_CULL if args.no_cull else 0
    my_steps = len(range(rank, args.steps, world)) if by_frames else args.steps       # frames this rank renders (a share of each, or whole ones)
    my_warmup = len(range(rank, args.warmup, world)) if by_frames else args.warmup

    # One rank of several renders a share of every frame: its stages are small, and fewer, longer sequences serve it better
    # (one rank of eight, 20 steps: 0.73 ms per frame with 2 sequences of 10 in flight, 0.81 ms with 4 of 5).
    whole = world == 1 or by_frames  # this rank renders whole frames
    # (a short job fills three streams with longer sequences better than four with short ones: 20 steps as 7 + 7 + 6 take 3.58 ms per
    # step, as 4 x 5 3.69 - sequences of five pool too few rays per stage)
    F = max(1, min(8, args.frames_in_flight or ((4 if my_steps >= 32 else 3) if whole else 2)))
    K = max(1, min(api.RT_MAX_FRAMES, args.frames_per_sequence or (8 if whole else 16), -(-max(my_steps, 1) // F)))  # frames per launch sequence
    if args.sequence_sizes:
        K = max(1, min(api.RT_MAX_FRAMES, max(int(x) for x in args.sequence_sizes.split(","))))
    elif K > 2 and not args.keep_pooling:
        # Frames that render faster than a few copies of a frame to the host (25 MB: 0.45 ms) are bound by those copies, and
        # long sequences only bunch the copies up: such frames go two at a time.  Measured here, before anything is timed.
        probe = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        for _ in range(2):
            scene.render_device(cam, feat, W, H, wl["start_depth"], probe.data_ptr(), region=region, flags=flags, stream=0)
            try:
                scene.sync(0)
            except api.RtError as e:  # queues growing: a heavy frame
                if e.code != api.RT_ERR_AGAIN:
                    raise
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        scene.render_device(cam, feat, W, H, wl["start_depth"], probe.data_ptr(), region=region, flags=flags, stream=0)
        torch.cuda.synchronize()
        probe_ms = torch.tensor([(time.perf_counter() - t1) * 1e3], dtype=torch.float64)
        if world > 1:  # every rank must cut the steps into the same sequences (their exchanges pair up): the slowest rank's time decides
            if args.backend == "nccl":
                probe_ms = probe_ms.to(dev)
            dist.all_reduce(probe_ms, op=dist.ReduceOp.MAX)
        if float(probe_ms.item()) < 3.0:
            K = 2
            if whole and not args.frames_in_flight:
                F = 4  # (frames bound by their copies: 20 glossy x1 steps 0.91 ms per step on four streams, 1.03 on three)
        del probe
        scene.trim()  # the probe's buffers (a scene's first frame is given room for the worst case); what it held sizes the sequences' buffers
    if by_frames:  # every rank cuts its own frames into sequences of the same size: the smallest any rank arrived at
        kk = torch.tensor([float(K)], dtype=torch.float64)
        if args.backend == "nccl":
            kk = kk.to(dev)
        dist.all_reduce(kk, op=dist.ReduceOp.MIN)
        K = int(kk.item())
    K = max(1, min(K, (8 * 1920 * 1080 * (1 if by_frames else world)) // (W * H)))  # ... within 16.6 M pixels per rank (about 3 KB of queues and path states per pixel)
    flags_timed = flags | (api.RT_RENDER_OVERLAPPED if F > 1 else 0)
    # Equal sequences started together on equal streams run in step; with half of the streams at the higher of the GPU's two
    # stream priorities they drift apart (5 x 4: 4.08-4.10 against 4.20-4.23 ms per frame; 10 x 2: 4.08 against 4.53)
    prio = [int(x) for x in args.stream_priorities.split(",")] if args.stream_priorities else [-1 if k < (F + 1) // 2 else 0 for k in range(F)]
    streams = [torch.cuda.Stream(device=dev, priority=prio[k % len(prio)]) for k in range(F)]
    frames = [torch.zeros((K, H, W, 3), dtype=torch.float32, device=dev) for _ in range(F)]
    scratch = [torch.empty(api.lib().rt_bloom_scratch_bytes(W, H) // 4, dtype=torch.float32, device=dev) for _ in range(F)]
    host = [torch.empty((K, H, W, 3), dtype=torch.float32, pin_memory=True) for _ in range(F)]  # what the caller's Screen::pixels() alias
    from raytracer_amd.stripes import StripePlan
    plans = [StripePlan(H, W, rank, world, stripe_rows=8, device=dev if args.backend == "nccl" else "cpu") for _ in range(F)]
    last_count = [0] * F

    def sequences(n):
        """n steps as launch sequences of up to K frames: (first frame, frames).  Equal sequences started together on F streams
        run in step - F launches of one kind at a time, all after the same resource - and a run of three waves or more repays
        putting them out of step with a first wave of K, K - 2, K - 4 ... frames (96 steps, 8 x 4: 3.91 against 4.13 ms per
        frame); in a shorter run the shorter sequences cost what that saves."""
        sizes = [max(2, K - 2 * k) for k in range(F)] if (F > 1 and K >= 4 and n >= 3 * F * K) else []
        rest = n - sum(sizes)
        sizes += [K] * (rest // K) + ([rest % K] if rest % K else [])
        return [(sum(sizes[:i]), c) for i, c in enumerate(sizes)]

    step_of = (lambda j: rank + j * world) if by_frames else (lambda j: j)  # which step of the job this rank's j-th frame is

    def run(i, first, count, ev=None, fl=None, cam_list=None, host_dst=None):
        """Launch sequence i = frames [first, first + count): queued on stream i % F; nothing waits on the host.  ev: four events
        recorded on the sequence's stream - before the render, after it, after the stripe exchange, after bloom + copy."""
        k = i % F
        st = streams[k]
        src = cam_list if cam_list is not None else cams
        seq_cams = [src[step_of(first + f) % len(src)] for f in range(count)] if src else [cam] * count
        dst = host[k] if host_dst is None else host_dst
        last_count[k] = count
        with torch.cuda.stream(st):
            if ev:
                ev[0].record(st)
            scene.render_frames_device(seq_cams, feat, W, H, wl["start_depth"], frames[k].data_ptr(), region=region,
                                       flags=flags_timed if fl is None else fl, stream=st.cuda_stream)
            if ev:
                ev[1].record(st)
            mine = range(count)  # the frames this rank completes
            if world > 1 and not by_frames:
                if args.backend == "nccl":
                    mine = plans[k].assemble(frames[k][:count], dist)
                else:  # stripes through host memory: this rank's stripes down, the frames it is the root of up
                    st.synchronize()
                    cpu = frames[k][:count].cpu()
                    mine = plans[k].assemble(cpu, dist)
                    frames[k][:count].copy_(cpu)
            if ev:
                ev[2].record(st)
            if feat.enableBloomEffect:
                for f in mine:
                    api.bloom_device(frames[k][f].data_ptr(), scratch[k].data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold,
                                     feat.bloomFilterIntensity, stream=st.cuda_stream)
            if whole:
                dst[:count].copy_(frames[k][:count], non_blocking=True)
            else:
                for f in mine:
                    dst[f].copy_(frames[k][f], non_blocking=True)
            if ev:
                ev[3].record(st)
        return list(mine)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def fail(msg):  # a bench line over incomplete or wrong frames is worse than none
        print(f"bench.py rank {rank}: {msg}", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(3)

    def settle(cam_list=None):
        """Set-up, not a step: one sequence per stream, repeated while a frame's deferred tasks overflow the scene's queues
        (rt_sync: RT_ERR_AGAIN - the next render allocates what the frame asked for).  The first pass also allocates
        every stream's buffers."""
        for attempt in range(5):
            for i in range(F):
                run(i, i * K, K, cam_list=cam_list)
            barrier()
            again = False
            for k in range(F):
                try:
                    scene.sync(streams[k].cuda_stream)
                except api.RtError as e:
                    if e.code != api.RT_ERR_AGAIN:
                        raise
                    again = True
            if world > 1:  # a rank whose queues have grown renders its set-up sequences again: all ranks go along (the exchanges pair up)
                flag = torch.tensor([1.0 if again else 0.0], dtype=torch.float64)
                if args.backend == "nccl":
                    flag = flag.to(dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                again = bool(flag.item() > 0)
            if not again:
                return
        fail("the scene's task queues still grow after five set-up passes")

    settle()
    for i, (first, count) in enumerate(sequences(my_warmup)):
        run(i, first, count)
    barrier()
    stats = scene.sync(streams[0].cuda_stream)  # this rank's share of the last sequence queued on stream 0
    stats_frames = last_count[0]
    timed = sequences(my_steps)
    if args.sequence_sizes:
        sizes = [int(x) for x in args.sequence_sizes.split(",")]
        if sum(sizes) != my_steps or max(sizes) > K or min(sizes) < 1:
            raise SystemExit(f"--sequence-sizes must add up to {my_steps} steps, each within 1..{K}")
        timed = [(sum(sizes[:i]), n) for i, n in enumerate(sizes)]
    ranks_seen = [0]
    if world > 1:
        # Every rank must queue the same sequences (their exchanges pair up): a rank that disagrees would leave the others
        # waiting in a collective for ever.  Compared here, before anything is timed; a difference ends every rank at once.
        mine_desc = torch.full((66,), -1.0, dtype=torch.float64)
        mine_desc[0], mine_desc[1] = float(K), float(F)
        for j, (_, n) in enumerate(timed[:64]):
            mine_desc[2 + j] = float(n)
        if args.backend == "nccl":
            mine_desc = mine_desc.to(dev)
        all_desc = [torch.empty_like(mine_desc) for _ in range(world)]
        dist.all_gather(all_desc, mine_desc)
        ranks_seen = list(range(len(all_desc)))
        if dist.get_world_size() != world or any(not torch.equal(d, all_desc[0]) for d in all_desc):
            fail("the ranks do not agree on the launch sequences of the timed region: " + "; ".join(str([int(x) for x in d.tolist() if x >= 0]) for d in all_desc))

    def timed_region(seqs, cam_list=None):
        """The steps as launch sequences queued back to back on the F streams, between two barriers.  Every sequence copies
        its finished frames into a pinned host buffer of its own (within 6 GB in all; beyond that the streams' buffers are
        reused and only each stream's last sequence can be looked at afterwards).  After the second barrier, outside the timed
        region, every stream is asked for its status (rt_sync): a sequence that overflowed a queue ends the run."""
        evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in seqs]
        own = sum(n for _, n in seqs) * H * W * 12 <= (6 << 30)
        hosts = [torch.empty((n, H, W, 3), dtype=torch.float32, pin_memory=True) if own else host[i % F] for i, (_, n) in enumerate(seqs)]
        done = []
        barrier()
        t0 = time.perf_counter()
        for i, (first, count) in enumerate(seqs):
            done.append(run(i, first, count, evs[i], cam_list=cam_list, host_dst=hosts[i] if own else None))
        barrier()
        dt = time.perf_counter() - t0
        for k in range(min(F, len(seqs))):
            try:
                scene.sync(streams[k].cuda_stream)
            except api.RtError as e:
                fail(f"a timed launch sequence on stream {k} is incomplete: {e}")
        keep = range(len(seqs)) if own else range(max(0, len(seqs) - F), len(seqs))
        return dt, evs, {i: (seqs[i][0], done[i], hosts[i]) for i in keep}

    elapsed, timed_ev, timed_frames = timed_region(timed)
    launch_ms_timed = float(np.mean([e[0].elapsed_time(e[1]) for e in timed_ev]))  # one launch sequence among the others in flight
    stage_ms = [float(np.mean([e[a].elapsed_time(e[a + 1]) for e in timed_ev])) for a in range(3)]  # render, exchange, bloom + copy
    free_b, total_b = torch.cuda.mem_get_info(dev)
    device_mem_gb = (total_b - free_b) / 1e9

    # one frame with nothing else queued: its latency - from the call to the frame in host memory and rt_sync's word that it is
    # complete (which also hands the frame's round sizes to the next frame's launch plan, as for any caller that renders
    # frame after frame) - and the duration of the rt_render launch sequence by itself (events on the stream the launches run on)
    alone, kernel_ms = [], []
    for i in range(6):
        ev = tuple(torch.cuda.Event(enable_timing=True) for _ in range(4))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(0, i, 1, ev, flags)
        torch.cuda.synchronize()
        try:
            scene.sync(streams[0].cuda_stream)
        except api.RtError as e:
            fail(f"the frame rendered alone is incomplete: {e}")
        alone.append((time.perf_counter() - t1) * 1e3)
        kernel_ms.append(ev[0].elapsed_time(ev[1]))
    alone_ms, kernel_ms = float(np.median(alone[1:])), float(np.median(kernel_ms[1:]))

    # Every timed frame against the same camera's frame rendered ALONE - whole, by one rt_render with nothing else queued,
    # then rt_bloom - bit for bit, outside the timed region.  A frame that differs ends the run.
    ref_dev = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    ref_cache = {}

    def reference_frame(c, key):
        if key not in ref_cache:
            for attempt in range(4):
                scene.render_device(c, feat, W, H, wl["start_depth"], ref_dev.data_ptr(), flags=flags, stream=streams[0].cuda_stream)
                try:
                    st_ref = scene.sync(streams[0].cuda_stream)
                    break
                except api.RtError as e:
                    if e.code != api.RT_ERR_AGAIN or attempt == 3:
                        fail(f"the frame rendered alone is incomplete: {e}")
            if feat.enableBloomEffect:
                api.bloom_device(ref_dev.data_ptr(), scratch[0].data_ptr(), W, H, feat.bloomFilterSize, feat.bloomFilterThreshold,
                                 feat.bloomFilterIntensity, stream=streams[0].cuda_stream)
            streams[0].synchronize()
            if len(ref_cache) >= 2:
                ref_cache.clear()
            ref_cache[key] = (ref_dev.cpu().view(torch.int32), st_ref)
        return ref_cache[key]

    def verify(kept, cam_list):
        n_ok, rays = 0, 0
        for i, (first, done, hbuf) in sorted(kept.items()):
            for f in done:
                step = step_of(first + f)
                c, key = (cam_list[step % len(cam_list)], step % len(cam_list)) if cam_list else (cam, -1)
                ref, st_ref = reference_frame(c, key)
                if not torch.equal(hbuf[f].view(torch.int32), ref):
                    bad = int((hbuf[f].view(torch.int32) != ref).sum().item())
                    fail(f"timed frame {step} (sequence {i}) differs from the frame rendered alone in {bad} values")
                n_ok += 1
                rays += st_ref.rays
        return n_ok, rays

    verified, _ = verify(timed_frames, cams)
    del timed_frames

    # The same steps from distinct cameras on an orbit (rotation.y = 20 + 360 k / steps degrees), as the reference's loop over the
    # configuration's cameras renders them (src/main.cpp:515-526) - a second figure beside the headline, whose steps all look
    # through the metric configuration's one camera.
    orbit = None
    if world == 1 and not cams and args.steps > 1:
        ocams = [api.camera_from_trackball(W, H, rot_deg=(20.0, 20.0 + 360.0 * k / args.steps, 0.0)) for k in range(args.steps)]
        settle(ocams)  # other views fill the queues differently
        o_dt, _, o_frames = timed_region(timed, ocams)
        o_ok, o_rays = verify(o_frames, ocams)
        del o_frames
        if o_ok == args.steps:
            orbit = {"cameras": args.steps, "ms_per_step": o_dt / args.steps * 1e3, "value": o_rays / o_dt / 1e6, "unit": "Mrays/s",
                     "rays_per_frame_mean": o_rays / args.steps, "frames_verified": o_ok,
                     "note": "the same steps and sequences from distinct cameras, rotation.y = 20 + 360 k / steps degrees"}
        else:
            orbit = {"cameras": args.steps, "ms_per_step": o_dt / args.steps * 1e3, "frames_verified": o_ok,
                     "note": "ray total not taken: not every frame's host copy was kept"}
        settle()  # back to the metric configuration's camera for the counting passes below
    ref_cache.clear()
    del ref_dev

    s0 = streams[0].cuda_stream
    if cams:  # every frame of the sweep has its own ray count: recount them outside the timed region
        tot = [0, 0, 0, 0]
        for c in cams:
            scene.render_device(c, feat, W, H, wl["start_depth"], frames[0].data_ptr(), region=region, flags=flags, stream=s0)
            st = scene.sync(s0)
            for k, v in enumerate((st.primaryRays, st.shadowRays, st.secondaryRays, st.pixels)):
                tot[k] += v / len(cams)
        rays_rank = torch.tensor(tot, dtype=torch.float64)
    else:
        rays_rank = torch.tensor([stats.primaryRays, stats.shadowRays, stats.secondaryRays, stats.pixels], dtype=torch.float64) / stats_frames
    elapsed_rank = elapsed
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        if args.backend == "nccl":
            rays_rank, tmax = rays_rank.to(dev), tmax.to(dev)
        dist.all_reduce(rays_rank)
        if by_frames:  # every rank counted whole frames
            rays_rank = rays_rank / world
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    rays_frame = float(rays_rank[:3].sum().item())

    # Algorithmic bytes of rank 0's render launch (one rt_render = a fixed sequence of kernels on one stream, dominated by
    # the trace kernels), counted by a counting pass of the frame as it is scheduled and timed: 128 B per 4-wide node of
    # the query tree, 64 B per node pair (group launches, rays traced again in the caller's tree), 36 B per triangle test,
    # 28 B ray in + 16 B hit out per ray, 12 B per pixel written.  Candidate rays that are
    # discarded and rays traced twice are in the node and triangle counts.  The same figure for the reference's own walk,
    # which visits every box a ray enters (no culling, the caller's tree, 36 B per test), is reported beside it and is not
    # the basis of `achieved`.
    def counted(fl, schedule):
        n = K if schedule else 1  # the schedule of a sequence depends on how many frames share its launches
        scene.render_frames_device([cam] * n, feat, W, H, wl["start_depth"], frames[0].data_ptr(), region=region,
                                   flags=fl | (api.RT_RENDER_COUNT_SCHEDULE if schedule else api.RT_RENDER_COUNT_WORK), stream=s0)
        wk = scene.sync(s0)
        for name in ("queryNodeSteps", "nodePairSteps", "triangleTests", "primaryRays", "shadowRays", "secondaryRays", "pixels"):
            setattr(wk, name, getattr(wk, name) // n)
        # a triangle test: 36 B, the nine floats of its positions, as the coverage contract prices it (the layout read here is a
        # 16-byte plane record and, for candidates in range, an 80-byte edge record: 48 B with two edge reads in five tests -
        # reported beside it as algorithmic_bytes_layout)
        base = 128.0 * wk.queryNodeSteps + 64.0 * wk.nodePairSteps + 44.0 * wk.rays + 12.0 * wk.pixels
        wk.layout_bytes = base + (48.0 if schedule else 36.0) * wk.triangleTests
        return wk, base + 36.0 * wk.triangleTests
    try:
        work, alg_bytes = counted(flags_timed, True)
        counted_what = "one frame of a launch sequence as scheduled and timed (query tree, candidate batches, second traces)"
    except api.RtError:  # schedules without counters (a pixel per lane): the reference's rays in the caller's tree, culled
        try:
            work, alg_bytes = counted(flags, False)
        except api.RtError:  # recursion trees too long for rounds: the depth-first kernel counts
            work, alg_bytes = counted(flags | api.RT_RENDER_SERIAL, False)
        counted_what = "the reference's rays, culled walk in the caller's tree"
    try:
        ref_work, ref_bytes = counted(api.RT_RENDER_NO_CULL, False)
    except api.RtError:
        ref_work, ref_bytes = counted(api.RT_RENDER_NO_CULL | api.RT_RENDER_SERIAL, False)
    achieved = alg_bytes * args.steps / elapsed / 1e9          # sustained over the timed region, frames in flight together
    achieved_alone = alg_bytes / (kernel_ms * 1e-3) / 1e9      # one launch sequence with the GPU to itself
    traffic, traffic_note, limiter, per_kernel = None, None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(args.workload)
            traffic_note = tj.get("_measured")
            if traffic is not None:  # counters were taken on this workload: what they say bounds it, and the per-kernel table they come with
                limiter, per_kernel = tj.get("_limiter"), tj.get("_per_kernel")
        except Exception:
            traffic = None
    # HBM bytes per frame from the counters against the time a frame takes here: the fraction of the HBM roof the frame's real
    # traffic amounts to (the algorithmic bytes are mostly served by the caches: the tree and the triangles are a few MB)
    hbm_traffic_frac = traffic * args.steps / elapsed / 1e9 / HBM_PEAK_GBS if traffic else None

    # The same steps through the one blocking call a host program makes (rt_render_images: pageable images in and out, the library's
    # own streams and pinned staging) - reported beside the headline, never as `value`.  The scene's buffers of the runs above are
    # given back first: both sets together would hold some 200 GB.
    host_api = None
    if world == 1:
        try:
            del frames, scratch
            torch.cuda.empty_cache()
            scene.trim()
            hcams = cams if cams else [cam] * args.steps
            scene.render_images(hcams, feat, W, H, wl["start_depth"], flags)  # allocates the library's lanes
            t1 = time.perf_counter()
            hr = scene.render_images(hcams, feat, W, H, wl["start_depth"], flags)
            hdt = time.perf_counter() - t1
            host_api = {"entry": "rt_render_images", "frames": len(hcams), "ms_per_frame": hdt / len(hcams) * 1e3,
                        "mrays_per_s": hr["stats"].rays / hdt / 1e6, "note": "one blocking call, host images in and out (PCIe and host copies included)"}
        except Exception as e:  # never at the expense of the line
            host_api = {"entry": "rt_render_images", "error": str(e)[:200]}

    # what each rank's sequences spent where (events on their streams): render, stripe exchange, bloom + copy to the host
    per_rank = None
    if world > 1:
        mine_ms = torch.tensor([float(rank)] + stage_ms + [elapsed_rank * 1e3], dtype=torch.float64)
        if args.backend == "nccl":
            mine_ms = mine_ms.to(dev)
        all_ms = [torch.empty_like(mine_ms) for _ in range(world)]
        dist.all_gather(all_ms, mine_ms)
        per_rank = [{"rank": int(t[0].item()), "render_ms": float(t[1].item()), "exchange_ms": float(t[2].item()),
                     "bloom_copy_ms": float(t[3].item()), "elapsed_ms": float(t[4].item())} for t in all_ms]
    else:
        per_rank = [{"rank": 0, "render_ms": stage_ms[0], "exchange_ms": stage_ms[1], "bloom_copy_ms": stage_ms[2], "elapsed_ms": elapsed_rank * 1e3}]
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
            "ms_per_frame_alone": alone_ms,
            "orbit": orbit,
            "verified": {"timed_frames_equal_to_frames_rendered_alone": verified, "of": my_steps, "timed_sequences_complete": True,
                         "note": "after the timed region every stream's status was read with rt_sync and every timed frame kept in host "
                                 "memory was compared bit for bit with the same camera's frame rendered alone; a difference ends the run"},
            "host_api": host_api,
            "per_rank": {"ranks_seen": ranks_seen, "sequences": [n for _, n in timed], "stages_ms_per_sequence": per_rank,
                         "note": "mean over a rank's timed launch sequences, events on the sequence's stream; sequences of different streams overlap"},
            "mrays_per_s_one_frame_alone": rays_frame / world / alone_ms / 1e3 if world == 1 else None,
            "config": {"workload": args.workload, "description": wl["desc"], "width": W, "height": H,
                       "triangles": int(len(sc["prims"])), "mesh_hash": _scenes.mesh_hash(sc), "mesh": args.obj or wl["scene"],
                       "rays_per_frame": int(rays_frame), "frames_per_sequence": K, "sequences_in_flight": F, "frames_in_flight": K * F,
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "device_mem_gb": round(device_mem_gb, 2),
                       "step": "one frame: its share of rt_render_frames" + (" + stripe all-to-all (frame f whole on rank f % N)" if world > 1 and not by_frames else "")
                               + " + rt_bloom + frame copy to pinned host memory",
                       "partition": (f"whole frames dealt to {world} GPUs in turn, no exchange" if by_frames else
                                     f"interleaved 8-row stripes over {world} GPU(s)" + (f", {args.backend} all-to-all" if world > 1 else "")),
                       "culling": not args.no_cull},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "hbm_traffic_frac": hbm_traffic_frac, "limiter": limiter,
                         "per_kernel": per_kernel,
                         "kernel": "launch sequence of rt_render_frames (camera, trace/group_trace per stage, path/leaf/shade logic, resolve)",
                         "note": "achieved = algorithmic bytes of one frame x steps / elapsed time of the timed region (sequences_in_flight "
                                 "launch sequences of frames_per_sequence frames share the GPU); launch_ms_timed = mean duration of one such "
                                 "sequence there, kernel_ms = the launch sequence of ONE frame alone (rt_render), both from events on the "
                                 "sequence's stream",
                         "launch_ms_timed": launch_ms_timed, "kernel_ms": kernel_ms, "achieved_one_launch_alone": achieved_alone,
                         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_layout": work.layout_bytes, "counted": counted_what,
                         "bytes_per_unit": "128 B per 4-wide query node, 64 B per node pair, 36 B per triangle test, 28 B ray in + 16 B hit out, 12 B per pixel",
                         "query_nodes_per_ray": work.queryNodeSteps / max(work.rays, 1),
                         "node_pairs_per_ray": work.nodePairSteps / max(work.rays, 1), "tri_tests_per_ray": work.triangleTests / max(work.rays, 1),
                         "reference_walk": {"bytes_per_launch": ref_bytes,
                                            "node_pairs_per_ray": ref_work.nodePairSteps / max(ref_work.rays, 1),
                                            "tri_tests_per_ray": ref_work.triangleTests / max(ref_work.rays, 1)}},
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
