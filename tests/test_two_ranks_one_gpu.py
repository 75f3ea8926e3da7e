"""The multi-rank render path executed end to end on one GPU: two fresh processes share GPU 0 and exchange
their stripes over gloo (host memory).  What runs is the code the N > 1 launch runs: rt_render with an
rt_region per rank, StripePlan's all-gather, bloom on the gathered frame (and rt_render_frames with StripePlan's all-to-all); only the transport differs from
RCCL.  No scaling figure comes out of this: two ranks on one GPU time-share it."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(script_args, timeout):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_render_stripes_and_gather_the_one_call_frame(tmp_path):
    out = tmp_path / "res.json"
    p = launch([os.path.join(ROOT, "tests", "two_rank_worker.py"), str(out), "328", "200"], 300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads(out.read_text())
    print("two ranks on one GPU:", res)
    assert res["foreign_rows_written"] == 0        # a rank writes only rows of its own stripes
    assert res["lit"] > 1000
    assert res["equal"], f"{res['differing']} values differ from the one-call frame"
    assert res["rays"] == res["rays_one_call"]     # the ranks' ray counts add up to the frame's
    assert res["sequence_frames_equal"]            # three frames in one launch sequence, each assembled on its root rank


def test_bench_runs_with_two_ranks_over_gloo():
    p = launch([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--workload",
                "teapot_monkey_phong", "--width", "320", "--height", "200", "--cpu-seconds", "0"], 300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    print("bench, 2 ranks on one GPU over gloo:", {k: j[k] for k in ("value", "ms_per_step", "n_gpus", "scaling")})
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["rays_per_frame"] > 320 * 200
    assert "gloo" in j["config"]["partition"]
