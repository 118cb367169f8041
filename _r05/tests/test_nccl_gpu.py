"""bench.py's N>1 path over RCCL with one rank per GPU (`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`):
needs two GPUs -- skipped on the one-GPU boxes of this pool, run by the driver's multi-GPU tier."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL over xGMI)")
def test_two_rank_rccl_bench_line():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--small",
           "--config", "detector"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"


def _run_world1(extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--settle", "2", "--small", "--no-cpu-baseline", "--no-fp32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-2500:])
    return json.loads(lines[0])


def test_one_rank_rccl_gradient_reducer_plumbing(dev):
    """The gradient reducer over RCCL itself, on ONE GPU: a world of one rank (D3_DIST_WORLD1=1) initialises the `nccl` process
    group and runs every collective of the step through RCCL -- in-place all-reduces of the executors' flat buffers and their tail
    chunks started from inside backward(), the heads' early bucket, the packed rest, AVG, the async handles' waits, the barrier /
    max-over-ranks timing -- next to the streams the executors own.  Averaging over one rank changes nothing: the loss must equal the
    run without a process group.  (The two-rank test above needs two GPUs; this one catches API-level breakage on this pool.)"""
    ref = _run_world1({})
    out = _run_world1({"D3_DIST_WORLD1": "1", "MASTER_PORT": "29578"})
    assert out["config"]["world"]["size_seen_by_process_group"] == 1 and out["config"]["world"]["backend"] == "nccl"
    gs = out["config"]["grad_sync"]
    assert gs["heads_bucket_floats"] > 0 and gs["heads_bucket_started_inside_backward"] > 0, gs
    assert gs["executor_chunk_collectives_started_inside_backward"] > 0, gs
    assert abs(out["final_loss"] - ref["final_loss"]) <= 1e-5 * abs(ref["final_loss"]), (out["final_loss"], ref["final_loss"])
    # the reducer must not cost the step more than its collectives' launch overhead (a starved or serialised stream would)
    assert out["ms_per_step"] < 1.5 * ref["ms_per_step"] + 2.0, (out["ms_per_step"], ref["ms_per_step"])
