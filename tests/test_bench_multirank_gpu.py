"""Plumbing test of bench.py's N>1 path on a one-GPU box: two ranks under torch.distributed.run share cuda:0 and average
their gradients through BucketGradAllReduce over gloo (D3_DIST_BACKEND / D3_SHARE_DEVICE are test switches; the benchmark
itself uses RCCL, one rank per GPU).  Checks the JSON contract of the printed line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line(dev):
    env = dict(os.environ, D3_DIST_BACKEND="gloo", D3_SHARE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--small"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["unit"] == "scenes/sec"
    assert out["value"] > 0 and out["final_loss"] == out["final_loss"]          # finite
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1
    assert "cpu_baseline" not in out                                            # rank 0 at N=1 only
    # the heads' bucket starts from inside backward() in every step after the first (which compares the layouts first):
    # 1 dry-run + 1 warm-up + 2 timed steps -> 3 early starts; and it changes nothing in the result
    gs = out["config"]["grad_sync"]
    assert gs["heads_bucket_floats"] > 0 and gs["heads_bucket_started_inside_backward"] == 3, gs
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, D3_EARLY_ALLREDUCE="0"), cwd=ROOT)
    late = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert late["config"]["grad_sync"]["heads_bucket_floats"] == 0
    assert abs(late["final_loss"] - out["final_loss"]) <= 1e-4 * abs(out["final_loss"]), (late["final_loss"], out["final_loss"])
