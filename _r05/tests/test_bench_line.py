"""bench.py's final stdout line must fit the driver's stdout tail (the 21.8 KB line of round 4 lost its
head -- metric, value, ms_per_step -- to the ~8 KB tail): compact_line() keeps it under 4 KB whatever the full record holds."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full_record():
    # the full line round 4's bench.py printed (21.8 KB: per_kernel / families / hg_gemm_shapes / compulsory-byte detail)
    return json.load(open(os.path.join(ROOT, "profiles", "r04_i_bench.json")))


def test_final_line_fits_the_driver_tail_and_keeps_the_contract():
    full = _full_record()
    assert len(json.dumps(full)) > 16000          # (the record that overflowed)
    o = bench.compact_line(full, "gpurun_out/bench_detail_speaker.json")
    line = json.dumps(o)
    assert len(line) < bench.LINE_BUDGET == 4096, len(line)
    for k in CONTRACT:
        assert k in o, k
    assert o["metric"] == full["metric"] and abs(o["value"] - full["value"]) < 1e-3 * full["value"]
    assert abs(o["ms_per_step"] - full["ms_per_step"]) < 1e-3 * full["ms_per_step"]
    rf = o["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 2e-3 * rf["frac"]
    cb = o["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert o["config"]["workload"].startswith("BASELINE configs[2]") and "model" not in o["config"]
    assert o["detail"].endswith(".json")


def test_final_line_survives_a_bloated_record():
    """multi-rank extras, long notes and a huge per-kernel table still give a parsable line under the budget"""
    full = _full_record()
    full["config"]["grad_sync"] = {"collectives_per_step": 4, "ranks_seen_by_backend": 8, "backend": "nccl",
                                   "bytes_per_collective": [3_000_000, 1_200_000, 31_000_000, 900_000],
                                   "executor_chunks": [[300000], [7750000]], "heads_bucket_floats": 750000}
    full["config"]["per_rank_ms_per_step"] = [18.3] * 8
    full["config"]["launched_by"] = "bench.py itself: 8 child ranks via torch.distributed.run, rc 0"
    full["config"]["workload"] += " " + "x" * 3000
    full["roofline"]["per_kernel"].update({"k%d<%s>" % (i, "true, " * 8): dict(frac=0.1) for i in range(200)})
    full["strong_scaling_ceiling"].update(t_4_scenes_ms=18.3, t_32_scenes_ms=131.0, ratio_32=7.2)
    o = bench.compact_line(full, "gpurun_out/bench_detail_speaker.json")
    line = json.dumps(o)
    assert len(line) < bench.LINE_BUDGET, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "ms_per_step", "config", "roofline", "cpu_baseline"):
        assert back.get(k) is not None, k


def test_line_without_optional_parts():
    full = _full_record()
    for k in ("cpu_baseline", "fp32_exact", "strong_scaling_ceiling"):
        full[k] = None
    full["roofline"] = None
    o = bench.compact_line(full)
    assert o["roofline"] is None and "cpu_baseline" not in o and len(json.dumps(o)) < bench.LINE_BUDGET
