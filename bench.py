#!/usr/bin/env python3
"""bench.py -- scenes/sec of the PointGroup detector training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N=1 default)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): PointGroup detector only, the canonical synthetic ScanNet-shaped scene of
SURVEY.md section 8(d) (200x150x100 grid @ 2 cm, 142,920 voxels = 4.8 % occupancy, ~164k points, 134 input channels), one
scene per GPU per step (weak scaling), random-init weights (seed 123), "teacher" clustering inputs (labels and GT
offsets drive the ball query / BFS so that the clustering stage carries a realistic load with untrained weights).
A step = feed (voxelise -> sparse U-Net -> heads -> 2x ball query + BFS clustering -> cluster re-voxelisation ->
ScoreNet -> proposals) + loss + backward + gradient all-reduce (N>1) + AdamW step, inputs resident in HBM.
One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-teacher", action="store_true", help="cluster on the network's own predictions")
    ap.add_argument("--small", action="store_true", help="quarter-size scene (debug)")
    return ap.parse_args()


CPU_THREADS = 8   # torch-CPU sparse conv is fastest at ~8 threads (256 threads on the GPU box: 1000x slower)


def cpu_baseline_child():
    """`bench.py --cpu-baseline-only`: the oracle (CPU restatement of the reference step) timed on this host:
    one forward+loss+backward of the SAME canonical scene.  Never touches the GPU.  Prints one JSON object."""
    import numpy as np
    import torch
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup
    from oracle import pg_oracle as pg
    from oracle.pointgroup_oracle import PointGroupOracle
    cores = min(os.cpu_count() or 1, CPU_THREADS)
    torch.set_num_threads(cores)
    cfg = default_conf()
    torch.manual_seed(cfg.general.manual_seed)
    state_dict = PointGroup(cfg).state_dict()          # same random init as the GPU run (CPU tensors)
    occ, sem, inst, _ = S.occupancy_grid()
    scene = S.scene_from_grid(occ, sem, inst)
    b = S.collate([scene])
    vl, p2v, v2p = pg.voxelization_idx(b["locs_scaled"], 1, 4)   # loader-side work, not timed (as on the GPU)
    b["voxel_locs"], b["p2v_map"], b["v2p_map"] = vl, p2v, v2p
    cpu = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
    orc = PointGroupOracle(cfg, state_dict)
    orc.teacher = True
    t0 = time.time()
    d = orc.loss(orc.feed(cpu, 0))
    d["total_loss"].backward()
    dt = time.time() - t0
    print(json.dumps({"value": 1.0 / dt, "unit": "scenes/sec", "cores": cores, "kind": "port",
                      "sample": "1 step (forward+loss+backward, no optimizer) of the same %d-point canonical scene "
                                "through oracle/ (torch-CPU gather-mm sparse conv, %d threads; C ball query / BFS / "
                                "segment ops single-threaded): %.1f s" % (cpu["locs"].shape[0], cores, dt)}), flush=True)


def cpu_baseline(limit_s=420):
    """run the baseline in a child process (bounded; it must never take the GPU number down with it)"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(CPU_THREADS)))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "error": (r.stderr or r.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": "cpu baseline exceeded %d s" % limit_s}


def main():
    args = parse()
    if args.cpu_baseline_only:
        return cpu_baseline_child()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI ("nccl" on ROCm); D3_DIST_BACKEND=gloo + D3_SHARE_DEVICE=1 is a plumbing test of the N>1 path on a
        # one-GPU box (all ranks on cuda:0), not a benchmark configuration
        dist.init_process_group(os.environ.get("D3_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("D3_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from d3net_amd import _lib, synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.pointgroup import PointGroup

    cfg = default_conf()
    torch.manual_seed(cfg.general.manual_seed)
    model = PointGroup(cfg).to(dev).train()
    model.teacher = not args.no_teacher
    params = [p for p in model.parameters() if p.requires_grad]
    from d3net_amd.optim import FusedAdamW
    opt = FusedAdamW(params, lr=cfg.train.optim.lr, weight_decay=cfg.train.optim.weight_decay)
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    if world > 1:  # identical replicas
        broadcast_module(model)
    # the executors' flat gradient buffers are all-reduced in place (RCCL, sum -> mean); heads share one packed collective
    grad_sync = BucketGradAllReduce(params, model.gradient_buckets) if world > 1 else None

    if args.small:
        occ, sem, inst, _ = S.occupancy_grid((100, 75, 50), 4, (8, 30), (8, 25), 0)
        scene = S.scene_from_grid(occ, sem, inst, feat_seed=2 + rank)
    else:
        occ, sem, inst, _ = S.occupancy_grid()
        scene = S.scene_from_grid(occ, sem, inst, feat_seed=2 + rank)   # same geometry, per-rank features
    batch = S.make_batch([scene], dev)
    n_points, n_voxels = int(batch["locs"].shape[0]), int(batch["voxel_locs"].shape[0])

    def step():
        d = dict(batch)
        model.zero_grad(set_to_none=True)
        loss, d = model.training_step(d)
        loss.backward()
        if grad_sync is not None:   # one fused gradient all-reduce over RCCL (sum -> mean), gradients only
            grad_sync()
        opt.step()
        return loss, d

    L = _lib.lib()
    # set-up, not a step of the run: one dry pass sizes the cached workspaces (the clustering scratch is ~3 GB) and loads
    # the code objects, so that a run with a very small --warmup does not time one-off allocations (reported as "setup")
    loss, d = step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        loss, d = step()
    torch.cuda.synchronize()
    # the interpreter's cyclic collector: a full (generation-2) pass every ~26 steps walks the whole module / tensor heap
    # (2-7 ms each, tools/step_jitter.py); freezing the long-lived objects after warm-up keeps those passes short
    import gc
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    PROF_STRIDE = 13   # every 13th convolution launch is bracketed by HIP events (coprime with the 257 launches per step)
    L.d3_prof_enable(PROF_STRIDE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, d = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel launch durations measured with HIP events on the launch stream during the timed region
    prof = {}
    for fam, name in ((0, "spconv_fwd2_kernel"), (2, "spconv_fwd2_split_kernel"), (1, "spconv_wgrad2_kernel")):
        n, ms, by, fl = C.c_longlong(0), C.c_double(0), C.c_double(0), C.c_double(0)
        L.d3_prof_collect(fam, C.byref(n), C.byref(ms), C.byref(by), C.byref(fl))
        prof[name] = dict(launches=n.value, total_ms=ms.value, bytes=by.value)   # the sampled launches
    L.d3_prof_enable(0)

    if rank == 0:
        dom = max(prof, key=lambda k: prof[k]["total_ms"])
        pd = prof[dom]
        avg_ms = pd["total_ms"] / max(pd["launches"], 1)
        achieved = (pd["bytes"] / max(pd["launches"], 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch of the dominant kernel from the PMC counters (cannot be sampled from inside this
        # process): the committed rocprofv3 --pmc measurement of this same command (tools/gpu_round.sh)
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            traffic = pmc[dom]["hbm_bytes_per_launch"]
            traffic_src = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; (2*FETCH+WRITE)*1024)"
        except Exception:
            pass
        out = {
            "metric": "scenes/sec fwd+bwd (PointGroup detector)", "value": world * args.steps / elapsed,
            "unit": "scenes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: PointGroup detector only, canonical synthetic ScanNet scene "
                                   "(200x150x100 @ 2 cm, %d voxels, %d points, 134 ch), 1 scene/GPU/step, m=16, 7-level "
                                   "U-Net, teacher clustering, AdamW" % (n_voxels, n_points),
                       "scenes_per_gpu": 1, "points": n_points, "voxels": n_voxels,
                       "raw_proposals": int(d.get("num_raw_proposals", 0)), "parallelism": "scene-parallel dp%d" % world,
                       "precision": "fp32 residual stream, bf16 BN->ReLU activations and MFMA operands, fp32 accumulate",
                       "setup": "1 untimed dry-run step before the warm-up (workspace allocation, code-object loads)"},
            "final_loss": float(loss.detach()),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": pd["bytes"] / max(pd["launches"], 1),
                         "launches_per_step": pd["launches"] * PROF_STRIDE / args.steps, "avg_launch_us": avg_ms * 1e3,
                         "launches_sampled": pd["launches"],
                         "timing": "HIP events on the launch stream around every %d-th convolution launch of the timed "
                                   "region, minus the elapsed time of an empty event pair" % PROF_STRIDE,
                         "share_of_step": pd["total_ms"] * PROF_STRIDE / (1e3 * elapsed),
                         "other": {k: {"launches_per_step": v["launches"] * PROF_STRIDE / args.steps,
                                       "avg_launch_us": 1e3 * v["total_ms"] / max(v["launches"], 1),
                                       "achieved": (v["bytes"] / max(v["total_ms"], 1e-9)) / 1e6}
                                   for k, v in prof.items() if k != dom}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
