#!/usr/bin/env python3
"""bench.py -- scenes/sec of D3Net's training step on MI355X (BASELINE.json metric).

    python bench.py [--config speaker|detector|listener|joint] --gpus N --steps K --warmup W [--scaling weak|strong]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment bench.py starts its N ranks ITSELF (one child process group through
torch.distributed.run, as the reference spawns its DDP ranks: scripts/train.py:265-268), decided before anything touches the
GPU; rank 0's line is relayed and the children's return code is this process's.

Default workload = the configuration BASELINE.json's metric is quoted on ("scenes/sec fwd+bwd (PointGroup+speaker)",
configs[2]): `PipelineNet` mode 1 (reference step: model/pipeline.py:152-185) with conf/pointgroup_captioning.yaml --
batch_size 4 scenes per GPU per step, 8 descriptions per scene, vocabulary 3004, cross-entropy captioning -- on the 40-box
synthetic ScanNet-shaped scenes of SURVEY.md 8(d) "Config 3" (200x150x100 grid @ 2 cm, 40 hollow boxes of 8..30 cells per
side, ~160 k voxels / ~185 k points per scene, 134 input channels), random-init weights (seed 123), "teacher" clustering
inputs (labels and GT offsets drive the ball query / BFS so that the clustering stage carries a realistic load with
untrained weights: ~40 instances per scene).  A step = detector feed (voxelise -> sparse U-Net -> heads -> 2x ball query +
BFS -> cluster re-voxelisation -> ScoreNet -> batched proposals) + relation graph + top-down captioner (teacher forcing)
+ losses + backward + gradient all-reduce (N>1) + AdamW, inputs resident in HBM.  Weak scaling: every rank runs its own
4 scenes.  `--config detector` is BASELINE configs[1] (one canonical 142,920-voxel scene per step), `listener` configs[3]
(mode 2), `joint` configs[4] (mode 3, self-critical).  One JSON line on stdout (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_16x16x4_f32 (exact fp32 in / fp32 accumulate), dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (quoted beside the HBM fraction of the convolutions: the secondary figure of SURVEY 8(d))
PROF_TAGS = 12
VOCAB = 3004
CONF = {"speaker": "pointgroup_captioning.yaml", "detector": "pointgroup.yaml", "listener": "pointgroup_grounding.yaml",
        "joint": "pointgroup_joint.yaml"}
METRIC = {"speaker": "scenes/sec fwd+bwd (PointGroup+speaker)", "detector": "scenes/sec fwd+bwd (PointGroup detector)",
          "listener": "scenes/sec fwd+bwd (PointGroup+listener)", "joint": "scenes/sec fwd+bwd (PointGroup+speaker+listener, self-critical)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONF), default="speaker")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle", type=int, default=SETTLE_STEPS,
                    help="untimed steps between the dry pass and the --warmup steps (a fresh box reaches its sustained rate after some tens of steps)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the untimed exact-fp32 (reference precision) steps")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the untimed 8-scene / 1-scene steps behind `strong_scaling_ceiling`")
    ap.add_argument("--scene-count", type=int, default=0, help=argparse.SUPPRESS)   # (child of the 32-scene ceiling measurement: scenes 0..N-1 on this rank)
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-teacher", action="store_true", help="cluster on the network's own predictions")
    ap.add_argument("--no-prefetch", action="store_true", help="build every step's input stage inside the step (no look-ahead on a side stream)")
    ap.add_argument("--small", action="store_true", help="quarter-size scenes (debug)")
    ap.add_argument("--exact", action="store_true",
                    help="time the step at the REFERENCE'S precision (minkowski.set_exact: fp32 storage, fp32 MFMA convolutions) instead of bf16")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the config's scenes per rank per step (4); strong: the GLOBAL batch is fixed at 8 scenes, 8/N per rank")
    return ap.parse_args()


CPU_THREADS = 8            # default thread count of the CPU baseline child (the parent sweeps CPU_THREAD_SWEEP; all 256 hardware threads of the GPU box: 1000x slower)
CPU_THREAD_SWEEP = (8, 16, 32)   # the sweep behind that "8", re-run with every bench line (the best one is `cpu_baseline.value`)
CPU_WARMUP_STEPS, CPU_TIMED_STEPS = 1, 3   # per thread count: ~4 x 5 s of CPU work


# ------------------------------------------------------------------------------------------ workloads
SETTLE_STEPS = 40            # untimed steps between the dry pass and the --warmup steps (see main())
STRONG_GLOBAL_BATCH = 8


def make_scenes(config, rank, small=False, scene_ids=None):
    """the synthetic scenes of one rank's step (numpy, host side); scene_ids: global scene numbers (strong scaling)"""
    from d3net_amd import synthetic as S
    if config == "detector":
        if small:
            occ, sem, inst, _ = S.occupancy_grid((100, 75, 50), 4, (8, 30), (8, 25), 0)
        else:
            occ, sem, inst, _ = S.occupancy_grid()
        return [S.scene_from_grid(occ, sem, inst, feat_seed=2 + rank)]   # same geometry, per-rank features
    scenes = []
    ids = [4 * rank + b for b in range(4)] if scene_ids is None else scene_ids   # data.batch_size 4 (conf/pointgroup_captioning.yaml)
    for g in ids:        # 40-box variant of SURVEY.md 8(d)
        dims, nb, side = ((100, 75, 50), 10, (6, 16)) if small else ((200, 150, 100), 40, (8, 30))
        occ, sem, inst, _ = S.occupancy_grid(dims, nb, side, side, seed=g)
        scenes.append(S.scene_from_grid(occ, sem, inst, seed=1 + g % 4, feat_seed=2 + g))
    return scenes


def make_dataset(n_scenes, chunk, joint):
    """the `dataset["train"]` object PipelineNet reads: vocabulary, GloVe table (N(0,1), seed 3), RL annotation store"""
    import numpy as np
    from d3net_amd import synthetic as S
    ds = types.SimpleNamespace(vocabulary=S.make_vocabulary(VOCAB),
                               glove=np.random.default_rng(3).standard_normal((VOCAB, 300)).astype(np.float32))
    if joint:
        ds.chunked_data, ds.organized = S.make_language_corpus(n_scenes, chunk=chunk, vocab=VOCAB)
    return {"train": ds}


def code_sha():
    """content hash of what the PMC traffic figures depend on (bench.py + the HIP sources): profiles/pmc_traffic.json carries
    the hash of the tree it was measured on, and the bench line says `traffic_stale` when that is not this tree"""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "d3net_amd", "csrc")
    for f in [os.path.abspath(__file__)] + sorted(os.path.join(csrc, n) for n in os.listdir(csrc) if n.endswith((".hip", ".h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_info():
    """CPU model, sockets, physical cores and hardware threads of this host (lscpu; /proc/cpuinfo as a fallback)"""
    import subprocess
    info = {"model": None, "sockets": None, "physical_cores": None, "threads": os.cpu_count()}
    try:
        kv = {}
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if ":" in line:
                k, v = line.split(":", 1)
                kv[k.strip()] = v.strip()
        info["model"] = kv.get("Model name")
        info["sockets"] = int(kv.get("Socket(s)", 0)) or None
        if info["sockets"] and kv.get("Core(s) per socket"):
            info["physical_cores"] = info["sockets"] * int(kv["Core(s) per socket"])
    except Exception:
        pass
    if info["model"] is None:
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    info["model"] = line.split(":", 1)[1].strip()
                    break
        except Exception:
            pass
    return info


def compulsory_bytes(detector, batch):
    """Compulsory HBM bytes of ONE step's two sparse U-Nets + input pooling by SURVEY.md 8(d)'s formulas (every feature row read
    once and written once, weights once, one (in, out) int32 pair per kernel-map entry; bf16 storage e = 2; backward = 2x the
    forward): per convolution `e*(Nin*Cin + Nout*Cout) + e*K*Cin*Cout + 8*P`; input pooling `4*N*C + 4*M*(mA+1) + 4*M*C`.
    Row counts and kernel-map pair counts are those of the step's own coordinate pyramids (read once, outside the timed region)."""
    from d3net_amd import netexec
    total, detail = 0.0, {}
    for name, ex in detector._execs.items():
        if ex is None or ex.debug_last is None:
            continue
        rows, pairs3 = ex.debug_last[1], ex.debug_pairs
        fwd = 0.0
        for op in ex.b.ops:
            if op[0] != netexec.OP_CONV:
                continue
            _, x, out, _res, _w, kind, mlevel, K, cin, _st = op[:10]
            lin, lout = ex.b.tensors[x][0], ex.b.tensors[out][0]
            cout = ex.b.tensors[out][1]
            nin, nout = rows[lin], rows[lout]
            P = pairs3[mlevel] if kind == netexec.MAP_K3 else (max(nin, nout) if kind in (netexec.MAP_DOWN, netexec.MAP_UP) else nout)
            fwd += 2.0 * (nin * cin + nout * cout) + 2.0 * K * cin * cout + 8.0 * P
        detail[name] = {"rows": list(rows), "pairs27": list(pairs3), "forward_bytes": fwd}
        total += 3.0 * fwd
    N, C = batch["feats"].shape[0], batch["feats"].shape[1] + 3
    M, mA1 = batch["v2p_map"].shape
    pool = 4.0 * N * C + 4.0 * M * mA1 + 4.0 * M * C
    detail["input_pooling_bytes"] = pool
    return total + pool, detail



def _b(x):
    return "true" if x else "false"


def kernel_family(name):
    """the kernel function a rocprofv3 name belongs to; the entry points that share spconv_fwd2_body (csrc/spconv2.hip: the
    wave-per-tile forward / data-gradient convolution) count as ONE function, under the name the earlier rounds' records carry"""
    base = name.split("<")[0].replace("void ", "").strip('" ')
    # round 6: spconv_fwd3_kernel (csrc/spconv3.hip, the K = 27 layers of the big levels on the lane table) is the third generation of
    # the same function -- the wave-per-tile forward / data-gradient convolution; the second generation keeps the stem, the stride-2 /
    # transposed / 1x1 layers and the shapes without an instance.  One family, named after both.
    return "spconv_fwd3_kernel+spconv_fwd2_kernel" if base in ("spconv_fwd2_c_kernel", "spconv_fwd2_kernel", "spconv_fwd3_kernel") else base


def prof_kernel_name(fam, t):
    """the kernel a profiling record timed, named as rocprofv3 prints it (template arguments included where the record has them)"""
    t = [int(v) for v in t]
    if fam == 0:
        nt, wlds, xbf, nw, f32, kt, st = t[5:12]
        st, fl = st % 1000, st // 1000            # (the record packs flags into the ST tag: + 1000 the T16 template flag, + 4000 / + 8000 the kernel)
        t16 = bool(fl & 1)
        if fl & 8:      # spconv_fwd3_kernel<ST, NT, EPI, OBF, BXBF, NW, QC>: EPI / (OBF | BXBF << 1) / QC ride in the WLDS / XBF / F32M slots
            return "spconv_fwd3_kernel<%d, %d, %d, %s, %s, %d, %d>" % (st, nt, wlds, _b(xbf & 1), _b(xbf & 2), nw, f32)
        if fl & 4:
            return "spconv_fwd2_c_kernel<%d, %d, %d, %s>" % (nt, nw, st, _b(t16))
        return "spconv_fwd2_kernel<%d, %s, %s, %d, %s, %d, %d, %s>" % (nt, _b(wlds), _b(xbf), nw, _b(f32), kt, st, _b(t16))
    if fam == 2:
        return "spconv_fwd2_split_kernel<%d, %s, %s>" % (t[5], _b(t[6]), _b(t[7]))
    if fam == 1:
        return {3: "spconv_wgrad3_kernel", 2: "spconv_wgrad2_kernel", 1: "spconv_wgrad2_wide_kernel", 32: "spconv_wgrad_f32_kernel"}.get(t[5], "spconv_wgrad")
    if fam == 3:
        if t[4] in (0, 2):
            return "hg_gemm_tiled_kernel"
        return "hg_gemm_kernel<%d, %s, %d>" % (t[5], _b(t[6] > 0), max(t[6], 4))
    if fam == 4:
        return "td_gru4_fwd_kernel<1>"
    if fam == 5:
        return "cl_bfs2_kernel"
    return "family%d" % fam


def conv_bytes_8d(t, pairs27):
    """Bytes = e*(Nin*Cin + Nout*Cout) + e*K*Cin*Cout + 8*P with e = 2 (bf16 storage) and P = the kernel map's
    (in, out) pairs: the 27-offset rulebook size of the level for K = 27 (counted from this step's own maps), the fine level's
    rows for the stride-2 / transposed convolutions (K = 8: every fine row has exactly one parent), Nout for K = 1.
    FLOPs = 2*P*Cin*Cout.  The weight-gradient launch of a layer is priced like its forward (8(d): backward = 2x the forward)."""
    Min, Mout, K, Cin, Cout = (int(v) for v in t[:5])
    if K == 27:
        P = pairs27.get(Mout)
        if P is None:
            P = pairs27.get(Min, 9.3 * Mout)
    elif K == 8:
        P = max(Min, Mout)
    else:
        P = Mout * K
    return 2.0 * (Min * Cin + Mout * Cout) + 2.0 * K * Cin * Cout + 8.0 * P, 2.0 * P * Cin * Cout


def collect_kernel_rooflines(L, steps, stride, pairs27):
    """every sampled launch of the timed region (HIP events on its own stream, csrc/prof.h) -> per-kernel records:
    {name: {family, launches_sampled, launches_per_step, avg_launch_us, ms_per_step, bytes_8d, bytes_design, flops}}"""
    out = {}
    shapes = {}          # hg_gemm*: (kernel, max M, max N, K, problems) -> [launches sampled, ms]
    W = 3 + PROF_TAGS
    for fam in (0, 1, 2, 3, 4, 5):
        n = C.c_int(0)
        L.d3_prof_dump(fam, None, 0, C.byref(n))
        if n.value == 0:
            continue
        buf = (C.c_double * (W * n.value))()
        L.d3_prof_dump(fam, buf, n.value, C.byref(n))
        for i in range(n.value):
            row = buf[i * W:(i + 1) * W]
            ms, bdesign, flops, tags = row[0], row[1], row[2], row[3:]
            name = prof_kernel_name(fam, tags)
            if fam in (0, 1, 2):
                b8, flops = conv_bytes_8d(tags, pairs27)
            else:
                b8 = bdesign
            if fam == 3:
                k = (name.split("<")[0], int(tags[0]), int(tags[1]), int(tags[2]), int(tags[3]))
                sh = shapes.setdefault(k, [0, 0.0, 0.0])
                sh[0] += 1; sh[1] += ms; sh[2] += flops
            r = out.setdefault(name, dict(family=fam, n=0, ms=0.0, bytes_8d=0.0, bytes_design=0.0, flops=0.0))
            r["n"] += 1; r["ms"] += ms; r["bytes_8d"] += b8; r["bytes_design"] += bdesign; r["flops"] += flops
    res = {}
    for name, r in out.items():
        n = max(r["n"], 1)
        res[name] = {"family": r["family"], "launches_sampled": r["n"], "launches_per_step": r["n"] * stride / steps,
                     "avg_launch_us": 1e3 * r["ms"] / n, "ms_per_step": r["ms"] * stride / steps,
                     "algorithmic_bytes_per_launch": r["bytes_8d"] / n, "bytes_moved_by_design_per_launch": r["bytes_design"] / n,
                     "flops_per_launch": r["flops"] / n,
                     "achieved_gbs": r["bytes_8d"] / max(r["ms"], 1e-9) / 1e6, "achieved_tflops": r["flops"] / max(r["ms"], 1e-9) / 1e9}
    top = sorted(shapes.items(), key=lambda kv: -kv[1][1])[:10]
    res["__hg_shapes__"] = [{"kernel": k[0], "M": k[1], "N": k[2], "K": k[3], "problems": k[4], "launches_per_step": v[0] * stride / steps,
                             "avg_launch_us": 1e3 * v[1] / v[0], "tflops": v[2] / max(v[1], 1e-9) / 1e9} for k, v in top]
    return res


def roofline_object(name, r, traffic_table, stride):
    """the bench line's `roofline` for kernel `name`: HBM-bound kernels price SURVEY 8(d)'s algorithmic bytes against 8 TB/s; the
    dense fp32 GEMM of the heads (hg_gemm*) is MFMA-bound work priced against the 157.3 TFLOP/s fp32-MFMA peak, its byte-side
    fraction quoted beside it"""
    mfma = r["family"] == 3
    mfma_peak = MFMA_F32_PEAK_TFLOPS
    o = {"bound": "mfma" if mfma else "hbm", "kernel": name,
         "achieved": r["achieved_tflops"] if mfma else r["achieved_gbs"],
         "peak": mfma_peak if mfma else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma else "GB/s"}
    o["frac"] = o["achieved"] / o["peak"]
    t = traffic_table.get(name) or {}          # (exact instance only: a family average next to one instance's bytes would mislead)
    o["traffic"] = t.get("hbm_bytes_per_launch")
    o["algorithmic_bytes_per_launch"] = r["algorithmic_bytes_per_launch"]
    o["traffic_over_algorithmic"] = (o["traffic"] / r["algorithmic_bytes_per_launch"]) if (o["traffic"] and r["algorithmic_bytes_per_launch"]) else None
    o["bytes_moved_by_design_per_launch"] = r["bytes_moved_by_design_per_launch"]
    o["flops_per_launch"] = r["flops_per_launch"]
    o["hbm_frac"] = r["achieved_gbs"] / HBM_PEAK_GBS
    o["mfma_frac"] = r["achieved_tflops"] / (mfma_peak if r["family"] in (3, 4) else MFMA_BF16_PEAK_TFLOPS)
    o["launches_per_step"] = r["launches_per_step"]; o["avg_launch_us"] = r["avg_launch_us"]; o["launches_sampled"] = r["launches_sampled"]
    o["ms_per_step"] = r["ms_per_step"]
    o["timing"] = ("HIP events on the launch's own stream around every %d-th instrumented launch (convolutions, hg_gemm, GRU cell, BFS "
                   "replay) of the timed region, minus the elapsed time of an empty event pair" % stride)
    o["bytes"] = ("SURVEY.md 8(d): e*(Nin*Cin + Nout*Cout) + e*K*Cin*Cout + 8*P, e = 2, P from this step's kernel maps" if r["family"] in (0, 1, 2)
                  else "SURVEY.md 8(d): operands and outputs once (heads) / 4*nActive + 12*n + 8*S (BFS)")
    return o

LINE_BUDGET = 4096     # the driver keeps ~8 KB of stdout tail: the final line must fit with margin (tests/test_bench_line.py)


def _r(x, nd=4):
    """floats to nd significant digits (the side file keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    return x


def _pick(d, keys, nd=4):
    return {k: _r(d.get(k), nd) for k in keys if d is not None and k in d}


def compact_line(full, detail_path=None):
    """The ONE JSON line the driver parses: the contract's keys + `roofline`, `step_roofline`,
    `cpu_baseline`, `fp32_exact`, `strong_scaling_ceiling` in short form.  Everything else (per-kernel / per-family tables,
    heads-GEMM shapes, the compulsory-byte detail, long notes) goes to the side file `detail_path`."""
    o = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline", "dtype", "data")}
    o["value"], o["ms_per_step"] = _r(full.get("value"), 6), _r(full.get("ms_per_step"), 6)
    c = full.get("config") or {}
    o["config"] = _pick(c, ("workload", "scenes_per_gpu", "global_batch", "points", "voxels", "raw_proposals", "proposals_per_scene",
                            "parallelism", "precision", "setup", "input_prefetch", "world", "grad_sync", "launched_by", "per_rank_ms_per_step"))
    rf = full.get("roofline")
    if rf:
        o["roofline"] = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                   "algorithmic_bytes_per_launch", "avg_launch_us", "launches_per_step", "ms_per_step",
                                   "share_of_step", "traffic_stale", "timing"))
        # the three most expensive instrumented kernels (any family), short form (the full tables: side file)
        pk = rf.get("per_kernel") or {}
        o["roofline"]["top_kernels"] = [dict(kernel=k, **_pick(v, ("frac", "avg_launch_us", "launches_per_step", "traffic_over_algorithmic"), 3))
                                          for k, v in list(pk.items())[:3]]
    else:
        o["roofline"] = None
    sr = full.get("step_roofline")
    if sr:
        o["step_roofline"] = _pick(sr, ("bound", "frac", "achieved", "peak", "unit", "compulsory_bytes_per_step"))
    cb = full.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "error"))
        if cb.get("host"):
            o["cpu_baseline"]["host"] = _pick(cb["host"], ("model", "physical_cores", "threads"))
        if cb.get("thread_sweep"):
            o["cpu_baseline"]["thread_sweep"] = [[t["threads"], _r(t["value"], 3)] for t in cb["thread_sweep"]]
    if full.get("fp32_exact"):
        o["fp32_exact"] = _pick(full["fp32_exact"], ("value", "ms_per_step", "unit"))
    ce = full.get("strong_scaling_ceiling")
    if ce:
        o["strong_scaling_ceiling"] = _pick(ce, ("ratio", "t_8_scenes_ms", "t_1_scene_ms", "ratio_32", "t_32_scenes_ms", "ratio_16", "t_16_scenes_ms", "t_4_scenes_ms", "error", "error_32", "error_16"))
    o["final_loss"] = _r(full.get("final_loss"), 6)
    o["eval_program"] = full.get("eval_program")
    o["detail"] = detail_path
    line = json.dumps(o)
    if len(line) > LINE_BUDGET:          # never lose the line to the tail limit: drop the optional parts, longest first
        for k in ("top_kernels",):
            if o.get("roofline"):
                o["roofline"].pop(k, None)
        for k in ("eval_program", "strong_scaling_ceiling", "fp32_exact"):
            if len(json.dumps(o)) > LINE_BUDGET:
                o.pop(k, None)
        if len(json.dumps(o)) > LINE_BUDGET:
            o["config"] = _pick(o["config"], ("workload", "scenes_per_gpu", "global_batch", "voxels", "parallelism"))
            o["config"]["workload"] = str(o["config"].get("workload"))[:200]
    return o


def cpu_baseline_child(config, threads=0):
    """`bench.py --cpu-baseline-only`: the oracle (CPU restatement of the reference step) timed on this host, on a bounded
    sample of the same workload: forward + loss + backward + AdamW step.  Never touches the GPU.  Prints one JSON object."""
    import numpy as np
    import torch
    from d3net_amd import synthetic as S
    from d3net_amd.config import default_conf
    from oracle import pg_oracle as pg
    from oracle.pointgroup_oracle import PointGroupOracle
    cores = min(os.cpu_count() or 1, threads or CPU_THREADS)
    torch.set_num_threads(cores)
    cfg = default_conf(CONF[config])
    torch.manual_seed(cfg.general.manual_seed)
    scenes = make_scenes(config, 0)[:1]          # ONE scene of the step's batch (the bounded sample)
    b = S.collate(scenes)
    vl, p2v, v2p = pg.voxelization_idx(b["locs_scaled"], 1, 4)   # loader-side work, not timed (as on the GPU)
    b["voxel_locs"], b["p2v_map"], b["v2p_map"] = vl, p2v, v2p
    cpu = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
    if config == "detector":
        from d3net_amd.pointgroup import PointGroup
        state = PointGroup(cfg).state_dict()          # same random init as the GPU run (CPU tensors)
        spk = None
    else:
        from d3net_amd.pipeline import PipelineNet
        net = PipelineNet(cfg, make_dataset(1, cfg.data.num_des_per_scene, False))
        state = net.detector.state_dict()
        spk = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and k != "caption.embeddings")
               for k, v in net.speaker.state_dict().items()} if config in ("speaker", "joint") else None
        lang = S.add_language({k: v for k, v in cpu.items()}, torch.device("cpu"), chunk=cfg.data.num_des_per_scene, vocab=VOCAB)
        lang["lang_len"] = lang["spk_lang_len"]
    orc = PointGroupOracle(cfg, state)
    orc.teacher = True
    leaves = [v for v in orc.p.values() if v.requires_grad] + ([v for v in spk.values() if v.requires_grad] if spk is not None else [])
    opt = torch.optim.AdamW(leaves, lr=cfg.train.optim.lr, weight_decay=cfg.train.optim.weight_decay)   # (model/pipeline.py:738-757)
    what = "detector" if spk is None else "detector + relation graph + captioner (XE)"

    def one_step():
        opt.zero_grad(set_to_none=True)
        d = orc.loss(orc.feed(cpu, 0))
        loss = d["total_loss"]
        if spk is not None:   # relation graph + top-down captioner (teacher forcing) + cross-entropy, oracle/speaker_oracle.py
            import torch.nn.functional as F
            from oracle import speaker_oracle as spo
            d.update({k: v for k, v in lang.items() if k not in d})
            d["lang_len"] = lang["lang_len"]
            g = spo.graph_module({k[len("graph."):]: v for k, v in spk.items() if k.startswith("graph.")}, d, cfg.model.num_graph_steps,
                                 cfg.model.num_locals)
            d.update(g)
            cp = {k[len("caption."):]: v for k, v in spk.items() if k.startswith("caption.")}
            out = spo.forward_sample_batch(cp, d, cfg, cfg.model.max_num_proposal, cfg.model.num_locals)
            logits = out["lang_cap"]
            tgt = d["lang_ids"].reshape(-1, cfg.data.max_spk_len + 2)[:, 1:logits.shape[1] + 1]
            good = out["good"]
            if bool(good.any()):
                loss = loss + F.cross_entropy(logits[good].reshape(-1, logits.shape[-1]), tgt[good].reshape(-1), ignore_index=0)
        loss.backward()
        opt.step()

    # steady state (BASELINE.md section 2 / VERDICT r3): untimed warm-up step(s), then the timed steps
    for _ in range(CPU_WARMUP_STEPS):
        one_step()
    t0 = time.time()
    for _ in range(CPU_TIMED_STEPS):
        one_step()
    dt = (time.time() - t0) / CPU_TIMED_STEPS
    print(json.dumps({"value": 1.0 / dt, "unit": "scenes/sec", "cores": cores, "kind": "port",
                      "sample": "1 scene (of the step's %d; %d points) x (%d warm-up + %d timed) steps, forward+loss+backward+AdamW: %s "
                                "through oracle/ (torch-CPU gather-mm sparse conv and the reference's brute-force ball query on %d "
                                "threads; BFS / segment ops single-threaded): %.1f s per step.  Deviation from BASELINE.md section 2 (>= 3 warm-up "
                                "+ >= 10 timed steps at os.cpu_count() threads): a bounded sample (the default run must finish within minutes; a "
                                "step is seconds) at the best thread count of the sweep 8 / 16 / 32 (256 hardware threads: ~1000x slower)" %
                                (1 if config == "detector" else 4, cpu["locs"].shape[0], CPU_WARMUP_STEPS, CPU_TIMED_STEPS, what, cores, dt)}), flush=True)


def cpu_baseline(config, limit_s=420):
    """run the baseline in child processes (bounded; it must never take the GPU number down with it): one per thread count
    of CPU_THREAD_SWEEP, the fastest is reported, all are listed; the host's CPU model / physical cores come from lscpu"""
    import subprocess
    best, sweep, err = None, [], None
    t_start = time.time()
    for th in [t for t in CPU_THREAD_SWEEP if t <= (os.cpu_count() or 1)] or [os.cpu_count() or 1]:
        left = limit_s - (time.time() - t_start)
        if left < 30:
            break
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", config, "--cpu-threads", str(th)],
                               capture_output=True, text=True, timeout=left,
                               env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(th)))
            line = next((l for l in reversed(r.stdout.strip().splitlines()) if l.startswith("{")), None)
            if line is None:
                err = (r.stderr or r.stdout)[-300:]
                continue
            res = json.loads(line)
            sweep.append({"threads": th, "value": res["value"]})
            if best is None or res["value"] > best["value"]:
                best = res
        except subprocess.TimeoutExpired:
            err = "cpu baseline exceeded %d s" % limit_s
            break
    if best is None:
        return {"value": None, "error": err}
    best["thread_sweep"] = sweep
    best["host"] = cpu_info()
    return best


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as ONE child process group (torch.distributed.run,
    rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Runs before this process imports torch.cuda / touches HIP, and
    the ranks are children (never an exec of a process that initialised the GPU)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    env = dict(os.environ, D3_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (RCCL / dmabuf IPC on this pool)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
    if line is not None:
        out = json.loads(line)
        out.setdefault("config", {})["launched_by"] = "bench.py itself: %d child ranks via torch.distributed.run, rc %d" % (args.gpus, r.returncode)
        print(json.dumps(out), flush=True)
    else:
        sys.stdout.write(r.stdout)
    return r.returncode


def main():
    args = parse()
    if os.environ.get("D3_NO_COREDUMP") == "1":          # (side-measurement children: a fault must not leave a multi-GB core file)
        try:
            import resource
            resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
        except Exception:
            pass
    if args.cpu_baseline_only:
        return cpu_baseline_child(args.config, args.cpu_threads)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # D3_DIST_WORLD1=1 (test switch, like D3_DIST_BACKEND / D3_SHARE_DEVICE): ONE rank still initialises the process group and
    # runs the gradient reducer -- the RCCL plumbing (in-place collectives on the flat buffers, AVG, async handles started inside
    # backward) on a one-GPU box; a world of one averages nothing, so the loss must equal the plain run's
    dist_on = world > 1 or os.environ.get("D3_DIST_WORLD1") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        # RCCL over xGMI ("nccl" on ROCm); D3_DIST_BACKEND=gloo + D3_SHARE_DEVICE=1 is a plumbing test of the N>1 path on a
        # one-GPU box (all ranks on cuda:0), not a benchmark configuration
        dist.init_process_group(os.environ.get("D3_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if os.environ.get("D3_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from d3net_amd import _lib, minkowski as ME, synthetic as S
    from d3net_amd.config import default_conf
    from d3net_amd.distributed import BucketGradAllReduce, broadcast_module
    from d3net_amd.optim import FusedAdamW

    config = args.config
    cfg = default_conf(CONF[config])
    torch.manual_seed(cfg.general.manual_seed)
    scene_ids = None
    if args.scaling == "strong":     # the GLOBAL batch is fixed (8 scenes); rank r steps scenes r, r + W, ...
        assert config != "detector" and STRONG_GLOBAL_BATCH % world == 0, "strong scaling: 8 scenes over 1 / 2 / 4 / 8 ranks (not --config detector)"
        scene_ids = list(range(rank, STRONG_GLOBAL_BATCH, world))
    if args.scene_count:
        scene_ids = list(range(args.scene_count))
    scenes = make_scenes(config, rank, args.small, scene_ids)
    n_scenes = len(scenes)
    chunk = cfg.data.num_des_per_scene
    if config == "detector":
        from d3net_amd.pointgroup import PointGroup
        model = PointGroup(cfg).to(dev).train()
        detector = model
    else:
        from d3net_amd.pipeline import PipelineNet
        model = PipelineNet(cfg, make_dataset(n_scenes, chunk, config == "joint")).to(dev).train()
        detector = model.detector
    detector.teacher = not args.no_teacher
    params = [p for p in model.parameters() if p.requires_grad]
    opt = FusedAdamW(params, lr=cfg.train.optim.lr, weight_decay=cfg.train.optim.weight_decay)
    opt.register_step_pre_hook(lambda *a: detector.drop_stale_grads())
    if dist_on:  # identical replicas
        broadcast_module(model)
    # the executors' flat gradient buffers are all-reduced in place (RCCL, sum -> mean); the other parameters share one
    # packed collective; the bucket layout is static (identical on every rank whatever its scenes produce)
    # -- and the speaker / listener heads' bucket starts from inside the backward, as soon as it crosses into the detector
    grad_sync = None
    if dist_on:
        det_ids = {id(p) for p in detector.parameters()}
        grad_sync = BucketGradAllReduce(params, detector, early=[p for p in params if id(p) not in det_ids])
        if model is not detector:
            model.grad_boundary = grad_sync

    batch = S.make_batch(scenes, dev)
    if config != "detector":
        batch = S.add_language(batch, dev, chunk=chunk, vocab=VOCAB)
        if config in ("speaker", "joint"):
            batch["lang_len"] = batch["spk_lang_len"]     # the speaker's lang_len is the caption length (+2)
        if config == "joint":                               # second (listener) batch of the joint step (pipeline.py:229-274)
            lis = S.add_language(S.make_batch(scenes, dev), dev, chunk=chunk, vocab=VOCAB, seed=9)
    n_points, n_voxels = int(batch["locs"].shape[0]), int(batch["voxel_locs"].shape[0])

    # the batch source: the step's batch + the NEXT step's input stage (voxel features, the backbone's coordinate maps: nothing a
    # parameter touches) started on a side stream during this step -- K such builds inside the K timed steps (d3net_amd.pointgroup)
    from d3net_amd.pointgroup import InputPrefetcher
    from d3net_amd import pointgroup as PG_MOD
    if args.no_prefetch:
        PG_MOD.PREFETCH_MODE = 0
    feeder = InputPrefetcher(detector, (lambda: [dict(batch), dict(lis)]) if config == "joint" else (lambda: dict(batch)))

    def step():
        model.zero_grad(set_to_none=True)
        loss, d = model.training_step(feeder.next())
        if config == "joint":
            d = d["speaker"]
        loss.backward()
        if grad_sync is not None:   # gradient all-reduce over RCCL (sum -> mean), gradients only
            grad_sync()
        opt.step()
        return loss, d

    if args.exact:
        ME.set_exact(True)
    L = _lib.lib()
    # set-up, not a step of the run: one dry pass sizes the cached workspaces (the clustering scratch is several GB) and
    # loads the code objects, so that a run with a very small --warmup does not time one-off allocations ("setup")
    loss, d = step()
    torch.cuda.synchronize()
    # ... and a process on a fresh box needs more than a handful of steps to reach its sustained rate (measured on fresh MI355X boxes:
    # first run 20.8 ms per step with 5 warm-up steps, 18.9 with 60, 18.4 for any later process -- clocks, allocator growth, code
    # objects): SETTLE_STEPS untimed steps (~1 s), independent of --warmup, before the contract's W warm-up steps
    settle_steps = max(0, args.settle)   # (a fixed count: every rank issues the same collectives)
    for _ in range(settle_steps):
        loss, d = step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        loss, d = step()
    torch.cuda.synchronize()
    # the interpreter's cyclic collector: a full (generation-2) pass every ~26 steps walks the whole module / tensor heap
    # (2-7 ms each, tools/step_jitter.py); freezing the long-lived objects after warm-up keeps those passes short
    # compulsory bytes of a step from this batch's own coordinate pyramids: one extra untimed step that keeps
    # the level row counts and counts the kernel-map pairs
    for ex in detector._execs.values():
        if ex is not None:
            ex.debug_keep = True
    step(); torch.cuda.synchronize()
    comp_bytes, comp_detail = compulsory_bytes(detector, batch)
    for ex in detector._execs.values():
        if ex is not None:
            ex.debug_keep, ex.debug_last = False, None
    if config == "joint":
        comp_bytes *= 2          # two detector passes per step
    import gc
    gc.collect()
    gc.freeze()
    if dist_on:
        dist.barrier()
    PROF_STRIDE = 13   # every 13th convolution launch is bracketed by HIP events (coprime with the launches per step)
    L.d3_prof_enable(PROF_STRIDE)
    torch.cuda.synchronize()
    host_prof = None
    if os.environ.get("D3_BENCH_CPROFILE"):   # (diagnostics: host-side cProfile of the timed steps; the value is then not a clean number)
        import cProfile
        host_prof = cProfile.Profile()
        host_prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, d = step()
    torch.cuda.synchronize()
    if host_prof is not None:
        import pstats
        host_prof.disable()
        with open(os.environ["D3_BENCH_CPROFILE"], "w") as f:
            pstats.Stats(host_prof, stream=f).sort_stats("cumulative").print_stats(60)
            pstats.Stats(host_prof, stream=f).sort_stats("tottime").print_stats(40)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms = None
    if dist_on:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        tl = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(tl, t)                      # every rank's own clock around the same K steps (the line reports the MAX)
        per_rank_ms = [round(1e3 * float(x.item()) / args.steps, 3) for x in tl]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel launch durations measured with HIP events on the launch's own stream during the timed region, priced with
    # SURVEY.md 8(d)'s algorithmic bytes / flops (the kernel maps' pair counts are this step's own)
    pairs27 = {}
    for dname, dd in comp_detail.items():
        if isinstance(dd, dict) and "pairs27" in dd:
            for r_, p_ in zip(dd["rows"], dd["pairs27"]):
                pairs27[int(r_)] = int(p_)
    kernels = collect_kernel_rooflines(L, args.steps, PROF_STRIDE, pairs27)
    hg_shapes = kernels.pop("__hg_shapes__", [])
    L.d3_prof_enable(0)
    final_loss = float(loss.detach())

    # reference precision beside the bf16 number: a few untimed-for-`value` steps with the exact-fp32 kernels
    fp32 = None
    if args.exact:
        ME.set_exact(False)
    if not dist_on and not args.no_fp32 and not args.exact:
        ME.set_exact(True)
        try:
            for _ in range(6):                            # (the fp32 program's own plan / arena / workspaces / code objects; the caching
                step()                                    # allocator needs a few steps to stop growing)
            torch.cuda.synchronize()
            k = 10
            t1 = time.perf_counter()
            for _ in range(k):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            fp32 = {"value": n_scenes * k / dt, "unit": "scenes/sec", "ms_per_step": 1e3 * dt / k, "steps": k,
                    "note": "same step with minkowski.set_exact(True): the reference's precision -- fp32 storage, exact fp32 products on "
                            "v_mfma_f32_16x16x4_f32 (D3_CONV_F32 kernels), through the native executor's fp32 program"}
        finally:
            ME.set_exact(False)

    # strong-scaling ceiling of ONE GPU's step: the 8-scene global batch of `--scaling strong` on this GPU
    # against one scene per step -- what 8 ranks with 1 scene each can gain at best before any link is involved
    ceiling = None
    if not dist_on and not args.no_ceiling and not args.exact and config in ("speaker", "listener") and args.scaling == "weak" and not args.small:
        try:
            def timed(scene_list, k=10, warm=6):
                b = S.make_batch(scene_list, dev)
                b = S.add_language(b, dev, chunk=chunk, vocab=VOCAB)
                if config in ("speaker", "joint"):
                    b["lang_len"] = b["spk_lang_len"]
                l2 = S.add_language(S.make_batch(scene_list, dev), dev, chunk=chunk, vocab=VOCAB, seed=9) if config == "joint" else None

                def one():
                    model.zero_grad(set_to_none=True)
                    loss_, _ = model.training_step([dict(b), dict(l2)] if config == "joint" else dict(b))
                    loss_.backward()
                    opt.step()
                for _ in range(warm):
                    one()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    one()
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t1) / k
            sc8 = make_scenes(config, 0, False, list(range(STRONG_GLOBAL_BATCH)))
            t8 = timed(sc8)
            t1s = timed(sc8[:1])
            ceiling = {"t_8_scenes_ms": t8, "t_1_scene_ms": t1s, "ratio": t8 / t1s,
                       "what": "ms per step of this config with the strong-scaling global batch (8 scenes) on ONE GPU / with 1 scene: the best "
                               "speed-up 8 ranks x 1 scene can reach before communication (10 timed steps each after 6 warm-up steps)"}
            # ... and for the 32-scene global batch of the weak-scaling default (4 scenes per rank x 8 ranks): t(32 scenes) / t(4 scenes).
            # The 32-scene step runs in a CHILD process (5.2 M voxels in one batch is beyond anything the suite covers: a fault there
            # must not take this line down); the child is this same program with --scene-count 32
            t4 = timed(sc8[:4])
            ceiling["t_4_scenes_ms"] = t4
            import subprocess
            torch.cuda.empty_cache()
            # 32 scenes in ONE batch may not be representable at all: in this workload (exact "teacher" offsets) every instance
            # collapses onto its centre, the ball-query lists are all capped at 1000 entries and nActive = ~1000 x the object points
            # passes the reference's int range (bfs_cluster.cpp: `int nActive`) -- the library then reports D3_ERR_RANGE.  The child
            # falls back to 16 scenes (global batch of 4 ranks x 4 scenes) and the line says which batch it measured.
            for n32 in (32, 16):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", config, "--scene-count", str(n32), "--steps", "5",
                                    "--warmup", "2", "--settle", "3", "--no-cpu-baseline", "--no-fp32", "--no-ceiling"], capture_output=True,
                                   text=True, timeout=400, env=dict(os.environ, D3_NO_COREDUMP="1"))
                cl = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
                if cl is not None:
                    ceiling["t_%d_scenes_ms" % n32] = json.loads(cl)["ms_per_step"]
                    ceiling["ratio_%d" % n32] = ceiling["t_%d_scenes_ms" % n32] / t4
                    break
                err = r.stderr or ""
                ceiling["error_%d" % n32] = ("n/a: nActive > INT_MAX under teacher offsets (D3_ERR_RANGE, the reference's `int nActive`)"
                                             if ("D3_ERR_RANGE" in err or "-2" in err[-400:]) else "n/a: child failed (%s)" % " ".join(err.split())[-60:])
        except Exception as e:      # (never lose the bench line over the side measurement)
            ceiling = dict(ceiling or {}, error=repr(e)[:200])

    if rank == 0:
        # HBM traffic per launch from the PMC counters (cannot be sampled from inside this process): the committed rocprofv3 --pmc
        # measurement of this same command (tools/gpu_round.sh -> tools/pmc_traffic.py), per kernel instance
        traffic_table, traffic_src, traffic_stale = {}, None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pmc = pmc.get(config, pmc)
            traffic_table = {k: v for k, v in pmc.items() if isinstance(v, dict)}
            measured_on = pmc.get("code_sha")
            traffic_stale = measured_on != code_sha()      # measured on another state of bench.py / csrc: quoted, but flagged
            traffic_src = ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE of this command in separate passes; "
                           "(2*FETCH+WRITE)*1024 per MI355X_MICROARCH.md); measured on code %s, this is %s" % (measured_on, code_sha()))
        except Exception:
            pass
        # families: all template instances of a kernel function (what a profile reader calls "the kernel"); the dominant one is the
        # family with the most time per step, priced as a whole (sum of its launches' algorithmic bytes / flops over the sum of
        # their durations); its instances follow in `per_kernel`
        fam_rec = {}
        for k, r in kernels.items():
            f = fam_rec.setdefault(kernel_family(k), dict(family=r["family"], n=0.0, ms=0.0, b8=0.0, bd=0.0, fl=0.0, samp=0))
            w = r["launches_per_step"]
            f["n"] += w; f["ms"] += r["ms_per_step"]; f["samp"] += r["launches_sampled"]
            f["b8"] += r["algorithmic_bytes_per_launch"] * w; f["bd"] += r["bytes_moved_by_design_per_launch"] * w; f["fl"] += r["flops_per_launch"] * w
        families = {}
        for k, f in fam_rec.items():
            n = max(f["n"], 1e-9)
            families[k] = {"family": f["family"], "launches_sampled": f["samp"], "launches_per_step": f["n"], "avg_launch_us": 1e3 * f["ms"] / n,
                           "ms_per_step": f["ms"], "algorithmic_bytes_per_launch": f["b8"] / n, "bytes_moved_by_design_per_launch": f["bd"] / n,
                           "flops_per_launch": f["fl"] / n, "achieved_gbs": f["b8"] / max(f["ms"], 1e-9) / 1e6,
                           "achieved_tflops": f["fl"] / max(f["ms"], 1e-9) / 1e9}
        dom = max(families, key=lambda k: families[k]["ms_per_step"]) if families else None
        workload = {
            "speaker": "BASELINE configs[2]: PipelineNet mode 1 (PointGroup detector -> relation graph -> top-down captioner, "
                       "XE), conf/pointgroup_captioning.yaml: %d scenes/GPU/step (40-box synthetic ScanNet scenes, 200x150x100 "
                       "@ 2 cm), %d descriptions/scene, V=%d, teacher clustering, AdamW" % (n_scenes, chunk, VOCAB),
            "detector": "BASELINE configs[1]: PointGroup detector only, canonical synthetic ScanNet scene (200x150x100 @ 2 cm), "
                        "1 scene/GPU/step, m=16, 7-level U-Net, teacher clustering, AdamW",
            "listener": "BASELINE configs[3]: PipelineNet mode 2 (detector -> GRU language encoder -> transformer match), "
                        "conf/pointgroup_grounding.yaml: %d scenes/GPU/step, %d descriptions/scene (T=128)" % (n_scenes, chunk),
            "joint": "BASELINE configs[4]: PipelineNet mode 3 (self-critical speaker-listener, beam 3 / top-3, CIDEr reward), "
                     "conf/pointgroup_joint.yaml: 2 x %d scenes/GPU/step" % n_scenes}[config]
        out = {
            "metric": METRIC[config], "value": world * n_scenes * args.steps / elapsed,
            "unit": "scenes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32" if args.exact else "bf16", "data": "synthetic",
            "config": {"workload": workload + " (%d voxels, %d points, 134 ch per step)" % (n_voxels, n_points),
                       "scenes_per_gpu": n_scenes, "global_batch": world * n_scenes, "points": n_points, "voxels": n_voxels,
                       "raw_proposals": int(d.get("num_raw_proposals", 0)),
                       "proposals_per_scene": float(d["proposal_batch_mask"].sum() / n_scenes) if "proposal_batch_mask" in d else None,
                       "parallelism": "scene-parallel dp%d" % world,
                       "world": {"size_seen_by_process_group": dist.get_world_size() if dist_on else 1,
                                 "backend": dist.get_backend() if dist_on else None,
                                 "scaling": "weak: %d scenes per rank per step" % n_scenes if args.scaling == "weak" else
                                            "strong: global batch fixed at %d scenes, %d per rank" % (STRONG_GLOBAL_BATCH, n_scenes)},
                       "per_rank_ms_per_step": per_rank_ms,
                       "precision": ("fp32 storage, exact fp32 products on v_mfma_f32_16x16x4_f32, fp32 accumulate (the reference's precision)" if args.exact else
                                     "fp32 residual stream, bf16 BN->ReLU activations and MFMA operands, fp32 accumulate; heads fp32"
                                     + ("; BASELINE configs[4] names fp16: bf16 operands here (same 16-bit MFMA rate on CDNA4, fp32's exponent range, "
                                        "no loss scaling; the reference itself trains fp32)" if config == "joint" else "")),
                       "setup": "1 untimed dry-run step (workspace allocation, code-object loads) + %d untimed settle steps (a fresh "
                                "box reaches its sustained rate only after some tens of steps) before the --warmup steps" % settle_steps,
                       "input_prefetch": ("mode %d: the next step's parameter-free input stage is built on a side stream inside the step "
                                          "(K builds in the K timed steps)" % PG_MOD.PREFETCH_MODE) if PG_MOD.PREFETCH_MODE else "off"},
            "final_loss": final_loss, "fp32_exact": fp32, "strong_scaling_ceiling": ceiling,
            "eval_program": "value = the bf16 TRAINING step; eval()/mAP/CIDEr run another program (fp32 twin executors, minkowski.exact_for; DESIGN 5.1)",
            "metric_parity": {"policy": "training steps (this line's value) run bf16 MFMA operands; evaluation -- every mAP / CIDEr the library reports -- runs "
                                        "the reference-precision kernels (d3net_amd/minkowski.py exact_for; DESIGN.md 5.1)",
                              "asserted": "tests/test_metric_parity_gpu.py: 128 held-out scenes x 3 training seeds, evaluation path within 0.5 % of the fp32 CPU "
                                          "oracle on mAP@0.5 and CIDEr@0.5IoU; bf16 kernels forced onto evaluation: measured +0.04 / -0.21 / -0.19 / -0.12 / -1.70 / +0.12 % CIDEr over six trained models "
                                          "(reported, bound 3 %)",
                              "same_step_at_reference_precision": "fp32_exact"},
        }
        if dom is not None:
            rf = roofline_object(dom, families[dom], traffic_table, PROF_STRIDE)
            rf["traffic_source"], rf["traffic_stale"] = traffic_src, traffic_stale
            rf["share_of_step"] = families[dom]["ms_per_step"] / (1e3 * elapsed / args.steps)
            rf["dominant"] = ("the instrumented kernel function (all template instances of it; rocprofv3 lists the instances separately) with the "
                              "most time per step; `per_kernel`: every instance of every instrumented kernel under its rocprofv3 name, `families`: "
                              "every kernel function -- same pricing, most expensive first")
            keys = ("bound", "achieved", "peak", "unit", "frac", "hbm_frac", "mfma_frac", "traffic", "traffic_over_algorithmic",
                    "algorithmic_bytes_per_launch", "launches_per_step", "avg_launch_us", "ms_per_step")
            order = sorted(kernels, key=lambda k: -kernels[k]["ms_per_step"])
            rf["per_kernel"] = {k: {kk: vv for kk, vv in roofline_object(k, kernels[k], traffic_table, PROF_STRIDE).items() if kk in keys} for k in order[:24]}
            rf["families"] = {k: {kk: vv for kk, vv in roofline_object(k, families[k], traffic_table, PROF_STRIDE).items() if kk in keys}
                              for k in sorted(families, key=lambda k: -families[k]["ms_per_step"])}
            rf["hg_gemm_shapes"] = hg_shapes       # the heads' GEMM shapes that cost the most time (largest problem of a batched launch)
            out["roofline"] = rf
        else:
            out["roofline"] = None
        step_ms = 1e3 * elapsed / args.steps
        out["step_roofline"] = {"bound": "hbm", "compulsory_bytes_per_step": comp_bytes, "achieved": comp_bytes / (step_ms * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": comp_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "what": "SURVEY.md 8(d) compulsory traffic of the step's two sparse U-Nets (forward + 2x backward, bf16 "
                                        "storage, one int32 pair per kernel-map entry) + input pooling, over the whole step time "
                                        "(clustering, heads, captioner and optimizer included in the time, not in the bytes)",
                                "detail": comp_detail}
        if grad_sync is not None:
            items = grad_sync._items()
            out["config"]["grad_sync"] = {"collectives_per_step": 1 + sum(len(it["ranges"]) for it in items) + (1 if grad_sync.early else 0),
                                          "ranks_seen_by_backend": dist.get_world_size(), "backend": dist.get_backend(),
                                          # in schedule order: heads bucket, executor chunks, packed rest
                                          "bytes_per_collective": ([4 * sum(p.numel() for p in grad_sync.early)] if grad_sync.early else []) +
                                                                  [4 * (hi - lo) for it in items for lo, hi in it["ranges"]] +
                                                                  [4 * sum(p.numel() for p in (grad_sync._rest or []))],
                                          "heads_bucket_floats": sum(p.numel() for p in grad_sync.early),
                                          "heads_bucket_started_inside_backward": grad_sync.early_launches,
                                          "executor_chunks": [[hi - lo for lo, hi in it["ranges"]] for it in items],
                                          "executor_chunk_collectives_started_inside_backward": grad_sync.chunk_launches}
        if not dist_on and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(config)
    # The JSON line must be the LAST thing on stdout: RCCL writes a five-line version banner through C stdio at communicator
    # set-up, and with stdout on a pipe that text sits in the C buffer of EVERY rank until the process exits -- after the line.
    # So: every rank flushes its C buffers, all ranks meet, the process group goes away, and only then rank 0 prints.
    if dist_on:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if rank == 0:
        # full tables -> side file; the line itself stays under LINE_BUDGET bytes (the driver keeps only the tail of stdout)
        detail_path = None
        try:
            ddir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(ddir, exist_ok=True)
            detail_path = os.path.join("gpurun_out", "bench_detail_%s%s.json" % (config, "_exact" if args.exact else ""))
            with open(os.path.join(ROOT, detail_path), "w") as f:
                json.dump(out, f)
        except Exception:
            detail_path = None
        print(json.dumps(compact_line(out, detail_path)), flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
