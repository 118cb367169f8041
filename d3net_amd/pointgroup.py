"""PointGroup detector on MI355X: the reference's `model.pointgroup.PointGroup` with the same constructor,
methods (`feed`, `forward`, `parse_feed_ret`, `loss`, `clusters_voxelization`, `convert_stack_to_batch`,
`get_object_assignments`, `get_batch_offsets`), `data_dict` keys and state-dict layout
(reference: model/pointgroup.py:24-541; key flow in SURVEY.md Appendix A), rebuilt so that the whole
forward stays on the device:

  reference (model/pointgroup.py)                          here
  ----------------------------------------------------    ---------------------------------------------
  :112-122 python loop + .sum() sync per scene             bincount + cumsum on the device
  :296-305 ball query on GPU, D2H of the lists, CPU BFS    HIP ball query + HIP clustering, no D2H
  :166-169 D2H of cluster coords, CPU hash voxelisation    HIP voxelization_idx on the device
  :342-344 python loop over proposals, O(P*S) on the CPU   proposals_offset[1:] - proposals_offset[:-1]
  :233-235 numpy box corners on the CPU                    fp64 corner arithmetic on the device
Host RNG draws the reference makes (`torch.rand(3)` x2 at :161, `torch.randperm(128)` at :251) are drawn
from the same CPU generator in the same order, so a seeded run is comparable with the oracle.
"""
import functools
import threading

import numpy as np
import torch
import torch.nn as nn

from . import common
from . import heads
from . import minkowski as ME
from . import nativelinear
from . import netexec
from . import pointgroup_ops


class _HostStage:
    """Small host -> device transfers without draining the stream: a pageable `.to(device)` blocks the host until
    every kernel already queued has run (the copy is stream ordered), which costs the run-ahead of the whole step for
    a 12-byte vector.  A ring of pinned buffers + asynchronous copies keeps the host ahead; the ring is deeper than the
    number of steps the host can run ahead (every step has blocking count phases)."""

    def __init__(self, slots=64, nbytes=4096):
        self.slots, self.nbytes, self.ring, self.i = slots, nbytes, None, 0

    def put(self, t, device):
        if self.ring is None:
            self.ring = [torch.empty(self.nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.slots)]
        t = t.contiguous()
        n = t.numel() * t.element_size()
        if n > self.nbytes:
            return t.to(device)
        pin = self.ring[self.i][:n].view(t.dtype).view(t.shape)
        self.i = (self.i + 1) % self.slots
        pin.copy_(t)
        return torch.empty(t.shape, dtype=t.dtype, device=device).copy_(pin, non_blocking=True)


_STAGE = _HostStage()
_CONST = {}


def _const(key, device, build):
    """device-resident constants (one upload per device instead of one per step)"""
    k = (key, device.index)
    if k not in _CONST:
        _CONST[k] = build().to(device)
    return _CONST[k]


_WORKER = None


def _cluster_worker():
    """ONE persistent helper thread per process for the concurrent clustering branch.  A fresh thread per step would
    give the per-thread operator workspaces (pointgroup_ops._workspace, up to 0.5 GB for the ball-query stash) a new
    key whenever the OS hands out a new thread id, i.e. leak them."""
    global _WORKER
    if _WORKER is None:
        import concurrent.futures
        _WORKER = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="d3-cluster")
    return _WORKER


_PF_WORKER = None


def _prefetch_worker():
    """ONE persistent helper thread per process for the input prefetch (its own per-thread operator workspaces, like the clustering helper)"""
    global _PF_WORKER
    if _PF_WORKER is None:
        import concurrent.futures
        _PF_WORKER = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="d3-prefetch")
    return _PF_WORKER


class _Done:
    """a finished `future` (the thread-less prefetch completes on the step's own thread)"""

    def result(self, timeout=None):
        return None


class _Prefetch:
    """ticket of PointGroup.prefetch(): the weight-independent input stage of a LATER step (voxel features, the backbone's coordinate
    pyramid and kernel maps) built on a side stream while the current step runs"""
    __slots__ = ("inputs", "future", "cm", "voxel_feats", "event", "error")

    def __init__(self, inputs):
        self.inputs, self.future, self.cm, self.voxel_feats, self.event, self.error = inputs, None, None, None, None, None


PREFETCH_MODE = 3   # InputPrefetcher: 0 off; where in the current step the next batch's input stage starts: 1 clustering begins, 2 at once,
                    # 3 the first clustering branch is enqueued (its single-workgroup BFS replay leaves the chip idle: measured best,
                    # speaker step 17.49 -> 16.84 ms, detector 8.30 -> 7.86 ms in-process, gpurun_out/r05_j8), 4 behind ScoreNet, 5 at the captioner


class InputPrefetcher:
    """The training loop's batch source with one batch of look-ahead: `next()` hands out the batch of this step and announces the
    following one to the detector (PointGroup.prefetch), whose input stage then overlaps this step.  `make()` -> a data_dict (or a
    list of them for PipelineNet's joint step: the first one is prefetched)."""

    def __init__(self, detector, make):
        self.detector, self.make, self.ahead = detector, make, None

    def next(self):
        cur = self.ahead if self.ahead is not None else self.make()
        self.ahead = None
        if PREFETCH_MODE:
            self.detector.prefetch_at = {1: "cluster", 2: "start", 3: "bfs", 4: "scorenet", 5: "caption"}[PREFETCH_MODE]
            self.ahead = self.make()
            self.detector.prefetch(self.ahead[0] if isinstance(self.ahead, (list, tuple)) else self.ahead)
        return cur


PREFETCH_TIMEOUT_S = 120
PREFETCH_PADCAST = 1    # (A/B switch: the prefetch stage also prepares the stem's padded bf16 operand)
SELECT_WITH_OFFSETS = 1  # (A/B switch) the object points' batch offsets come out of the cluster_select launch (0: six library launches behind it)
EARLY_POINT_GRADS = 1   # (A/B switch of tools/ab.py; the per-model switch is PointGroup.early_point_grads)
PHASES = None   # tools/phase_times.py: list of (name, cuda event) marks on the current stream when not None


def _mark(name):
    if PHASES is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        PHASES.append((name, ev))


class _PointLossGraft(torch.autograd.Function):
    """The weighted point-level loss with its gradient ALREADY computed (PointGroup._early_point_losses): forward hands out the value,
    backward scales the stored gradients by the incoming one.  inputs: (point features, value, n stored gradients, n tensors they
    belong to: the point features first, then the two heads' parameters)."""

    @staticmethod
    def forward(ctx, value, n, *rest):
        ctx.n = n
        ctx.save_for_backward(*rest[:n])
        return value.clone()

    @staticmethod
    def backward(ctx, go):
        grads = ctx.saved_tensors
        return (None, None) + (None,) * ctx.n + tuple(g * go for g in grads)


class PointGroup(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.task = cfg.general.task
        in_channel = cfg.model.use_color * 3 + cfg.model.use_normal * 3 + cfg.model.use_coords * 3 + \
            cfg.model.use_multiview * 128
        m = cfg.model.m
        D = 3
        classes = cfg.data.classes
        block = common.ResidualBlock if cfg.model.block_residual else common.VGGBlock

        self.requires_gt_mask = cfg.data.requires_gt_mask
        self.cluster_radius = cfg.cluster.cluster_radius
        self.cluster_meanActive = cfg.cluster.cluster_meanActive
        self.cluster_shift_meanActive = cfg.cluster.cluster_shift_meanActive
        self.cluster_npoint_thre = cfg.cluster.cluster_npoint_thre
        self.freeze_backbone = cfg.cluster.freeze_backbone
        self.score_scale = cfg.train.score_scale
        self.score_fullscale = cfg.train.score_fullscale
        self.mode = cfg.train.score_mode
        # True (the reference's data_dict): the proposals that pass the thresholds are compacted into `proposal_feats`,
        # `proposals_batchId`, `proposal_objectness_scores`, `proposal_crop_bbox` -- which needs their COUNT on the host, a round
        # trip in the middle of the step.  False (PipelineNet's training steps, where only the batched tensors are consumed):
        # convert_stack_to_batch takes the uncompacted rows with the batch id of a rejected proposal set to -1; same batched
        # tensors, same gradients, no round trip, and the four compact keys are not produced.
        self.compact_proposals = True
        self.prepare_epochs = cfg.cluster.prepare_epochs
        self.current_epoch = 0

        sp_norm = functools.partial(ME.MinkowskiBatchNorm, eps=1e-4, momentum=0.1)
        norm = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)

        # backbone: stem conv, 7-level U-Net, BN, ReLU                       (reference :69-74)
        self.backbone = nn.Sequential(
            ME.MinkowskiConvolution(in_channel, m, kernel_size=3, bias=False, dimension=D),
            common.UBlock([m * c for c in cfg.model.blocks], sp_norm, cfg.model.block_reps, block),
            sp_norm(m),
            ME.MinkowskiReLU(inplace=True))
        self.sem_seg = nn.Linear(m, classes)                                # (reference :77)
        self.offset_net = nn.Sequential(nn.Linear(m, m), norm(m), nn.ReLU(inplace=True), nn.Linear(m, 3))  # :80-85
        self.score_net = nn.Sequential(                                     # (reference :88-92)
            common.UBlock([m * c for c in cfg.model.cluster_blocks], sp_norm, 2, block),
            sp_norm(m),
            ME.MinkowskiReLU(inplace=True))
        if cfg.model.pred_bbox:
            raise NotImplementedError("pred_bbox=True is not on the hot path (conf/pointgroup.yaml: pred_bbox False)")
        self.score_linear = nn.Linear(m, 1)                                 # (reference :108)
        ME.fuse_bn_relu(self)

        # test hooks: override predictions before clustering ("teacher" switch of SURVEY.md 8(d))
        self.teacher = False
        self.concurrent_clustering = True
        self.padded_list_budget = None      # bytes; None: 35 % of the device's memory (_padded_list_budget)
        self._mem_total = {}
        self._streams = {}
        # native executors (csrc/unet.hip) for the two sparse U-Nets: one C-ABI call per forward / backward instead of
        # one python call per module.  Built lazily; the module tree above stays the owner of every parameter.
        self.native_unet = True
        self.native_exact = True      # minkowski.set_exact(True) also runs through the native executor (False: module by module)
        self.__dict__["_execs"] = {}
        # input prefetch (prefetch()): the pending ticket, where in the step its work is started ("cluster": when the main stream
        # reaches the latency-bound clustering stage; "start": at once), and the two newest consumed tickets (kept alive: their
        # tensors live in the side stream's allocator pool and must not return to it while this step still reads them)
        self.__dict__["_pf_pending"] = None
        self.__dict__["_pf_live"] = [None, None]
        self.__dict__["_pf_inflight"] = None
        self.prefetch_at = "bfs"
        # the semantic / offset losses' BACKWARD (both point heads down to the gradient of the backbone's point features) runs inside
        # forward() as well, right behind the losses themselves: in the clustering stage the chip is mostly idle, in the backward these
        # ~15 launches sit on the critical path between ScoreNet's and the backbone's backward (0.35 ms of the 4-scene step)
        self.early_point_grads = True

    def _padded_list_budget(self, device):
        """bytes the padded ball-query lists + BFS records of one forward may take (pointgroup_ops.padded_clustering_bytes);
        `self.padded_list_budget` overrides the default of 35 % of the device's memory"""
        if self.padded_list_budget is not None:
            return int(self.padded_list_budget)
        key = device.index
        if key not in self._mem_total:
            self._mem_total[key] = int(torch.cuda.get_device_properties(device).total_memory)
        return int(0.35 * self._mem_total[key])

    def _side_stream(self, device):
        key = (device.index, threading.get_ident())
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _exec(self, name, exact=False):
        """the native executor of `name` ("backbone" / "score_net"); exact: its reference-precision twin (fp32 storage,
        fp32 MFMA kernels) -- same parameters, its own program and gradient buffer (minkowski.set_exact)"""
        key = name + "/f32" if exact else name
        ex = self._execs.get(key)
        if ex is None:
            if name == "backbone":
                in_channel = self.backbone[0].in_channels
                ex = netexec.NativeUNet(self.backbone[0], self.backbone[1], self.backbone[2], in_channel, False, exact=exact)
            else:
                ex = netexec.NativeUNet(None, self.score_net[0], self.score_net[1], self.cfg.model.m, True, exact=exact)
            self._execs[key] = ex
        return ex

    def _begin_maps(self, name, voxel_locs):
        """coordinate manager of a U-Net input with its pyramid already enqueued (the native executor's path; None otherwise)"""
        if not (self.native_unet and voxel_locs.is_cuda and voxel_locs.size(0) > 0) or (ME._EXACT and (ME._EXACT_FMA or not self.native_exact)):
            return None
        cm = ME.CoordinateManager(voxel_locs.int().contiguous())
        exact = ME.exact_for(self.training, name)
        if exact:
            cm.want16 = False        # (only the bf16 executors read the 16-bit kernel maps)
        cm.begin_pyramid(self._exec(name, exact=exact).nlevels)
        return cm

    def release_eval_executors(self):
        """free the fp32 twin executors (packed fp32 weights, fp32 arena plan, flat gradient buffer) that evaluation-mode forwards
        instantiate beside the bf16 ones under the precision policy (minkowski.exact_for): call after a validation pass inside a
        training run when the memory matters; the next evaluation forward rebuilds them (one plan + one weight packing)."""
        if ME._EXACT:
            return       # the fp32 executors ARE the training executors in this mode
        for key in [k for k in self._execs if k.endswith("/f32")]:
            del self._execs[key]

    # ------------------------------------------------------------------------------------- input prefetch
    def _input_stage(self, data_dict):
        """voxel features + the backbone's coordinate manager (pyramid enqueued) of one batch -> (cm or None, voxel_feats).
        Nothing here depends on a parameter: model/pointgroup.py:466-474 (`feed`: the input voxelisation) and the coordinate maps
        MinkowskiEngine builds on first use inside the backbone."""
        f = data_dict["feats"]
        # The coordinate pyramid of the backbone needs one host round trip (the level sizes).  Its kernels and the copy of
        # the counts are enqueued BEFORE the input voxelisation, the wait comes after it: the device pools the point features
        # (~0.27 ms for four scenes) while the host reads the counts and enqueues the kernel-map fills.
        cm = self._begin_maps("backbone", data_dict["voxel_locs"])
        if self.cfg.model.use_coords and f.is_cuda and f.dtype == torch.float32 and not f.requires_grad:
            # voxelization(cat(feats, locs)) without the concatenated copy (csrc/voxelize.hip: d3_voxelize_fp2)
            vf = pointgroup_ops.voxelization_cat(f, data_dict["locs"], data_dict["v2p_map"], self.cfg.data.mode)
        else:
            if self.cfg.model.use_coords:
                f = torch.cat((f, data_dict["locs"]), 1)
                data_dict["feats"] = f
            vf = pointgroup_ops.voxelization(f.contiguous(), data_dict["v2p_map"], self.cfg.data.mode)
        return cm, vf

    def prefetch(self, data_dict):
        """Announce the batch of a LATER training step: its input stage (`_input_stage` + every level's kernel map) is built on a side
        stream from a helper thread while the current step runs -- the device-side counterpart of the reference's DataLoader workers,
        which voxelise the next batch on the CPU during the step (data/scannet/dataset.py collate).  The work starts
        when the current step's main stream reaches the clustering stage (single-workgroup BFS levels, union-find: most of the chip
        idle; `prefetch_at`: "cluster" / "bfs" / "scorenet" / "caption" name the point) or, with `prefetch_at = "start"`, at once; `feed()` of that batch waits for it (or runs the stage inline when it never
        started).  Same kernels, same results; one ticket may be pending at a time."""
        v = data_dict.get("voxel_locs")
        f = data_dict.get("feats")
        if not (torch.is_tensor(v) and v.is_cuda and self.native_unet and v.size(0) > 0) or "_prefetch" in data_dict:
            return data_dict
        if not (self.cfg.model.use_coords and torch.is_tensor(f) and f.dtype == torch.float32 and not f.requires_grad):
            return data_dict
        t = _Prefetch({k: data_dict[k] for k in ("feats", "locs", "v2p_map", "voxel_locs")})
        data_dict["_prefetch"] = t
        self._pf_pending = t
        if self.prefetch_at == "start":
            self._kick_prefetch()
        return data_dict

    def _kick_prefetch(self, at=None):
        t = self._pf_pending
        if t is None or t.future is not None or (at is not None and at != self.prefetch_at):
            return
        self._pf_pending = None
        dev = t.inputs["voxel_locs"].device
        cur = torch.cuda.current_stream(dev)
        # every block the side stream's pool holds was last read by work enqueued before this point (the tickets in _pf_live are the
        # only side-pool tensors this stream still reads, and they are released at a later feed()): the helper's first act is to wait here
        gate = torch.cuda.Event()
        gate.record(cur)
        training = self.training
        def work():
            try:
                with torch.cuda.device(dev):
                    key = (dev.index, threading.get_ident())
                    if key not in self._streams:
                        self._streams[key] = torch.cuda.Stream(device=dev)
                    side = self._streams[key]
                    with torch.cuda.stream(side):
                        side.wait_event(gate)
                        cm, vf = self._input_stage(t.inputs)
                        if cm is not None:
                            ex = self._exec("backbone", exact=ME.exact_for(training, "backbone"))
                            ex.maps(cm)   # pyramid counts (host round trip) + all kernel maps
                            if PREFETCH_PADCAST:
                                cm.padded_input = ex.pad_input(vf)      # the stem's zero-padded bf16 operand (the forward's first launch)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    t.cm, t.voxel_feats, t.event = cm, vf, ev
            except BaseException as e:      # re-raised by the consumer
                t.error = e
        t.future = _prefetch_worker().submit(work)
        self._pf_inflight = t.future

    def _take_prefetch(self, t, data_dict):
        """-> (cm, voxel_feats) of ticket t for THIS data_dict, or None (never started, or built from other tensors)"""
        if self._pf_pending is t:
            self._pf_pending = None
        same = t.inputs is not None and all(t.inputs.get(k) is data_dict.get(k) for k in ("feats", "locs", "v2p_map", "voxel_locs"))
        if t.future is None:            # never started (no clustering stage since prefetch()): the caller runs the stage inline
            t.inputs = None
            return None
        try:
            t.future.result(timeout=PREFETCH_TIMEOUT_S)
        except Exception as e:      # (concurrent.futures.TimeoutError: fail loudly instead of hanging the step for ever)
            raise RuntimeError("PointGroup: the input prefetch of this batch did not finish within %d s (helper thread stuck?); "
                               "pointgroup.PREFETCH_MODE = 0 disables the look-ahead" % PREFETCH_TIMEOUT_S) from e
        t.inputs = None
        if t.error is not None:
            raise t.error
        # The ticket before the previous one is released here.  Its blocks go back to the side stream's pool, so no helper whose gate
        # was recorded before that ticket's last use may still be allocating: the newest helper (the only one that can be running) is
        # waited for first -- it finished long ago unless two detector passes per step are prefetched back to back.
        if self._pf_inflight is not None:
            self._pf_inflight.result(timeout=PREFETCH_TIMEOUT_S)
        self._pf_live = [self._pf_live[1], t]
        if not same or t.voxel_feats is None:
            return None
        torch.cuda.current_stream(t.voxel_feats.device).wait_event(t.event)
        return t.cm, t.voxel_feats

    def _run_unet(self, name, module, x):
        """x: ME.SparseTensor -> (M, m) features of `module` (backbone / score_net)"""
        if self.native_unet and not (ME._EXACT and (ME._EXACT_FMA or not self.native_exact)) and x.F.size(0) > 0:
            return self._exec(name, exact=ME.exact_for(self.training, name))(x.F, x.coordinate_manager, self.training)
        return module(x).features

    def static_gradient_buckets(self):
        """[(flat gradient buffer, its parameters, executor)] for BOTH executors, created eagerly so that every rank of
        a data-parallel job has the same bucket layout whatever its scenes produce (d3net_amd.distributed).  The executors are
        the ones the current precision mode runs (`minkowski.set_exact`: the fp32 twins own their own flat buffers); a reducer
        caches the list, so the mode must be chosen before the first gradient sync."""
        out = []
        dev = self.score_linear.weight.device
        for name in ("score_net", "backbone"):      # backward order: ScoreNet's gradients are complete first
            ex = self._exec(name, exact=ME._EXACT)
            ex._param_ptrs()
            ex._grads(dev)
            out.append((ex._flat_grad, ex.owned_params(), ex))
        return out

    def drop_stale_grads(self):
        """call before optimizer.step(): executors whose backward did not run since zero_grad() must not re-apply the
        previous step's gradient (FusedAdamW / torch.optim skip tensors whose grad is None)"""
        for ex in self._execs.values():
            if ex is not None:
                ex.drop_stale_grads()

    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad; gradients owned by the native executors are marked stale instead of being detached
        one by one (the next backward overwrites them)."""
        native = set()
        for ex in self._execs.values():
            if ex is not None and ex._grad_views is not None and set_to_none:
                ex.fresh_grads = True
                native.update(id(p) for p, v in zip(ex.b.params, ex._grad_views) if v is not None and p.grad is v)
        key = (len(native), set_to_none)
        cache = self.__dict__.get("_zg_cache")
        if cache is None or cache[0] != key:   # walking ~500 parameters through nn.Module.parameters() costs ~1 ms per step
            cache = (key, [p for p in self.parameters() if id(p) not in native])
            self.__dict__["_zg_cache"] = cache
        for p in cache[1]:
            if p.grad is None:
                continue
            if set_to_none:
                p.grad = None
            else:
                p.grad.detach_(); p.grad.zero_()

    # ------------------------------------------------------------------------------------- helpers
    @staticmethod
    def get_batch_offsets(batch_idxs, batch_size):
        """(B+1) int32 offsets of the (sorted) batch index column (reference :110-122), without host syncs."""
        # (torch.bincount reads min / max back to the host: two blocking round trips per call)
        ids = _const(("arange_i32", batch_size), batch_idxs.device, lambda: torch.arange(batch_size, dtype=torch.int32))
        counts = (batch_idxs.view(-1, 1) == ids.view(1, -1)).sum(0)
        offsets = torch.zeros(batch_size + 1, dtype=torch.int32, device=batch_idxs.device)
        offsets[1:] = torch.cumsum(counts, 0).int()
        return offsets

    def clusters_voxelization(self, clusters_idx, clusters_offset, feats, coords, fullscale, scale, mode, rand=None):
        """Normalise every cluster into a <= fullscale^3 grid and voxelise it (reference :125-178).
        clusters_idx (S,2) int32, clusters_offset (P+1) int32 -- on the device here.
        rand: optional (2,3) tensor standing in for the two `torch.rand(3)` draws of the reference (:161)."""
        dev = feats.device
        c_idxs = clusters_idx[:, 1].long()
        clusters_feats = heads.gather_cluster_rows(feats, c_idxs)
        _mark("cv_gather")
        # per-cluster mean / extrema of the member coordinates straight from the (cluster, point) pairs, no (S,3) temporaries
        # (csrc/seg_ops.hip: d3_cluster_coords_stats; min(x - mean) == min(x) - mean under monotone rounding)
        clusters_coords_mean, raw_min, raw_max = pointgroup_ops.cluster_coords_stats(coords, clusters_idx, clusters_offset)
        _mark("cv_sec_mean")
        _mark("cv_sec_minmax")
        if rand is None:
            r0, r1 = torch.rand(3), torch.rand(3)   # CPU generator, same order as the reference
        else:
            r0, r1 = rand[0].cpu(), rand[1].cpu()
        # size / centre / grid scale / random placement offset per cluster (:146-165): one launch, bit-equal to the ~30
        # elementwise library launches on (P,3) tensors it replaces (csrc/seg_ops.hip: d3_cluster_norm_params)
        clusters_size, clusters_center, clusters_scale, offset = pointgroup_ops.cluster_norm_params(
            clusters_coords_mean, raw_min, raw_max, fullscale, scale, r0.float(), r1.float())
        # (coords[point] - mean) * scale + offset, truncated (:166), with the cluster id in front: one pass over the S pairs
        clusters_coords = pointgroup_ops.cluster_transform(coords, clusters_idx, clusters_coords_mean, clusters_scale, offset)
        n_clusters = int(clusters_offset.numel() - 1)
        _mark("cv_elementwise")
        voxel_coords, p2v_map, v2p_map = pointgroup_ops.voxelization_idx(clusters_coords, n_clusters, mode)
        _mark("cv_voxelization_idx")
        cm = self._begin_maps("score_net", voxel_coords)      # (as in feed(): the pooling below runs during the round trip)
        voxel_feats = pointgroup_ops.voxelization(clusters_feats, v2p_map, mode)
        voxel_feats = ME.SparseTensor(features=voxel_feats, coordinates=None if cm is not None else voxel_coords.int(), coordinate_manager=cm)
        voxel_feats.v2p_map = v2p_map
        return voxel_feats, p2v_map, (clusters_center, clusters_size)

    def get_object_assignments(self, data_dict):
        """nearest GT centre in L1 for every proposal slot (reference :216-221; lib/utils/nn_distance.py:32-59)."""
        pc1, pc2 = data_dict["proposal_center_batched"], data_dict["center_label"]
        dist = (pc1.unsqueeze(2) - pc2.unsqueeze(1)).abs().sum(-1)
        data_dict["object_assignment"] = dist.min(2)[1]
        return data_dict

    @staticmethod
    def _box_corners(center, size):
        """lib/utils/bbox.py:54-74 (get_3d_box_batch) for heading 0, in fp64 like the numpy original."""
        c = center.double(); s = size.double()
        sgn = _const("corner_signs", c.device, lambda: torch.tensor(
            [[1, 1, -1, -1, 1, 1, -1, -1], [1, -1, -1, 1, 1, -1, -1, 1], [1, 1, 1, 1, -1, -1, -1, -1]], dtype=torch.float64))
        corners = torch.stack([s[:, 0:1] / 2 * sgn[0], s[:, 1:2] / 2 * sgn[1], s[:, 2:3] / 2 * sgn[2]], -1)  # (P,8,3)
        return corners + c.unsqueeze(1)

    def convert_stack_to_batch(self, data_dict, perms=None):
        """stacked proposals -> (B,128,.) padded + shuffled tensors (reference :223-263)."""
        batch_size = len(data_dict["batch_offsets"]) - 1
        K = self.cfg.model.max_num_proposal
        lazy = data_dict.pop("_stb_inputs", None)
        if lazy is not None:
            pf, scores_in, crop, bids_in = lazy
        else:
            pf, scores_in, crop, bids_in = (data_dict["proposal_feats"], data_dict["proposal_objectness_scores"],
                                            data_dict["proposal_crop_bbox"], data_dict["proposals_batchId"])
        dev = pf.device
        keys = ("proposal_feats_batched", "proposal_bbox_batched", "proposal_center_batched", "proposal_sem_cls_batched",
                "proposal_scores_batched", "proposal_batch_mask")
        perm = data_dict.pop("_slot_perm_staged", None) if perms is None else None
        if perm is None:
            perm = torch.stack([torch.randperm(K) if perms is None else perms[b].cpu() for b in range(batch_size)])   # (:251)
            perm = _STAGE.put(perm, dev)
        # one fill + three launches (csrc/heads.hip) when the shapes allow; the library-op form below otherwise
        want_assign = self.cfg.general.task != "test"
        fused = heads.stack_to_batch(pf, scores_in, crop.detach(), bids_in,
                                     perm, data_dict["center_label"] if want_assign else None, batch_size, K)
        if fused is None and lazy is not None:
            raise RuntimeError("compact_proposals=False needs the fused stack_to_batch (shapes outside its limits)")
        if fused is not None:
            data_dict.update(zip(keys, fused[:6]))
            if want_assign:
                data_dict["object_assignment"] = fused[6]
            return data_dict
        corners = self._box_corners(crop[:, :3].detach(), crop[:, 3:6].detach()).to(pf.dtype)
        out = {
            "proposal_feats_batched": pf.new_zeros(batch_size, K, self.cfg.model.m),
            "proposal_bbox_batched": pf.new_zeros(batch_size, K, 8, 3),
            "proposal_center_batched": pf.new_zeros(batch_size, K, 3),
            "proposal_sem_cls_batched": pf.new_zeros(batch_size, K),
            "proposal_scores_batched": pf.new_zeros(batch_size, K),
            "proposal_batch_mask": pf.new_zeros(batch_size, K),
        }
        # No host round trips: rank of every proposal inside its scene by a one-hot cumulative sum, the first K per
        # scene are scattered to slot inv_perm[rank] (out[b][j] = buf[perm[j]] with buf[:n] = rows, as in the reference).
        bids = data_dict["proposals_batchId"].long()
        inv = torch.empty_like(perm).scatter_(1, perm, _const(("arange", K), dev, lambda: torch.arange(K)).expand(batch_size, K))
        onehot = torch.nn.functional.one_hot(bids, batch_size)
        rank = (onehot.cumsum(0) - onehot).gather(1, bids.view(-1, 1)).squeeze(1)
        ok = rank < K
        slot = bids * K + inv.view(-1)[bids * K + rank.clamp(max=K - 1)]
        slot = torch.where(ok, slot, torch.full_like(slot, batch_size * K))     # overflow rows -> a dump slot
        rows = {
            "proposal_feats_batched": pf, "proposal_bbox_batched": corners,
            "proposal_center_batched": crop[:, :3], "proposal_sem_cls_batched": crop[:, 7],
            "proposal_scores_batched": data_dict["proposal_objectness_scores"],
            "proposal_batch_mask": pf.new_ones(pf.size(0)),
        }
        for k, v in rows.items():
            flat = out[k].new_zeros((batch_size * K + 1,) + tuple(out[k].shape[2:]))
            flat.index_copy_(0, slot, v.to(flat.dtype))
            out[k] = flat[:-1].view(out[k].shape)
        data_dict.update(out)
        if self.cfg.general.task != "test":
            data_dict = self.get_object_assignments(data_dict)
        return data_dict

    # ------------------------------------------------------------------------------------- forward
    def forward(self, data_dict):
        batch_size = len(data_dict["batch_offsets"]) - 1
        cm = data_dict.pop("_backbone_cm", None)
        x = ME.SparseTensor(features=data_dict["voxel_feats"], coordinates=None if cm is not None else data_dict["voxel_locs"].int(),
                            coordinate_manager=cm)
        _mark("voxelize")
        out_feats = self._run_unet("backbone", self.backbone, x)
        _mark("backbone_fwd")
        pt_feats = heads.devoxelize(out_feats, data_dict["p2v_map"], data_dict.get("v2p_map"))   # (N, m)

        # both point heads in three launches: x read once, arg-max and the batch-norm statistics in the same pass (csrc/heads.hip)
        semantic_scores, semantic_preds, pt_offsets = heads.point_heads(self.sem_seg, self.offset_net, pt_feats)
        data_dict["semantic_scores"] = semantic_scores
        data_dict["pt_offsets"] = pt_offsets
        data_dict["_pt_feats"] = pt_feats

        _mark("heads")
        if data_dict["epoch"] > self.prepare_epochs or self.freeze_backbone:
            if self.teacher:   # benchmark/test switch: cluster on the labels instead of the (random-init) predictions
                semantic_preds = data_dict["sem_labels"].clamp(min=0)
                cluster_offsets = (data_dict["instance_info"][:, 0:3] - data_dict["locs"]).detach()
                cluster_offsets = torch.where((data_dict["instance_ids"] >= 0).unsqueeze(1), cluster_offsets,
                                              torch.zeros_like(cluster_offsets))
            else:
                cluster_offsets = pt_offsets
            batch_idxs = data_dict["locs_scaled"][:, 0].int()
            if not self.requires_gt_mask:
                object_idxs = torch.nonzero(semantic_preds > 0, as_tuple=False).view(-1)   # ">0" as in the reference (:288)
                # the object points' batch ids / coordinates / shifted coordinates / classes in one pass (csrc/clusterprep.hip)
                # (+ the object points' batch offsets, :296 get_batch_offsets, off the boundaries of the sorted id column in the same launch)
                if SELECT_WITH_OFFSETS:
                    batch_idxs_, coords_, shifted_xyz, semantic_preds_, batch_offsets_ = pointgroup_ops.cluster_select(
                        data_dict["locs"], cluster_offsets.detach(), semantic_preds, batch_idxs, object_idxs, batch_size=batch_size)
                else:
                    batch_idxs_, coords_, shifted_xyz, semantic_preds_ = pointgroup_ops.cluster_select(
                        data_dict["locs"], cluster_offsets.detach(), semantic_preds, batch_idxs, object_idxs)
                    batch_offsets_ = self.get_batch_offsets(batch_idxs_, batch_size)

                # padded (sync-free) lists cost 20 KB per object point and branch whatever nActive is: beyond the budget (default: 35 % of
                # the device's memory -- 100 GB on MI355X, above the lists' own 2.1 M-point range) the branches use the compact form
                budget = self._padded_list_budget(coords_.device)
                padded_ok = pointgroup_ops.ballquery_padded_fits(coords_.shape[0]) and \
                    pointgroup_ops.padded_clustering_bytes(coords_.shape[0], 2) <= budget

                def cluster_branch(xyz, mean_active, marks=False):
                    # (padded lists: same neighbours, no host round trip for nActive; bfs_cluster reads either form)
                    padded = pointgroup_ops.ballquery_batch_p_padded(xyz, batch_idxs_, batch_offsets_, self.cluster_radius) if padded_ok else None
                    idx_, start_len_ = padded if padded is not None else pointgroup_ops.ballquery_batch_p(
                        xyz, batch_idxs_, batch_offsets_, self.cluster_radius, mean_active)
                    if marks:
                        _mark("cl_ballquery")
                    p_idx, p_off = pointgroup_ops.bfs_cluster(semantic_preds_, idx_, start_len_, self.cluster_npoint_thre, True)   # (ascending lists)
                    if marks:
                        _mark("cl_bfs")
                    return p_idx, p_off          # (compact point ids: mapped back to scene ids by the merge below)

                # The two clusterings (shifted :296-299, original :304-307) are independent until the merge: the shifted
                # one runs on a side stream from a helper thread (ctypes releases the GIL inside libd3hip), so its count
                # phases, which synchronise their own stream, overlap the other branch instead of serialising with it.
                _mark("cl_prepare")
                self._kick_prefetch("cluster")          # (a pending input prefetch starts here: the clustering leaves most of the chip idle)
                cur = torch.cuda.current_stream()
                if self.concurrent_clustering and padded_ok:
                    # Round 5: BOTH branches from this thread -- begin (everything enqueued: ball query, count kernels, the fill with
                    # its sizes read on the device), begin, then the two ends (each waits for its count's event only).  The helper
                    # thread of rounds 2-4 sat on the critical path with its wake-ups and interpreter-lock hand-overs.
                    side = self._side_stream(coords_.device)
                    side.wait_stream(cur)

                    def begin(xyz, mean_active, tag):
                        padded = pointgroup_ops.ballquery_batch_p_padded(xyz, batch_idxs_, batch_offsets_, self.cluster_radius, ws_tag=tag)
                        return pointgroup_ops.bfs_cluster_begin(semantic_preds_, padded[0], padded[1], self.cluster_npoint_thre, True, ws_tag=tag)
                    hs = hm = None
                    try:
                        with torch.cuda.stream(side):
                            hs = begin(shifted_xyz, self.cluster_shift_meanActive, "s")       # (the longer chain first)
                        hm = begin(coords_, self.cluster_meanActive, "m")
                        _mark("cl_ballquery")
                        self._kick_prefetch("bfs")
                        self._early_point_losses(data_dict)
                        h, hm = hm, None
                        first = pointgroup_ops.bfs_cluster_end(h)
                        _mark("cl_bfs")
                        with torch.cuda.stream(side):
                            h, hs = hs, None
                            shifted = pointgroup_ops.bfs_cluster_end(h)
                    finally:
                        #  an exception between the begins and their ends must not leak the tickets (pinned buffer, event) nor
                        # let this stream run ahead of the side stream's speculative fill, which still writes the tickets' tensors
                        for h in (hm, hs):
                            if h is not None:
                                try:
                                    pointgroup_ops.bfs_cluster_end(h)
                                except Exception:
                                    pass
                        cur.wait_stream(side)
                    for t in shifted:
                        t.record_stream(cur)
                elif self.concurrent_clustering:      # (also: batches beyond the padded lists' range, whose compact form has a host wait per branch)
                    side = self._side_stream(coords_.device)
                    side.wait_stream(cur)

                    def work():
                        with torch.cuda.device(coords_.device), torch.cuda.stream(side):
                            return cluster_branch(shifted_xyz, self.cluster_shift_meanActive)
                    fut = _cluster_worker().submit(work)
                    try:
                        first = cluster_branch(coords_, self.cluster_meanActive, True)
                        self._kick_prefetch("bfs")
                        # the shifted branch (capped lists: label push) runs ~0.4 ms longer: the point losses, which need
                        # nothing from the clustering, fill this stream's wait for it
                        self._early_point_losses(data_dict)
                    finally:
                        shifted = fut.result()      # (an exception of the helper is re-raised here)
                    cur.wait_stream(side)
                    for t in shifted:
                        t.record_stream(cur)
                else:
                    shifted = cluster_branch(shifted_xyz, self.cluster_shift_meanActive)
                    first = cluster_branch(coords_, self.cluster_meanActive)
                # merge (:299-316): scene point ids, batch ids, the shifted set's cluster ids / offsets behind the first set's,
                # including the reference's one-element-short batch-id concat -- one launch (d3_cluster_merge)
                proposals_idx, proposals_offset, proposals_batchId_all = pointgroup_ops.cluster_merge(
                    first[0], first[1], shifted[0], shifted[1], object_idxs, batch_idxs)
            else:
                proposals_idx = data_dict["gt_proposals_idx"].to(pt_feats.device)
                proposals_offset = data_dict["gt_proposals_offset"].to(pt_feats.device)
                proposals_batchId_all = batch_idxs[proposals_idx[:, 1].long()].int()

            _mark("clustering")
            num_proposals = proposals_offset.shape[0] - 1
            data_dict["num_raw_proposals"] = num_proposals
            if num_proposals == 0:
                return self._no_proposals(data_dict, pt_feats)

            proposals_voxel_feats, proposals_p2v_map, (proposals_center, proposals_size) = self.clusters_voxelization(
                proposals_idx, proposals_offset, pt_feats, data_dict["locs"], self.score_fullscale, self.score_scale,
                self.mode, rand=data_dict.get("cluster_rand"))

            _mark("cluster_voxelization")
            score_feats = self._run_unet("score_net", self.score_net, proposals_voxel_feats)
            _mark("score_net_fwd")
            self._kick_prefetch("scorenet")
            # Host work that does not depend on the proposals goes HERE: the device still has the cluster voxelisation and
            # ScoreNet queued, so the point losses' ~20 small launches and the slot permutation's CPU draw cost no device time;
            # after the `nonzero` below the queue is empty and every host microsecond is an idle device microsecond.
            self._early_point_losses(data_dict)
            if "slot_perms" not in data_dict:     # (same position in the CPU generator's stream as the reference's draw, :251 --
                K = self.cfg.model.max_num_proposal   # nothing between here and convert_stack_to_batch draws from it)
                data_dict["_slot_perm_staged"] = _STAGE.put(torch.stack([torch.randperm(K) for _ in range(batch_size)]), pt_feats.device)
            pt_score_feats = heads.devoxelize(score_feats, proposals_p2v_map, getattr(proposals_voxel_feats, "v2p_map", None))
            proposals_score_feats = pointgroup_ops.roipool(pt_score_feats, proposals_offset)   # (P, m)
            # (library GEMM on the device: the hipBLASLt call behind nn.Linear costs 50-100 us of host time per call in this host-bound
            # stretch of the step; CPU tensors take F.linear inside)
            scores = nativelinear.linear(proposals_score_feats, self.score_linear.weight, self.score_linear.bias)
            data_dict["proposal_scores"] = (scores, proposals_idx, proposals_offset)

            _mark("pr_roipool_score")
            sig = torch.sigmoid(scores.view(-1))
            fused = (self.cfg.model.crop_bbox and sig.is_cuda and proposals_offset.dtype == torch.int32 and proposals_idx.dtype == torch.int32
                     and proposals_batchId_all.dtype == torch.int32 and semantic_preds.dtype == torch.int64
                     and proposals_idx.is_contiguous() and semantic_preds.is_contiguous())
            if fused:   # npoint, threshold mask, batch id at the cluster start, crop box: one launch (csrc/clusterprep.hip)
                proposals_npoint, thres_mask, bid_start, crop = pointgroup_ops.proposal_prepare(
                    sig.detach(), proposals_offset.contiguous(), proposals_batchId_all.contiguous(), proposals_idx, semantic_preds,
                    proposals_center, proposals_size, self.cfg.test.TEST_SCORE_THRESH, self.cfg.test.TEST_NPOINT_THRESH)
            else:
                proposals_npoint = (proposals_offset[1:] - proposals_offset[:-1]).float()           # == the loop at :342-344
                thres_mask = torch.logical_and(sig > self.cfg.test.TEST_SCORE_THRESH,
                                               proposals_npoint > self.cfg.test.TEST_NPOINT_THRESH)
                # NOTE the reference reads the one-short batch-id vector at the cluster starts (:349); cluster starts of
                # the shifted set therefore read element start+1 of that set -- same cluster, same batch id.
                starts = proposals_offset[:-1].long().clamp(max=max(proposals_batchId_all.numel() - 1, 0))
                bid_start = proposals_batchId_all[starts]
            data_dict["proposals_npoint"] = proposals_npoint
            data_dict["proposal_thres_mask"] = thres_mask
            _mark("pr_mask")
            if (not self.compact_proposals and fused and num_proposals <= 4096 and proposals_score_feats.dtype == torch.float32
                    and batch_size * self.cfg.model.max_num_proposal <= 8192):
                # no host round trip: rejected proposals keep their rows and get batch id -1 (heads.stack_to_batch drops them)
                bids_masked = torch.where(thres_mask, bid_start, torch.full_like(bid_start, -1))
                data_dict["_stb_inputs"] = (proposals_score_feats, sig, crop, bids_masked)
                for m_ in ("pr_nonzero", "pr_index", "pr_select"):
                    _mark(m_)
                return data_dict
            keep = torch.nonzero(thres_mask).squeeze(1)    # one host round trip for the four selections below
            _mark("pr_nonzero")
            proposals_batchId = bid_start.index_select(0, keep)
            data_dict["proposals_batchId"] = proposals_batchId
            data_dict["proposal_feats"] = proposals_score_feats.index_select(0, keep)
            data_dict["proposal_objectness_scores"] = sig.index_select(0, keep)

            _mark("pr_index")
            if self.cfg.model.crop_bbox:
                if not fused:
                    crop = scores.new_zeros(num_proposals, 9)
                    crop[:, :3] = proposals_center
                    crop[:, 3:6] = proposals_size
                    crop[:, 7] = semantic_preds[proposals_idx[proposals_offset[:-1].long(), 1].long()].to(crop.dtype)
                    crop[:, 8] = sig
                data_dict["proposal_crop_bbox"] = crop.index_select(0, keep)
            _mark("pr_select")
        return data_dict

    def _no_proposals(self, data_dict, pt_feats):
        dev = pt_feats.device
        m = self.cfg.model.m
        z = lambda *s: torch.zeros(*s, device=dev)
        data_dict["proposal_scores"] = (z(0, 1), torch.zeros((0, 2), dtype=torch.int32, device=dev),
                                        torch.zeros(1, dtype=torch.int32, device=dev))
        data_dict["proposals_npoint"] = z(0)
        data_dict["proposal_thres_mask"] = torch.zeros(0, dtype=torch.bool, device=dev)
        data_dict["proposals_batchId"] = torch.zeros(0, dtype=torch.int32, device=dev)
        data_dict["proposal_feats"] = z(0, m)
        data_dict["proposal_objectness_scores"] = z(0)
        data_dict["proposal_crop_bbox"] = z(0, 9)
        return data_dict

    # ---------------------------------------------------------------------------------------- loss
    def _point_losses(self, semantic_scores, semantic_labels, pt_offsets, coords, instance_info, instance_ids):
        semantic_loss = heads.cross_entropy(semantic_scores, semantic_labels, ignore_index=self.cfg.data.ignore_label)
        offset_norm_loss, offset_dir_loss, n_valid = heads.offset_losses(pt_offsets, coords, instance_info, instance_ids,
                                                                         self.cfg.data.ignore_label)
        return semantic_loss, offset_norm_loss, offset_dir_loss, n_valid

    def _early_point_losses(self, data_dict):
        """the semantic / offset losses of `loss()` computed inside forward() (they need only the point heads' outputs and the
        labels); `loss()` takes them over when it is called with the very same tensors"""
        if self.mode == "test" or not torch.is_grad_enabled():
            return
        need = ("sem_labels", "locs", "instance_info", "instance_ids")
        if any(k not in data_dict for k in need) or not torch.is_tensor(data_dict["semantic_scores"]):
            return
        if "_point_losses" in data_dict:
            return
        args = (data_dict["semantic_scores"], data_dict["sem_labels"], data_dict["pt_offsets"], data_dict["locs"],
                data_dict["instance_info"], data_dict["instance_ids"])
        losses = self._point_losses(*args)
        graft = None
        pf = data_dict.pop("_pt_feats", None)
        if EARLY_POINT_GRADS and self.early_point_grads and pf is not None and all(torch.is_tensor(l) and l.requires_grad for l in losses[:3]):
            # d(w0 sem + w1 norm + w2 dir) / d(point features, head parameters) NOW; loss() then builds the total loss on a graft that
            # carries these gradients (scaled by whatever arrives in the backward) -- same arithmetic, earlier in the step
            w = self.cfg.train.loss_weight
            value = w[0] * losses[0] + w[1] * losses[1] + w[2] * losses[2]
            wrt = [t for t in [pf] + list(self.sem_seg.parameters()) + list(self.offset_net.parameters()) if t.requires_grad]
            grads = torch.autograd.grad(value, wrt, allow_unused=True)
            keep = [(t, g) for t, g in zip(wrt, grads) if g is not None]
            if keep:
                graft = _PointLossGraft.apply(value.detach(), len(keep), *[g for _, g in keep], *[t for t, _ in keep])
                losses = tuple(l.detach() for l in losses[:3]) + tuple(losses[3:])
        data_dict["_point_losses"] = (args, losses, graft)

    def loss(self, data_dict, epoch):
        """semantic CE + offset L1 / direction + soft-IoU score BCE (reference :387-463)."""
        semantic_scores, semantic_labels = data_dict["semantic_scores"]
        pt_offsets, coords, instance_info, instance_ids = data_dict["pt_offsets"]
        args = (semantic_scores, semantic_labels, pt_offsets, coords, instance_info, instance_ids)
        early = data_dict.pop("_point_losses", None)
        graft = None
        if early is not None and len(early[0]) == len(args) and all(a is b for a, b in zip(early[0], args)):
            semantic_loss, offset_norm_loss, offset_dir_loss, n_valid = early[1]
            graft = early[2] if len(early) > 2 else None
        else:
            if early is not None and len(early) > 2 and early[2] is not None:
                raise RuntimeError("PointGroup.loss: the point losses were computed (and back-propagated) inside forward() for other tensors "
                                   "than the ones passed to loss(); set early_point_grads = False to recompute them here")
            semantic_loss, offset_norm_loss, offset_dir_loss, n_valid = self._point_losses(*args)
        data_dict["semantic_loss"] = (semantic_loss, semantic_scores.shape[0])
        data_dict["offset_norm_loss"] = (offset_norm_loss, n_valid)
        data_dict["offset_dir_loss"] = (offset_dir_loss, n_valid)

        w = self.cfg.train.loss_weight
        loss = graft if graft is not None else w[0] * semantic_loss + w[1] * offset_norm_loss + w[2] * offset_dir_loss
        if epoch > self.cfg.cluster.prepare_epochs:
            scores, proposals_idx, proposals_offset, instance_pointnum = data_dict["proposal_scores"]
            if scores.shape[0] > 0:
                ious = pointgroup_ops.get_iou(proposals_idx[:, 1].contiguous(), proposals_offset, instance_ids,
                                              instance_pointnum)
                score_loss, gt_ious = heads.score_loss(scores, ious, self.cfg.train.fg_thresh, self.cfg.train.bg_thresh)
            else:  # the reference would produce NaN (mean of an empty tensor); keep the step finite
                gt_ious = scores.new_zeros(0)
                score_loss = scores.sum() * 0
            data_dict["score_loss"] = (score_loss, gt_ious.shape[0])
            loss = loss + w[3] * score_loss
        data_dict["total_loss"] = (loss, semantic_labels.shape[0])
        return data_dict

    # ------------------------------------------------------------------------------- entry points
    def feed(self, data_dict, epoch=0):
        """(reference :466-479)"""
        data_dict["epoch"] = epoch
        t = data_dict.pop("_prefetch", None)
        got = self._take_prefetch(t, data_dict) if t is not None else None
        if got is None:
            got = self._input_stage(data_dict)
        cm, data_dict["voxel_feats"] = got
        if cm is not None:
            data_dict["_backbone_cm"] = cm
        data_dict = self.forward(data_dict)
        if data_dict["epoch"] > self.prepare_epochs or self.freeze_backbone:
            data_dict = self.convert_stack_to_batch(data_dict, perms=data_dict.get("slot_perms"))
        return data_dict

    def parse_feed_ret(self, data_dict, epoch=0):
        """(reference :481-510)"""
        semantic_scores = data_dict["semantic_scores"]
        pt_offsets = data_dict["pt_offsets"]
        preds = {"semantic": semantic_scores, "pt_offsets": pt_offsets}
        if self.mode != "test":
            data_dict["semantic_scores"] = (semantic_scores, data_dict["sem_labels"])
            data_dict["pt_offsets"] = (pt_offsets, data_dict["locs"], data_dict["instance_info"], data_dict["instance_ids"])
        if epoch > self.cfg.cluster.prepare_epochs:
            scores, proposals_idx, proposals_offset = data_dict["proposal_scores"]
            preds["score"] = scores
            preds["proposals"] = (proposals_idx, proposals_offset)
            preds["proposal_crop_bboxes"] = data_dict.get("proposal_crop_bbox")
            if self.mode != "test":
                data_dict["proposal_scores"] = (scores, proposals_idx, proposals_offset, data_dict["instance_num_point"])
                if self.cfg.model.crop_bbox and "proposal_crop_bbox" in data_dict:
                    data_dict["proposal_crop_bboxes"] = data_dict["proposal_crop_bbox"]
        return preds, data_dict

    @torch.no_grad()
    def predict_instances(self, data_dict):
        """The instance predictions `PointGroup.test` writes out (reference :561-601), without the files: proposals above the
        score / size thresholds, point-mask NMS (lib/utils/eval.py:75-97).  Device-side: the pairwise mask IoUs come from the
        (cluster, point) lists (csrc/nms.hip) instead of a dense (nProposal, N) mask product and a host copy.
        -> dict(pick (n,) indices into the proposals, scores (n,), proposals_idx, proposals_offset, semantic_pred (N,))"""
        import ctypes as C
        from . import _lib
        data_dict = self.feed(data_dict, self.current_epoch)
        scores, proposals_idx, proposals_offset = data_dict["proposal_scores"][:3]
        sem_pred = data_dict["semantic_scores"].max(1)[1]
        dev = scores.device
        P, N = proposals_offset.numel() - 1, sem_pred.numel()
        empty = dict(pick=torch.zeros(0, dtype=torch.long, device=dev), scores=scores.new_zeros(0), proposals_idx=proposals_idx,
                     proposals_offset=proposals_offset, semantic_pred=sem_pred)
        if P == 0:
            return empty
        sig = torch.sigmoid(scores.view(-1)).contiguous()
        keep = data_dict["proposal_thres_mask"].to(torch.uint8).contiguous()
        ious = torch.empty((P, P), dtype=torch.float32, device=dev)
        member = torch.empty(2 * N, dtype=torch.int32, device=dev)
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        order = torch.empty(P, dtype=torch.int32, device=dev)
        picked = torch.empty(P, dtype=torch.int32, device=dev)
        L = _lib.lib()
        cidx, off = proposals_idx.contiguous(), proposals_offset.contiguous()
        p_, st = (lambda t: C.c_void_p(t.data_ptr())), C.c_void_p(torch.cuda.current_stream().cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(L.d3_instance_cross_iou(p_(cidx), p_(off), cidx.shape[0], P, N, p_(ious), p_(member), p_(flags), st), "instance_cross_iou")
            _lib.check(L.d3_nms_matrix(p_(ious), p_(sig), p_(keep), P, float(self.cfg.test.TEST_NMS_THRESH), p_(order), p_(picked),
                                       p_(flags[1:]), st), "nms_matrix")
        over, n = flags.tolist()
        if over:
            raise _lib.D3Error("predict_instances: a point belongs to more than two proposals")
        pick = picked[:n].long()
        empty.update(pick=pick, scores=sig[pick], cross_ious=ious)
        return empty

    def training_step(self, data_dict, idx=0):
        """(reference :513-528) minus the Lightning logging."""
        _mark("begin")
        data_dict = self.feed(data_dict, self.current_epoch)
        _mark("proposals")
        _, data_dict = self.parse_feed_ret(data_dict, self.current_epoch)
        data_dict = self.loss(data_dict, self.current_epoch)
        _mark("loss")
        return data_dict["total_loss"][0], data_dict
