"""`PipelineNet`: detector -> speaker / listener step logic with the reference's mode table, sub-module names
(`detector`, `speaker`, `listener`), loss composition and logged keys (reference: model/pipeline.py:25-123,134-226,
738-757).  A plain nn.Module: Lightning's `self.log` becomes `self.logged` (a dict filled per step; the reference's
per-key `sync_dist=True` scalar all-reduces collapse into one packed all-reduce in `reduce_logged`).
Built: modes 0 (detector), 1 (detector -> speaker), 2 (detector -> listener) and 3 (joint speaker-listener training,
self-critical when cfg.train.use_rl, through `moderator`: model/pipeline.py:228-309,759-892)."""
import random

import numpy as np
import torch
import torch.nn as nn

from .captioning_loss import get_captioning_loss
from .listener import ListenerNet, get_grounding_loss, get_lobjcls_loss
from .pointgroup import PointGroup, _mark
from .speaker import SpeakerNet

DEFER_DETECT_LOSS = 1   # (A/B switch, tools/ab.py py:d3net_amd.pipeline.DEFER_DETECT_LOSS=0,1) training_step mode 1: the detector's losses are built after the captioner's forward has been enqueued (speaker step 16.23 -> 16.17 ms, r05_j52)


class _HostScalars:
    """a few device scalars on their way to the host: the copy is enqueued at construction (pinned buffer, own event) and
    waited for only at `.get()` -- by then it has long completed, the stream is not drained"""

    def __init__(self, t):
        self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.host.copy_(t, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record()

    def get(self):
        self.ev.synchronize()
        return self.host.tolist()


class PipelineNet(nn.Module):
    def __init__(self, cfg, dataset=None):
        super().__init__()
        self.cfg = cfg
        self.init_random_seed()
        self.no_detection, self.no_captioning, self.no_grounding = cfg.model.no_detection, cfg.model.no_captioning, cfg.model.no_grounding
        self._get_current_mode()
        self.current_epoch, self.global_step = 0, 0
        self.logged = {}
        if dataset:
            # validation: the description store of the validation split (model/pipeline.py:41; lib/captioning/eval_helper.py:35-62)
            val = dataset.get("val", dataset["train"]) if isinstance(dataset, dict) else dataset["train"]
            self.dataset_chunk_data = getattr(val, "chunked_data", None)
            self.val_raw_data = getattr(val, "raw_data", None)
            self.vocabulary = dataset["train"].vocabulary
            self.register_buffer("embeddings", torch.as_tensor(dataset["train"].glove, dtype=torch.float32))
            self.beam_opt = {"train_beam_size": cfg.train.beam_size, "train_sample_topn": cfg.train.sample_topn,
                             "eval_beam_size": cfg.train.beam_size}
            self.loss_opt = {"use_rl": cfg.train.use_rl, "sample_topn": cfg.train.sample_topn, "idx2word": self.vocabulary["idx2word"],
                             "train_dataset_data": getattr(dataset["train"], "chunked_data", None),
                             "organized_data": getattr(dataset["train"], "organized", None),
                             "max_len": cfg.data.max_spk_len + 2, "loss_type": cfg.model.loss_type,
                             "ref_reward_weight": cfg.train.ref_reward_weight, "lang_reward_weight": cfg.train.lang_reward_weight,
                             "listener_reward_weight": cfg.train.listener_reward_weight,
                             "caption_reward_weight": cfg.train.caption_reward_weight}
        if self.no_detection:
            raise NotImplementedError("GT-proposal modes 4-6 (no_detection) are not on the hot path")
        self.detector = PointGroup(cfg)
        if not self.no_captioning:
            self.speaker = SpeakerNet(cfg, self.vocabulary, self.embeddings)
        if not self.no_grounding:
            self.listener = ListenerNet(cfg)
        self.use_lang_classifier = cfg.model.use_lang_classifier

    def _get_current_mode(self):
        """0 detector | 1 detector->speaker | 2 detector->listener | 3 detector->speaker->listener (model/pipeline.py:91-123)"""
        assert not (self.no_detection and self.no_captioning and self.no_grounding)
        if self.no_detection:
            self.mode = 4 if (self.no_grounding and not self.no_captioning) else 5 if (not self.no_grounding and self.no_captioning) else 6
        else:
            self.mode = 0 if (self.no_grounding and self.no_captioning) else 1 if self.no_grounding else 2 if self.no_captioning else 3

    def init_random_seed(self):
        s = self.cfg.general.manual_seed
        if s:
            random.seed(s); np.random.seed(s); torch.manual_seed(s)
            if torch.cuda.is_available():
                torch.cuda.manual_seed_all(s)

    def zero_grad(self, set_to_none=True):
        """the detector's executors own their gradients (PointGroup.zero_grad marks them stale instead of detaching
        ~500 views); everything else is plain nn.Module.zero_grad"""
        self.detector.zero_grad(set_to_none)
        rest = self.__dict__.get("_zg_rest")
        if rest is None:
            det = {id(p) for p in self.detector.parameters()}
            rest = self.__dict__["_zg_rest"] = [p for p in self.parameters() if id(p) not in det]
        for p in rest:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_(); p.grad.zero_()

    def log(self, name, value, **kw):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def reduce_logged(self):
        """one packed all-reduce for all logged scalars (the reference issues one per key via sync_dist=True)"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or not self.logged:
            return self.logged
        keys = sorted(self.logged)
        dev = next(self.parameters()).device
        vec = torch.tensor([float(self.logged[k]) for k in keys], device=dev)
        dist.all_reduce(vec)
        vec /= dist.get_world_size()
        return {k: vec[i] for i, k in enumerate(keys)}

    def _detect(self, data_dict, defer_loss=False):
        """defer_loss: the caller runs `self.detector.loss` itself, AFTER it has enqueued the heads (DEFER_DETECT_LOSS)"""
        keep = self.detector.compact_proposals      # (only for this call: the detector object may be driven directly as well)
        self.detector.compact_proposals = keep and not self.__dict__.get("_lazy_proposals", False)
        try:
            data_dict = self.detector.feed(data_dict, self.current_epoch)
        finally:
            self.detector.compact_proposals = keep
        _, data_dict = self.detector.parse_feed_ret(data_dict, self.current_epoch)
        if not defer_loss:
            data_dict = self.detector.loss(data_dict, self.current_epoch)
        gb = self.__dict__.get("grad_boundary")
        if gb is not None and torch.is_tensor(data_dict.get("proposal_feats_batched")):
            # multi-GPU: everything the heads hand back to the detector comes through the proposal features; when the
            # backward pass crosses this point the heads' gradients are complete and their all-reduce can start
            # underneath the detector's backward (distributed.BucketGradAllReduce.boundary)
            data_dict["proposal_feats_batched"], = gb.boundary(data_dict["proposal_feats_batched"])
        return data_dict

    def training_step(self, data_dict, idx=0):
        self.logged = {}
        # the heads consume only the batched proposal tensors: no compaction of the kept proposals, hence no host round trip
        # between ScoreNet and the heads (PointGroup.compact_proposals); the speaker's two host scalars depend only on the
        # batch's language inputs and are requested now, long before they are needed
        self._lazy_proposals = True
        if self.mode in (1, 3) and "lang_len" in data_dict and "annotated" in data_dict and data_dict["lang_len"].is_cuda:
            data_dict["_spk_host_meta"] = _HostScalars(torch.stack([data_dict["lang_len"].reshape(-1).max().long(),
                                                                    (data_dict["annotated"].reshape(-1) != 1).sum()]))
        if self.mode == 0:
            data_dict = self._detect(data_dict)
            loss = data_dict["total_loss"][0]
            for k, v in data_dict.items():
                if "loss" in k:
                    self.log("train/{}".format(k), v[0])
        elif self.mode == 1:
            # the detector's score loss (IoU table, BCE: a handful of small launches and their interpreter time) needs nothing from the
            # heads and the heads nothing from it: it is enqueued BEHIND the captioner's recurrence, whose ~300 launches keep the device
            # busy for longer than the host needs to issue them -- between ScoreNet and the relation graph the device waits for the host
            defer = bool(DEFER_DETECT_LOSS)
            data_dict = self._detect(data_dict, defer_loss=defer)
            self.detector._kick_prefetch("caption")      # (input prefetch, if set to start under the captioner's recurrence)
            data_dict = self.speaker(data_dict)
            _mark("speaker")
            if defer:
                data_dict = self.detector.loss(data_dict, self.current_epoch)
            _, data_dict = get_captioning_loss(data_dict, caption=not self.no_captioning, orientation=self.cfg.model.use_orientation,
                                               num_bins=self.cfg.data.num_ori_bins, loss_opt=self.loss_opt)
            loss = data_dict["total_loss"][0] + data_dict["cap_loss"] + 0.1 * data_dict["ori_loss"]
            _mark("caption_losses")
            for k, v in {"loss": loss, "detect_loss": data_dict["total_loss"][0], "captioning_loss": data_dict["cap_loss"],
                         "orientation_loss": data_dict["ori_loss"], "cap_acc": data_dict["cap_acc"], "ori_acc": data_dict["ori_acc"],
                         "pred_ious": data_dict["pred_ious"]}.items():
                self.log("train_{}/{}".format("loss" if "loss" in k else "score", k), v)
        elif self.mode == 2:
            # (the deferred detector loss of mode 1 measured no gain behind the listener's forward: 18.90 vs 19.03 ms, r05_j52)
            data_dict = self._ground(self.listener(self._detect(data_dict)), False)
            loss = data_dict["total_loss"][0] + data_dict["ref_loss"] + data_dict["lang_loss"]
            for k, v in {"loss": loss, "detect_loss": data_dict["total_loss"][0], "grounding_loss": data_dict["ref_loss"],
                         "lobjcls_loss": data_dict["lang_loss"], "ref_acc_mean": data_dict["ref_acc_mean"],
                         "ref_iou_mean": data_dict["ref_iou_mean"], "best_ious_mean": data_dict["best_ious_mean"],
                         "ref_iou_rate_0.25": data_dict["ref_iou_rate_0.25"], "ref_iou_rate_0.5": data_dict["ref_iou_rate_0.5"],
                         "lang_acc": data_dict["lang_acc"]}.items():
                self.log("train_{}/{}".format("loss" if "loss" in k else "score", k), v)
        elif self.mode == 3:
            assert len(data_dict) == 2
            use_rl = self.cfg.train.use_rl
            assert use_rl, "mode 3 only works self-critically: `moderator` needs the sampled and greedy captions (pipeline.py:773-774)"
            spk = self.speaker(self._detect(data_dict[0]), use_rl=use_rl, is_eval=False, beam_opt=self.beam_opt)
            spk = self.moderator(spk, self.cfg.data.max_spk_len + 2)
            spk = self._ground(self.listener(spk, use_rl=use_rl), use_rl)
            _, spk = get_captioning_loss(spk, caption=not self.no_captioning, orientation=self.cfg.model.use_orientation,
                                         num_bins=self.cfg.data.num_ori_bins, loss_opt=self.loss_opt)
            loss = spk["total_loss"][0] + spk["cap_loss"] + 0.1 * spk["ori_loss"] + spk["ref_loss"] + spk["lang_loss"]
            lis = self._ground(self.listener(self._detect(data_dict[1])), False)
            loss = loss + lis["total_loss"][0] + lis["ref_loss"] + lis["lang_loss"]
            avg = lambda k: (spk[k] + lis[k]) / 2
            logs = {"loss": loss, "detect_loss": (spk["total_loss"][0] + lis["total_loss"][0]) / 2, "captioning_loss": spk["cap_loss"],
                    "orientation_loss": spk["ori_loss"], "grounding_loss": avg("ref_loss"), "lobjcls_loss": avg("lang_loss"),
                    "cap_acc": spk["cap_acc"], "ori_acc": spk["ori_acc"], "pred_ious": spk["pred_ious"], "cap_rwd": spk["cap_rwd"],
                    "loc_rwd": spk["loc_rwd"], "ttl_rwd": spk["ttl_rwd"]}
            for k in ("ref_acc_mean", "ref_iou_mean", "best_ious_mean", "ref_iou_rate_0.25", "ref_iou_rate_0.5", "lang_acc"):
                logs[k] = avg(k)
            for k, v in logs.items():
                self.log("train_{}/{}".format("loss" if "loss" in k else "score", k), v)
            data_dict = {"speaker": spk, "listener": lis}
        else:
            raise NotImplementedError("GT-proposal modes 4-6 (no_detection) are not on the hot path")
        self.global_step += 1
        return loss, data_dict

    # --------------------------------------------------------------------------------------------- validation
    def _log_val(self, d, keys):
        for k in keys:
            self.log("val_{}/{}".format("loss" if "loss" in k else "score", k), d[k])
        return {k: d[k] for k in keys}

    _GROUND_KEYS = ("ref_acc_mean", "ref_iou_mean", "best_ious_mean", "ref_iou_rate_0.25", "ref_iou_rate_0.5", "lang_acc")

    @torch.no_grad()
    def validation_step(self, data_dict, idx=0, dataloader_idx=0):
        """(model/pipeline.py:457-643) detector losses (mode 0), dense-caption candidates of the batch (modes 1 / 3 loader 0:
        evaluation decode of all proposals + Hungarian assignment to the GT boxes), grounding scores (modes 2 / 3 loader 1).
        Precision policy: in eval() the U-Nets run their fp32 twin executors and the heads' GEMMs stay exact fp32."""
        from . import minkowski as ME
        with ME.heads_exact_for(self.training):
            return self._validation_step(data_dict, idx, dataloader_idx)

    def _validation_step(self, data_dict, idx=0, dataloader_idx=0):
        from .caption_eval import eval_caption_step
        if self.mode not in (0, 1, 2, 3):
            raise NotImplementedError("GT-proposal modes 4-6 (no_detection) are not on the hot path")
        self._lazy_proposals = False                # (the evaluation code reads the compact per-proposal tensors)
        data_dict = self._detect(data_dict)
        if self.mode == 0:
            for k, v in data_dict.items():
                if "loss" in k:
                    self.log("val_loss/{}".format(k), v[0])
            return None
        if self.mode == 1 or (self.mode == 3 and dataloader_idx == 0):
            data_dict = self.speaker(data_dict, use_tf=False, use_rl=False, is_eval=True, beam_opt=self.beam_opt)
            return eval_caption_step(data_dict, self.vocabulary)
        if self.mode == 2 or (self.mode == 3 and dataloader_idx == 1):
            data_dict = self._ground(self.listener(data_dict), False)
            out = self._log_val(data_dict, self._GROUND_KEYS) if self.mode == 2 else {k: data_dict[k] for k in self._GROUND_KEYS}
            return out if self.mode == 3 else None
        raise NotImplementedError("dataloader_idx %d" % dataloader_idx)

    def validation_epoch_end(self, outputs):
        """(model/pipeline.py:645-735) corpus scores over the epoch's candidates; mode 3 adds the mean grounding scores and
        `combined` = CIDEr + Acc@0.5IoU, the monitor of the joint training"""
        from .caption_eval import eval_caption_epoch
        log = {}
        # the fp32 twin executors evaluation forwards instantiated beside the bf16 ones (packed fp32 weights, arena plan, flat
        # gradient buffer) go away with the validation epoch; the next evaluation forward rebuilds them
        if not self.no_detection and hasattr(self.detector, "release_eval_executors"):
            self.detector.release_eval_executors()
        if self.mode in (0, 2):
            return log
        cap_outs = outputs if self.mode == 1 else outputs[0]
        candidates = {}
        for outs in cap_outs:
            for k, v in (outs or {}).items():
                candidates.setdefault(k, v)
        bleu, cider, rouge, meteor = eval_caption_epoch(candidates, self.val_raw_data or [], max_len=self.cfg.eval.max_des_len + 2,
                                                        min_iou=self.cfg.eval.min_iou_threshold)
        log = {"bleu-1": bleu[0][0], "bleu-2": bleu[0][1], "bleu-3": bleu[0][2], "bleu-4": bleu[0][3], "cider": cider[0],
               "meteor": meteor[0], "rouge": rouge[0]}
        if self.mode == 3:
            metrics = {}
            for outs in outputs[1]:
                for k, v in (outs or {}).items():
                    metrics.setdefault(k, []).append(float(v))
            for k, v in metrics.items():
                log[k] = float(np.mean(v))
            log["combined"] = log["cider"] + log.get("ref_iou_rate_0.5", 0.0)
        for k, v in log.items():
            self.log("val_score/{}".format(k), v)
        return log

    def forward(self, data_dict):
        """inference entry point (model/pipeline.py:894-925): detector -> speaker -> listener, whichever exist"""
        from . import minkowski as ME
        with ME.heads_exact_for(self.training):
            return self._forward(data_dict)

    def _forward(self, data_dict):
        if not self.no_detection:
            data_dict = self.detector.feed(data_dict, self.current_epoch)
        if not self.no_captioning:
            data_dict = self.speaker(data_dict)
        if not self.no_grounding:
            data_dict = self.listener(data_dict)
        return data_dict

    def _ground(self, data_dict, use_rl):
        """lib/grounding/loss_helper.py:304-335 `get_loss`"""
        _, data_dict = get_grounding_loss(data_dict, use_rl=use_rl)
        if self.use_lang_classifier:
            _, data_dict = get_lobjcls_loss(data_dict, use_rl=use_rl)
        else:
            data_dict["lang_loss"] = data_dict["ref_loss"].new_zeros(()); data_dict["lang_acc"] = data_dict["lang_loss"]
        return data_dict

    def moderator(self, data_dict, max_spk_len):
        """Turn the speaker's sampled and greedy captions into listener inputs, and the speaker's targets into the
        listener's (pseudo) ground truth (model/pipeline.py:759-892).  After it the listener batch is
        (scene, sample, chunk): `lang_feat[...]` is (B*topn, chunk, T, 300).

        One host transfer for all token lists; the one-hot x embedding products of the reference (:819-826) are row
        lookups (a one-hot row selects exactly one embedding row; padding positions select row 0, `pad_`).
        NOTE `lang_len[...]` keeps the (scene*chunk, sample) layout (:846-849) while the embeddings have the sample
        axis moved out -- the two only line up for chunk == 1 or topn == 1; kept as in the reference."""
        sampled, baseline = data_dict["lang_cap"], data_dict["baseline_cap"]
        assert len(sampled[0]) == len(baseline[0])
        N, topn = len(sampled), len(sampled[0])
        feats = data_dict["bbox_feature"]
        B, dev = feats.shape[0], feats.device
        Cn = N // B
        sos, eos = 2, 3                                                        # hard-coded in the reference (:786)

        def to_matrix(table):
            lens = [len(table[n][k]) for n in range(N) for k in range(topn)]
            flat = torch.cat([table[n][k].reshape(-1) for n in range(N) for k in range(topn)]).tolist()
            mat = np.zeros((N * topn, max_spk_len), np.int64)
            out_len = np.zeros(N * topn, np.int64)
            pos = 0
            for i, l in enumerate(lens):
                toks = [sos] + flat[pos:pos + l]
                pos += l
                if eos not in toks:
                    toks.append(eos)
                assert len(toks) <= max_spk_len
                mat[i, :len(toks)] = toks
                out_len[i] = len(toks)
            return (torch.from_numpy(mat).to(dev).view(N, topn, max_spk_len), torch.from_numpy(out_len).to(dev).view(N, topn))

        def embed(mat):
            e = self.embeddings[mat]                                               # (N, topn, T, 300)
            e = e.reshape(-1, Cn, topn, max_spk_len, e.shape[-1]).transpose(2, 1)
            return e.reshape(-1, Cn, max_spk_len, e.shape[-1])

        s_mat, s_len = to_matrix(sampled)
        b_mat, b_len = to_matrix(baseline)
        data_dict["sampled_topn"] = topn
        data_dict["lang_feat"] = {"sampled": embed(s_mat), "baseline": embed(b_mat)}
        data_dict["lang_len"] = {"sampled": s_len, "baseline": b_len}
        # pseudo ground truth: box / class of the GT object assigned to every description's target
        assigned = data_dict["assigned_bbox_id_labels"].reshape(-1, Cn).unsqueeze(1).repeat(1, topn, 1)   # (B, topn, Cn)
        corners, sems = data_dict["proposal_bbox_batched"], data_dict["proposal_sem_cls_batched"]
        scene = torch.arange(B, device=dev).view(B, 1, 1).expand_as(assigned)
        data_dict["ref_box_corner_label"] = corners[scene, assigned].reshape(B * topn, Cn, 8, 3)
        cat = sems[scene, assigned].reshape(B * topn, Cn) - 2                      # wall / floor are not ScanRefer classes
        cat[cat < 0] = 17
        data_dict["ref_cat_label"] = cat
        return data_dict

    def configure_optimizers(self):
        """AdamW + StepLR(10, 0.8) over the trainable parameters (model/pipeline.py:738-757)"""
        params = [p for p in self.parameters() if p.requires_grad]
        if params[0].is_cuda:   # one launch for all tensors (csrc/heads.hip); same update rule
            from .optim import FusedAdamW
            opt = FusedAdamW(params, lr=self.cfg.train.optim.lr, weight_decay=self.cfg.train.optim.weight_decay)
            opt.register_step_pre_hook(lambda *a: self.detector.drop_stale_grads())
        else:
            opt = torch.optim.AdamW(params, lr=self.cfg.train.optim.lr, weight_decay=self.cfg.train.optim.weight_decay)
        return [opt], [torch.optim.lr_scheduler.StepLR(opt, step_size=10, gamma=0.8)]
