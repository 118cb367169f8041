"""`PipelineNet`: detector -> speaker / listener step logic with the reference's mode table, sub-module names
(`detector`, `speaker`, `listener`), loss composition and logged keys (reference: model/pipeline.py:25-123,134-226,
738-757).  A plain nn.Module: Lightning's `self.log` becomes `self.logged` (a dict filled per step; the reference's
per-key `sync_dist=True` scalar all-reduces collapse into one packed all-reduce in `reduce_logged`).
Built: modes 0 (detector), 1 (detector -> speaker, cross-entropy) and 2 (detector -> listener).  Mode 3 (joint
self-critical training through `moderator`, model/pipeline.py:228-309,759-892) is the next row."""
import random

import numpy as np
import torch
import torch.nn as nn

from .captioning_loss import get_captioning_loss
from .listener import ListenerNet, get_grounding_loss, get_lobjcls_loss
from .pointgroup import PointGroup
from .speaker import SpeakerNet


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
            self.vocabulary = dataset["train"].vocabulary
            self.register_buffer("embeddings", torch.as_tensor(dataset["train"].glove, dtype=torch.float32))
            self.loss_opt = {"use_rl": cfg.train.use_rl, "sample_topn": cfg.train.sample_topn, "idx2word": self.vocabulary["idx2word"],
                             "max_len": cfg.data.max_spk_len + 2, "loss_type": cfg.model.loss_type}
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

    def _detect(self, data_dict):
        data_dict = self.detector.feed(data_dict, self.current_epoch)
        _, data_dict = self.detector.parse_feed_ret(data_dict, self.current_epoch)
        return self.detector.loss(data_dict, self.current_epoch)

    def training_step(self, data_dict, idx=0):
        self.logged = {}
        if self.mode == 0:
            data_dict = self._detect(data_dict)
            loss = data_dict["total_loss"][0]
            for k, v in data_dict.items():
                if "loss" in k:
                    self.log("train/{}".format(k), v[0])
        elif self.mode == 1:
            data_dict = self.speaker(self._detect(data_dict))
            _, data_dict = get_captioning_loss(data_dict, caption=not self.no_captioning, orientation=self.cfg.model.use_orientation,
                                               num_bins=self.cfg.data.num_ori_bins, loss_opt=self.loss_opt)
            loss = data_dict["total_loss"][0] + data_dict["cap_loss"] + 0.1 * data_dict["ori_loss"]
            for k, v in {"loss": loss, "detect_loss": data_dict["total_loss"][0], "captioning_loss": data_dict["cap_loss"],
                         "orientation_loss": data_dict["ori_loss"], "cap_acc": data_dict["cap_acc"], "ori_acc": data_dict["ori_acc"],
                         "pred_ious": data_dict["pred_ious"]}.items():
                self.log("train_{}/{}".format("loss" if "loss" in k else "score", k), v)
        elif self.mode == 2:
            data_dict = self.listener(self._detect(data_dict))
            _, data_dict = get_grounding_loss(data_dict)
            if self.use_lang_classifier:
                _, data_dict = get_lobjcls_loss(data_dict)
            else:
                data_dict["lang_loss"] = data_dict["ref_loss"].new_zeros(()); data_dict["lang_acc"] = data_dict["lang_loss"]
            loss = data_dict["total_loss"][0] + data_dict["ref_loss"] + data_dict["lang_loss"]
            for k, v in {"loss": loss, "detect_loss": data_dict["total_loss"][0], "grounding_loss": data_dict["ref_loss"],
                         "lobjcls_loss": data_dict["lang_loss"], "ref_acc_mean": data_dict["ref_acc_mean"],
                         "ref_iou_mean": data_dict["ref_iou_mean"], "best_ious_mean": data_dict["best_ious_mean"],
                         "ref_iou_rate_0.25": data_dict["ref_iou_rate_0.25"], "ref_iou_rate_0.5": data_dict["ref_iou_rate_0.5"],
                         "lang_acc": data_dict["lang_acc"]}.items():
                self.log("train_{}/{}".format("loss" if "loss" in k else "score", k), v)
        else:
            raise NotImplementedError("mode 3 (joint speaker-listener, self-critical) is not built yet")
        self.global_step += 1
        return loss, data_dict

    def configure_optimizers(self):
        """AdamW + StepLR(10, 0.8) over the trainable parameters (model/pipeline.py:738-757)"""
        params = [p for p in self.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=self.cfg.train.optim.lr, weight_decay=self.cfg.train.optim.weight_decay, fused=params[0].is_cuda)
        return [opt], [torch.optim.lr_scheduler.StepLR(opt, step_size=10, gamma=0.8)]
