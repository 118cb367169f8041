"""Checkpoint surface of the reference (SURVEY.md section 5 "Checkpoint / resume", Appendix B): the state-dict key layout is
part of the API -- stage hand-off goes through plain per-module state dicts (`scripts/prepare_weights.py:256-284` writes
`pretrained/{detector,speaker,listener}.pth`, `scripts/train.py:288-310` loads them, `:312-325` freezes modules) and
evaluation loads a Lightning checkpoint's "state_dict" non-strictly (`scripts/eval.py:120-121`).  Same functions here over
d3net_amd.pipeline.PipelineNet, whose module tree reproduces the reference's keys."""
import os

import torch

MODULES = ("detector", "speaker", "listener")


def module_state_dict(model, which):
    """state dict of one sub-module with the keys `prepare_weights.py` writes (no `detector.` prefix)"""
    assert which in MODULES, which
    return getattr(model, which).state_dict()


def split_state_dict(state_dict):
    """a whole-pipeline state dict (`checkpoint["state_dict"]`) -> {"detector": {...}, "speaker": {...}, "listener": {...}}"""
    out = {m: {} for m in MODULES}
    for k, v in state_dict.items():
        head, _, rest = k.partition(".")
        if head in out:
            out[head][rest] = v
    return out


def save_module_weights(model, which, path):
    """`python scripts/prepare_weights.py -m <which> -n <name>`"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(module_state_dict(model, which), path)


def load_lightning_checkpoint(model, path_or_dict, strict=False):
    """`model.load_state_dict(checkpoint["state_dict"], strict=False)` (scripts/eval.py:120-121)"""
    ckpt = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    return model.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt, strict=strict)


def load_pretrained(model, cfg, root=None):
    """scripts/train.py:288-310: cfg.model.pretrained_{detector,speaker,listener} under cfg.PRETRAINED_PATH, strictly"""
    root = root if root is not None else cfg.PRETRAINED_PATH
    loaded = []
    if cfg.model.get("use_checkpoint"):
        return loaded
    for which, absent in (("detector", cfg.model.no_detection), ("speaker", cfg.model.no_captioning), ("listener", cfg.model.no_grounding)):
        name = cfg.model.get("pretrained_" + which)
        if name and not absent:
            getattr(model, which).load_state_dict(torch.load(os.path.join(root, name), map_location="cpu"))
            loaded.append(which)
    return loaded


def apply_freeze(model, cfg):
    """scripts/train.py:312-325: cfg.model.freeze_* -> requires_grad False for the whole sub-module"""
    frozen = []
    for which in MODULES:
        if cfg.model.get("freeze_" + which) and hasattr(model, which):
            for p in getattr(model, which).parameters():
                p.requires_grad = False
            frozen.append(which)
    return frozen
