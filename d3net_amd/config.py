"""Attribute-style configuration with the reference's key surface (reference: conf/pointgroup.yaml,
scripts/train.py:25-39 `load_conf`).  OmegaConf is not available in this image; only attribute access is used
on the hot path (`cfg.model.m`, `cfg.cluster.cluster_radius`, ...), which this small tree provides."""
import os

import yaml


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def load_conf(*paths, overrides=None):
    """merge yaml files left to right (like OmegaConf.merge(conf/path.yaml, task.yaml))"""
    cfg = {}
    for p in paths:
        with open(p) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    if overrides:
        _merge(cfg, overrides)
    return _wrap(cfg)


def default_conf(name="pointgroup.yaml", overrides=None):
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")
    return load_conf(os.path.join(root, name), overrides=overrides)
