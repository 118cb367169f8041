"""Attribute-style configuration with the reference's key surface (reference: conf/pointgroup.yaml,
scripts/train.py:25-39 `load_conf`).  OmegaConf is not available in this image; only attribute access is used
on the hot path (`cfg.model.m`, `cfg.cluster.cluster_radius`, ...), which this small tree provides."""
import os
import re

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


_REF = re.compile(r"\$\{([A-Za-z0-9_.]+)\}")


def _resolve(cfg):
    """`${A.B}` -> the value at that dotted path of the merged tree (OmegaConf interpolation as conf/path.yaml uses it:
    whole-string references keep their type, embedded ones are substituted as text; chains are followed)."""
    def lookup(path):
        node = cfg
        for part in path.split("."):
            node = node[part]
        return node

    def walk(x, depth=0):
        if isinstance(x, dict):
            return {k: walk(v) for k, v in x.items()}
        if isinstance(x, list):
            return [walk(v) for v in x]
        if isinstance(x, str) and "${" in x:
            if depth > 16:
                raise ValueError("interpolation cycle at %r" % x)
            m = _REF.fullmatch(x)
            if m:
                return walk(lookup(m.group(1)), depth + 1)
            return walk(_REF.sub(lambda m: str(walk(lookup(m.group(1)), depth + 1)), x), depth + 1)
        return x
    return walk(cfg)


def load_conf(*paths, overrides=None):
    """merge yaml files left to right (like OmegaConf.merge(conf/path.yaml, task.yaml)), then resolve `${...}`"""
    cfg = {}
    for p in paths:
        with open(p) as f:
            _merge(cfg, yaml.safe_load(f) or {})
    if overrides:
        _merge(cfg, overrides)
    return _wrap(_resolve(cfg))


CONF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")


def default_conf(name="pointgroup.yaml", overrides=None):
    """conf/path.yaml merged under conf/<name>, as scripts/train.py:26-28 does"""
    return load_conf(os.path.join(CONF_DIR, "path.yaml"), os.path.join(CONF_DIR, name), overrides=overrides)
