"""Attribute-access config dicts.  The reference uses OmegaConf (absent here) for the
nested dicts of pixloc_tracker_r9.py:43-58 and pixloc_pose_refiners.py:46-52,122-127;
this keeps the same keys, `conf.a.b` access, in-place mutation
(`refiner.conf.multiscale = [4, 1]`, pixloc_tracker_r9.py:219) and a recursive merge."""
from __future__ import annotations

import copy
from typing import Any, Mapping


class Conf(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, Mapping) and not isinstance(v, Conf):
            v = Conf(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __deepcopy__(self, memo):
        return Conf({k: copy.deepcopy(v, memo) for k, v in self.items()})


def merge(*confs: Mapping[str, Any]) -> Conf:
    """Recursive right-biased merge (OmegaConf.merge semantics for plain dicts)."""
    out = Conf()
    for c in confs:
        if c is None:
            continue
        for k, v in c.items():
            if isinstance(v, Mapping) and isinstance(out.get(k), Mapping):
                out[k] = merge(out[k], v)
            else:
                out[k] = copy.deepcopy(v)
    return out
