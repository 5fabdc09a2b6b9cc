"""Per-frame optimisation record (``trackers.pkl``).

Interface kept from the reference: the attribute names ``costs``, ``T``, ``dt``, ``num_iters``,
``dense``, ``p3d``, ``p3d_ids``, ``debug`` are what ``trackers.pkl`` consumers unpickle
(pixtrack/localization/tracker.py:5-16; r9 reads ``costs`` at pixloc_tracker_r9.py:251), and an
optimizer may still drive the per-iteration hook ``log_optim_iter(i=, T_init=, T=, T_delta=, cost=,
valid=)`` (:32-46).

The implementation is built around what the HIP path actually produces: ``pxt_lm_refine`` runs every
iteration of a level inside one kernel and hands back the level's whole record at once (masked-mean
cost, step size and pose per iteration - include/pixtrack_hip.h "Log").  ``record_level`` files such
a record in one go; the per-iteration hook is a thin adapter on the same store.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Sequence

import numpy as np
import torch

from .geometry import Pose


def _attach(tracker, refiner) -> None:
    """pixloc BaseTracker wiring: the refiner knows its tracker, every optimizer reports to it."""
    refiner.tracker = tracker
    opts = refiner.optimizer
    for opt in (opts if isinstance(opts, (tuple, list)) else [opts]):
        opt.logging_fn = tracker.log_optim_iter
        opt.level_logging_fn = tracker.record_level


class BaseTracker:
    def __init__(self, refiner):
        self.refiner = refiner
        _attach(self, refiner)

    def log_dense(self, **args):
        raise NotImplementedError

    def log_optim_done(self, **args):
        raise NotImplementedError

    def log_optim_iter(self, **args):
        raise NotImplementedError

    def record_level(self, T_init, costs, poses12, steps):
        raise NotImplementedError


class DebugTracker(BaseTracker):
    """debug >= 1 keeps the iteration record, debug >= 2 also the dense maps and the point set."""

    def __init__(self, refiner, debug=0):
        self.debug = int(debug)
        self.costs = []      # one list per optimised level: cost before each update
        self.T = []          # start pose of every level followed by the pose after each update
        self.dt = []         # translation magnitude of every update, all levels concatenated
        self.num_iters = []  # updates run per level
        self.dense = defaultdict(dict)
        self.p3d = None
        self.p3d_ids = None
        super().__init__(refiner)

    # -- bulk path: one call per level with the kernel's record --------------------------------
    def record_level(self, T_init: Pose, costs: Sequence[float], poses12, steps: Sequence[float]) -> None:
        """``costs[i]`` masked-mean cost before update i, ``poses12[i]`` the pose after it
        ([n,12] tensor), ``steps[i]`` the translation norm of update i."""
        n = len(costs)
        if self.debug < 1 or n == 0:
            return
        self.costs.append([np.asarray([c], np.float32) for c in costs])
        self.T.append(T_init.cpu())
        self.T.extend(Pose(poses12[i].clone()) for i in range(n))
        self.dt.extend(np.asarray([s], np.float32) for s in steps)
        self.num_iters.append(n)

    # -- per-iteration hook (the protocol a pixloc-style optimizer drives) ----------------------
    def log_optim_iter(self, *, i, T_init, T, T_delta, cost, valid, **_unused) -> None:
        if self.debug < 1:
            return
        if i == 0:  # a level starts: open its record
            self.costs.append([])
            self.num_iters.append(0)
            self.T.append(T_init.cpu())
        w = valid.to(cost.dtype)
        masked_mean = (w * cost).sum(-1) / w.sum(-1)
        self.costs[-1].append(np.atleast_1d(masked_mean.cpu().numpy()))  # (1,) per iteration, as record_level
        self.T.append(T.cpu())
        self.dt.append(np.atleast_1d(T_delta.magnitude()[1].cpu().numpy()))
        self.num_iters[-1] = i + 1

    # -- debug >= 2 extras ------------------------------------------------------------------------
    def log_dense(self, *, name, image_scale, image, features, weight, **_unused) -> None:
        if self.debug >= 2:
            self.dense[name][image_scale] = (image, [f.cpu() for f in features], [w.cpu()[0] for w in weight])

    def log_optim_done(self, *, p3d, p3d_ids, **_unused) -> None:
        if self.debug >= 2:
            self.p3d, self.p3d_ids = p3d, p3d_ids

    def __getstate__(self):
        # trackers.pkl must not drag the refiner (device buffers, native handles) along
        return {k: v for k, v in self.__dict__.items() if k != "refiner"}
