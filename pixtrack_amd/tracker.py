"""DebugTracker (reference pixtrack/localization/tracker.py:5-46) and the pixloc BaseTracker
hook mechanics it relies on: the tracker registers itself on the refiner and installs
``logging_fn`` on every optimizer; per iteration it records the masked-mean cost, which the
r9 tracker's success gate consumes (pixloc_tracker_r9.py:251-263)."""
from __future__ import annotations

from collections import defaultdict


class BaseTracker:
    def __init__(self, refiner):
        self.refiner = refiner
        refiner.tracker = self
        opts = refiner.optimizer if isinstance(refiner.optimizer, (tuple, list)) else [refiner.optimizer]
        for opt in opts:
            opt.logging_fn = self.log_optim_iter

    def log_dense(self, **args):
        raise NotImplementedError

    def log_optim_done(self, **args):
        raise NotImplementedError

    def log_optim_iter(self, **args):
        raise NotImplementedError


class DebugTracker(BaseTracker):
    def __init__(self, refiner, debug=0):
        super().__init__(refiner)
        self.dense = defaultdict(dict)
        self.costs = []
        self.T = []
        self.dt = []
        self.p3d = None
        self.p3d_ids = None
        self.num_iters = []
        self.debug = debug

    def log_dense(self, **args):
        if self.debug < 2:
            return
        feats = [f.cpu() for f in args["features"]]
        weights = [w.cpu()[0] for w in args["weight"]]
        self.dense[args["name"]][args["image_scale"]] = (args["image"], feats, weights)

    def log_optim_done(self, **args):
        if self.debug < 2:
            return
        self.p3d = args["p3d"]
        self.p3d_ids = args["p3d_ids"]

    def log_optim_iter(self, **args):
        if self.debug < 1:
            return
        if args["i"] == 0:  # new scale or level
            self.costs.append([])
            self.T.append(args["T_init"].cpu())
            self.num_iters.append(None)
        valid = args["valid"].float()
        cost = (valid * args["cost"]).sum(-1) / valid.sum(-1)
        self.costs[-1].append(cost.cpu().numpy())
        self.dt.append(args["T_delta"].magnitude()[1].cpu().numpy())
        self.num_iters[-1] = args["i"] + 1
        self.T.append(args["T"].cpu())

    # pickling trackers.pkl must not drag the refiner (device buffers) along
    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("refiner", None)
        return d
