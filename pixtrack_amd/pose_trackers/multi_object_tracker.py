"""MultiObjectTracker -- K objects tracked in lock-step on ONE GPU.

The reference tracks one object per process: `pixloc_tracker_r9.py:287-318` builds one
``PixLocPoseTrackerR9`` from one ``config/<object>.sh`` and runs it over one video.  BASELINE configs[3]
("8 objects from config/*.sh tracked concurrently") on fewer than 8 GPUs means several such trackers per
GPU.  Running them one after the other (or as processes sharing the GPU) leaves the small stages of a
frame as they are: every tracker's deep UNet layers fill a third of the chip, every tracker's LM launch
pays its own inter-workgroup exchange and single-lane solve per iteration.  Here the K trackers keep
their own state and policy - each one IS a PixLocPoseTrackerR9, advanced through the same per-frame
methods - and only the device work of a frame step is merged:

* the 2 K images of a step (K reference renders, K masked queries) go through the UNet in ONE batched
  pass (``pxt_unet_forward_batch``): the 30x40 / 60x80 / decoder layers get 2 K x the workgroups;
* the K refinements run in ONE persistent launch (``pxt_lm_refine_batch``): workgroup i works on
  problem i mod K, an iteration's exchange and solve are paid once for all K;
* renders stay one renderer context per object (each object has its own NeRF), the render of step
  t + 1 queued behind the batched LM launch with the camera from the LM epilogue's slot, as in the
  one-object tracker.

A tracker in cold start (image scales [4, 1]: two dependent refinements) runs that frame by itself
through ``run_single_frame``; everything else is lock-step.  Per object the results are those of the
one-object tracker: bit for bit with ``per_image_plan=True`` and ``lm_workgroups`` equal to the solo
grid (the UNet layers then take a single image's tile / split-K plan, the LM folds the same number of
partials); with the defaults (batch-planned layers: no split-K where the batch fills the chip; 256 / K
workgroups per problem) equal up to fp32 summation order.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch

from .. import _lib
from ..optimizer import PixTrackOptimizer
from ..tracker import DebugTracker
from .pixloc_tracker_r9 import PixLocPoseTrackerR9


class _Group:
    """The objects whose device work is merged: one stream, one UNet context (workspace), one LM batch workspace."""

    def __init__(self, index, members, stream, model):
        self.index, self.members, self.stream, self.model = index, members, stream, model
        self.render_streams: List[torch.cuda.Stream] = []  # optional sub-streams of the group's queued renders
        self.batch_ws: Optional[torch.Tensor] = None
        self.render_ws: Optional[torch.Tensor] = None  # parameter records of the batched render chain
        self.unet_done: Optional[torch.cuda.Event] = None
        self.pend = None  # what _enqueue left for _finish


class MultiObjectTracker:
    def __init__(self, trackers: Sequence[PixLocPoseTrackerR9], lm_workgroups: int = 0, per_image_plan: bool = False,
                 max_unet_batch: int = _lib.PXT_UNET_MAX_BATCH, n_groups: int = 1, render_pipelines: int = -1,
                 render_streams: int = 1, batch_renders: Optional[bool] = None):
        """``n_groups`` > 1: the trackers are dealt to that many groups, each with its own stream, batched UNet pass and
        batched LM launch per step; the groups' UNet passes take turns (an event token), so that one group's MFMA-bound
        UNet pass runs beside the other group's latency-bound renders instead of beside its UNet pass."""
        if not trackers:
            raise ValueError("MultiObjectTracker needs at least one tracker")
        self.trackers: List[PixLocPoseTrackerR9] = list(trackers)
        self.device = self.trackers[0].device
        for tr in self.trackers:
            if tr.device != self.device:
                raise _lib.PxtError("lock-step trackers share one device")
            # the step below is PixLocPoseTrackerR9.refine() in pieces: a tracker with its own refine() (the YCB policy:
            # ground-truth-gated updates, pixloc_tracker_ycb.py:241-295) would silently run r9's policy here
            if type(tr).refine is not PixLocPoseTrackerR9.refine or type(tr)._frame_policy is not PixLocPoseTrackerR9._frame_policy:
                raise _lib.PxtError(f"{type(tr).__name__} overrides refine(): lock-step tracking implements PixLocPoseTrackerR9's policy only")
        # one UNet context runs every image of a group's step: the trackers must hold the same checkpoint
        # (pixloc_megadepth is one network for all objects; reference pixloc_pose_refiners.py:49-60)
        sig = getattr(self.trackers[0].localizer.extractor.model, "weights_signature", None)
        for tr in self.trackers[1:]:
            if getattr(tr.localizer.extractor.model, "weights_signature", None) != sig:
                raise _lib.PxtError("lock-step trackers must share one UNet checkpoint")
        self.groups: List[_Group] = []
        self.render_pipelines = int(render_pipelines)  # -1: one pipeline per render with several groups, else the default
        self.render_streams = int(render_streams)      # sub-streams per group for the queued renders (experiment: 1 = none)
        # a group's queued renders as ONE chain of launches carrying the rays of all its objects (pxt_ngp_render_frame_batch;
        # bit for bit the single renders); PXT_BATCH_RENDERS=0: one chain per object, one after the other
        self.batch_renders = (os.environ.get("PXT_BATCH_RENDERS", "1") != "0") if batch_renders is None else bool(batch_renders)
        self.set_groups(n_groups)
        self.lm_workgroups = int(lm_workgroups)      # grid per problem of the batched launch; 0: the library's default
        self.per_image_plan = bool(per_image_plan)   # UNet layers planned as for one image (bit-identity with solo runs)
        self.max_unet_batch = int(max_unet_batch)
        self.steps = 0
        self.solo_frames = 0       # frames that ran through run_single_frame (cold starts, several references)
        self.lockstep_frames = 0
        self.timing = None         # set to {} to collect HIP-event pairs per phase (bench.py's untimed diagnostic pass)

    def set_groups(self, n_groups: int) -> None:
        """(Re)deals the trackers to ``n_groups`` groups, tracker k to group k mod n_groups.  Synchronises the device: a
        tracker's tensors of the last step were produced on its old group's stream."""
        n_groups = max(1, min(int(n_groups), len(self.trackers)))
        if any(g.pend for g in self.groups):
            raise _lib.PxtError("set_groups between steps only")
        torch.cuda.synchronize(self.device)
        self.groups = []
        for g in range(n_groups):
            members = list(range(g, len(self.trackers), n_groups))
            stream = None if n_groups == 1 else _group_stream(self.device, g)
            self.groups.append(_Group(g, members, stream, self.trackers[members[0]].localizer.extractor.model))
            if n_groups > 1 and self.render_streams > 1:
                self.groups[-1].render_streams = [_group_stream(self.device, 100 + 10 * g + j) for j in range(self.render_streams)]
        self.model = self.groups[0].model
        self._last_unet_done = None
        # With several groups the concurrency a render needs comes from the OTHER group's UNet pass: one pipeline per
        # render (no fork / join between two half-renders) is faster there (898 -> 931 frames/s for eight objects); one
        # group keeps the renderer's default of two pipelines over the two halves of the rays.  Bit-identical either way.
        pipes = self.render_pipelines if self.render_pipelines >= 0 else (1 if n_groups > 1 else 0)
        for tr in self.trackers:
            tr.testbed.set_pipelines(pipes)

    # ------------------------------------------------------------------ helpers
    def _lm_batch_ws(self, grp: _Group) -> torch.Tensor:
        if grp.batch_ws is None:
            need = int(_lib.lib().pxt_lm_batch_workspace_bytes(_lib.PXT_LM_MAX_BATCH))
            grp.batch_ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return grp.batch_ws

    def _mark(self, name: str):
        if self.timing is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.timing.setdefault(name, []).append(e)
        return e

    def _unet_batch(self, grp: _Group, jobs) -> None:
        """jobs: [(extractor, image, scale, mask, normalize)] -> every pyramid computed in batched passes (images of
        one size per pass, at most max_unet_batch each) and handed to its extractor through preload()."""
        prepared = []
        for ex, image, scale, mask, normalize in jobs:
            img, rest_mask, scales = ex.prepared(image, scale, mask)
            prepared.append((img, rest_mask, scales))
        sizes = {}
        for i, (img, _m, _s) in enumerate(prepared):
            sizes.setdefault((int(img.shape[0]), int(img.shape[1])), []).append(i)
        cur = torch.cuda.current_stream(self.device)
        if len(self.groups) > 1 and self._last_unet_done is not None:
            cur.wait_event(self._last_unet_done)  # the groups' UNet passes take turns
        grp.model.set_batch_plan(self.per_image_plan)
        try:
            for idx in sizes.values():
                for a in range(0, len(idx), self.max_unet_batch):
                    part = idx[a:a + self.max_unet_batch]
                    outs = grp.model.forward_packed_batch([(prepared[i][0], prepared[i][1], jobs[i][4]) for i in part])
                    for i, maps in zip(part, outs):
                        ex, image, scale, mask, normalize = jobs[i]
                        ex.preload(image, scale, mask, normalize, maps, prepared[i][2])
        finally:
            grp.model.set_batch_plan(False)
        if len(self.groups) > 1:
            grp.unet_done = torch.cuda.Event()
            grp.unet_done.record(cur)
            self._last_unet_done = grp.unet_done

    # ------------------------------------------------------------------ one step = one frame of every object
    def run_single_frames(self, frames) -> List[bool]:
        """frames[k] = (path, image) of tracker k (None: that tracker sits this step out).  The lock-step counterpart
        of PoseTracker.run_single_frame (reference base_pose_tracker.py:24-30) for K trackers at once."""
        assert len(frames) == len(self.trackers)
        out: List[Optional[bool]] = [None] * len(frames)
        self._mark("step_begin")
        for grp in self.groups:
            if grp.stream is None:
                self._enqueue(grp, frames, out)
            else:
                with torch.cuda.stream(grp.stream):
                    self._enqueue(grp, frames, out)
        for grp in self.groups:
            if grp.stream is None:
                self._finish(grp, frames, out)
            else:
                with torch.cuda.stream(grp.stream):
                    self._finish(grp, frames, out)
        self._mark("step_end")
        self.steps += 1
        return out

    def _enqueue(self, grp: _Group, frames, out) -> None:
        live = []
        grp.pend = []
        # ---- phase A: per-object frame set-up (policy head, mask + reference render or the render queued last step)
        for k in grp.members:
            tr, frame = self.trackers[k], frames[k]
            if frame is None:
                continue
            scales = tr.steady_multiscale if tr.success else tr.localizer.refiner.conf.multiscale
            if tr.cold_start or len(tr.reference_ids) != 1 or list(scales or [1]) != [1]:
                tr.run_single_frame(frame)
                out[k] = bool(tr.success)
                self.solo_frames += 1
                continue
            tr._frame_setup(frame, lockstep=True)
            ref_u8 = tr.get_reference_image(tr.pose)
            tr._fused_reference = (tr.pose, ref_u8)  # get_dynamic_id asks for it again: the same tensor object
            live.append((k, tr, frame, ref_u8))
        self._mark("renders_enqueued")
        if not live:
            return
        # ---- phase B: the step's 2 K images through the UNet in batched passes
        jobs = []
        for k, tr, (path, image), ref_u8 in live:
            ex = tr.localizer.refiner.feature_extractor
            # extract_reference_features encodes the window of the render its points depend on (refiner.reference_window)
            # - but a reference render of the query's own size goes through the UNet with everybody else's: a window
            # would take it out of the batch (one 14-image pass became a 13-image pass + two small ones: 4.5 -> 5.3 ms)
            refiner = tr.localizer.refiner
            if tuple(ref_u8.shape[:2]) == tuple(image.shape[:2]):
                refiner._window_memo = (ref_u8, tr.pose, ref_u8, None)
            ref_in, _win = refiner.reference_window(tr.reference_ids, tr.pose, ref_u8)
            jobs.append((ex, ref_in, 1, None, False))                               # extract_reference_features
            jobs.append((ex, image, 1, tr.localizer.refiner.query_mask, True))      # refine_query_pose's query pass
        self._unet_batch(grp, jobs)
        self._mark("unet_enqueued")
        # ---- phase C: per object, the reference's bookkeeping + sparse sampling + the LM problem
        pend = []
        for k, tr, (path, image), ref_u8 in live:
            refiner = tr.localizer.refiner
            tr.dynamic_id = tr.get_dynamic_id(tr.pose)
            ref_id = tr.reference_ids[0]
            pose_init = tr._frame_pose_init()
            dbg = DebugTracker(refiner, tr.debug)
            status, x = refiner.begin_refine(path, tr.camera, pose_init, [ref_id], image, tr.dynamic_id)
            refiner.feature_extractor.unstage()  # (drops a preloaded pyramid nobody asked for)
            pend.append((k, tr, path, ref_id, dbg, status, x))
        self._mark("sampling_enqueued")
        # ---- phase D: ONE persistent launch for every refinement, then each object's next render behind it
        probs = [x for (_k, _tr, _p, _r, _d, status, x) in pend if status == "lm"]
        handles = []
        if probs:
            conf = probs[0]["conf"]
            for pr in probs[1:]:
                if bytes(pr["conf"]) != bytes(conf):
                    raise _lib.PxtError("lock-step trackers must share the optimizer configuration")
            conf.n_workgroups = self.lm_workgroups
            if len(self.groups) > 1:
                # the groups' launches may run side by side: together they must fit the device's resident workgroups
                # (one 8-wave workgroup per CU: pxt_lm.hip), or both would sit partly resident and time out - also when the
                # caller asked for a grid per problem (lm_workgroups > 0: clamped to the groups' share)
                share = max(8, _device_cus() // (len(self.groups) * len(probs)) // 8 * 8)
                conf.n_workgroups = share if conf.n_workgroups <= 0 else min(conf.n_workgroups, share)
            handles = PixTrackOptimizer.refine_levels_batch(probs, conf, self._lm_batch_ws(grp), pool_key=grp.index)
        self._mark("lm_enqueued")
        it = iter(handles)
        grp.pend = [(k, tr, path, ref_id, dbg, status, x, next(it) if status == "lm" else None)
                    for (k, tr, path, ref_id, dbg, status, x) in pend]
        subs = grp.render_streams
        if subs:  # the group's queued renders dealt to sub-streams behind the LM launch, joined before the next UNet pass
            cur = torch.cuda.current_stream(self.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            for st in subs:
                st.wait_event(fork)
        j = 0
        batched = set()
        if self.batch_renders and not subs:
            batched = self._batched_renders_ahead(grp)
        for k, tr, path, ref_id, dbg, status, x, handle in grp.pend:
            hook = getattr(tr.localizer.refiner, "after_lm_enqueued", None)
            if handle is not None and hook is not None and k not in batched:
                if subs:
                    with torch.cuda.stream(subs[j % len(subs)]):
                        hook(handle)
                    if tr._ahead is not None:  # (mask, 8-bit reference image: read on the group's stream next step)
                        for t in tr._ahead[1:3]:
                            if torch.is_tensor(t):
                                t.record_stream(cur)
                    j += 1
                else:
                    hook(handle)
        if subs:
            for st in subs:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
        self._mark("ahead_enqueued")

    def _batched_renders_ahead(self, grp: _Group) -> set:
        """The group's queued renders (behind the batched LM launch, cameras from its epilogue's slots) in one batched
        chain per (spp) - returns the members served; the others take their own _render_ahead()."""
        from ..ngp import Testbed

        by_spp = {}
        for k, tr, path, ref_id, dbg, status, x, handle in grp.pend:
            hook = getattr(tr.localizer.refiner, "after_lm_enqueued", None)
            if handle is None or hook is None or getattr(hook, "__func__", None) is not PixLocPoseTrackerR9._render_ahead:
                continue
            req = tr._render_ahead_request()
            if req is not None:
                by_spp.setdefault(req[2], []).append((k, tr, req))
        served = set()
        for spp, items in by_spp.items():
            if len(items) < 2:
                continue
            if grp.render_ws is None:
                grp.render_ws = torch.empty(Testbed.batch_workspace_bytes(_lib.PXT_NGP_MAX_BATCH), dtype=torch.uint8,
                                            device=self.device)
            for a in range(0, len(items), _lib.PXT_NGP_MAX_BATCH):
                part = items[a:a + _lib.PXT_NGP_MAX_BATCH]
                outs = Testbed.render_frame_batch_device([tr.testbed for _k, tr, _r in part], [(r[0], r[1]) for _k, _t, r in part],
                                                         spp, mode=2, from_slot=True, workspace=grp.render_ws)
                for (k, tr, _r), out in zip(part, outs):
                    tr._render_ahead_accept(out)
                    served.add(k)
        return served

    def _finish(self, grp: _Group, frames, out) -> None:
        # ---- phase E: results, per-object policy (cost gate, pose update, history), the loop's tail
        for k, tr, path, ref_id, dbg, status, x, handle in grp.pend or []:
            refiner = tr.localizer.refiner
            ret = refiner.finish_refine(x, handle.result()) if handle is not None else x
            ok = tr._frame_policy(path, {ref_id: ret}, {ref_id: tr._frame_cost()}, {ref_id: dbg})
            if not ok:
                tr.relocalize(frames[k])
            tr.update_reference_ids()
            out[k] = bool(ok)
            self.lockstep_frames += 1
        grp.pend = None

    def run(self, frame_iterators, max_frames=float("inf")):
        """Advances every tracker over its own frame iterator (``get_query_frame_iterator`` objects or any iterables of
        (path, image)), one frame of each per step, until all are exhausted."""
        its = [iter(f) for f in frame_iterators]
        n = 0
        while n < max_frames:
            frames = [next(it, None) for it in its]
            if all(f is None for f in frames):
                break
            self.run_single_frames(frames)
            n += 1
        return n


_GROUP_STREAMS = {}


def _group_stream(device, g: int) -> "torch.cuda.Stream":
    """Group g's stream on ``device``: made once per process (HIP deals streams to its hardware queues in creation order:
    fresh streams per tracker would make the groups' overlap depend on how many trackers the process has seen)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), g)
    if key not in _GROUP_STREAMS:
        _GROUP_STREAMS[key] = torch.cuda.Stream(device)
    return _GROUP_STREAMS[key]


def _device_cus() -> int:
    import ctypes as C

    n = C.c_int(0)
    _lib.check(_lib.lib().pxt_device_cus(C.byref(n)), "pxt_device_cus")
    return int(n.value)


def parse_config_sh(path) -> dict:
    """The two settings a reference `config/<object>.sh` exports for the tracker: OBJ_AABB and UPRIGHT_REF_IMG
    (e.g. /root/reference/config/premier_protein.sh:14; read as text, the file is not executed)."""
    import re

    out = {}
    for line in open(path):
        m = re.match(r"\s*(?:export\s+)?(OBJ_AABB|UPRIGHT_REF_IMG)\s*=\s*(.*?)\s*$", line)
        if m:
            out[m.group(1)] = m.group(2).strip().strip("'\"")
    return out


def main(argv=None):
    """Several objects, one GPU: the reference's tracker command line (pixloc_tracker_r9.py:288-318) once per object, in
    lock-step.  Every list takes one entry per object; an object's OBJ_AABB / UPRIGHT_REF_IMG come from --config (a
    reference config/<object>.sh) or from --obj_aabb / --upright_ref_img.  Outputs per object as r9's CLI writes them:
    <out_dir>/poses.pkl, <out_dir>/trackers.pkl, and the `Cache hits / Done` lines per object."""
    import argparse
    import gc
    import os
    from pathlib import Path

    import numpy as np

    from .pixloc_tracker_r9 import _dump

    ap = argparse.ArgumentParser()
    ap.add_argument("--object_path", type=Path, nargs="+", required=True)
    ap.add_argument("--query", type=Path, nargs="+", required=True)
    ap.add_argument("--out_dir", type=Path, nargs="+", required=True)
    ap.add_argument("--config", type=Path, nargs="*", default=None, help="config/<object>.sh per object")
    ap.add_argument("--obj_aabb", nargs="*", default=None)
    ap.add_argument("--upright_ref_img", nargs="*", default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--pixloc_pickles", action="store_true")
    args = ap.parse_args(argv)
    K = len(args.object_path)
    if len(args.query) != K or len(args.out_dir) != K:
        ap.error("--object_path, --query and --out_dir take one entry per object")
    if torch.cuda.is_available():
        from ..parallel import bind_to_device_numa

        bind_to_device_numa(0)
    trackers = []
    for k in range(K):
        conf = parse_config_sh(args.config[k]) if args.config else {}
        if args.obj_aabb:
            conf["OBJ_AABB"] = args.obj_aabb[k]
        if args.upright_ref_img:
            conf["UPRIGHT_REF_IMG"] = args.upright_ref_img[k]
        if "OBJ_AABB" not in conf or "UPRIGHT_REF_IMG" not in conf:
            ap.error(f"object {k}: OBJ_AABB and UPRIGHT_REF_IMG are needed (--config, or --obj_aabb / --upright_ref_img)")
        os.environ.update(OBJ_AABB=conf["OBJ_AABB"], UPRIGHT_REF_IMG=conf["UPRIGHT_REF_IMG"])  # read in the constructor (:77, :85)
        obj = args.object_path[k]
        os.makedirs(args.out_dir[k], exist_ok=True)
        trackers.append(PixLocPoseTrackerR9(object_path=str(obj), data_path=str(obj / "pixtrack/pixsfm/dataset"),
                                            eval_path=str(args.out_dir[k]), loc_path=str(obj / "pixtrack/aug_nerf_sfm"),
                                            debug=args.debug))
    multi = MultiObjectTracker(trackers, n_groups=args.groups)
    its = [tr.get_query_frame_iterator(q, args.frames if args.frames is not None else np.inf)
           for tr, q in zip(trackers, args.query)]
    gc.collect()
    gc.freeze()
    gc.disable()
    try:
        multi.run(its)
    finally:
        gc.enable()
    for tr in trackers:
        tr.save_poses(args.pixloc_pickles)
        print("Cache hits: %d, misses: %d" % (tr.hits, tr.misses))
        _dump(tr.pose_tracker_history, os.path.join(tr.eval_path, "trackers.pkl"), args.pixloc_pickles)
    print("Done")


if __name__ == "__main__":
    main()
