"""Host mirror of the optimizer interface pixtrack drives.

Reference: ``PixTrackOptimizer(LearnedOptimizer)`` -- pixtrack/optimizers/
pixtrack_optimizer.py:5-18 -- objects are created by pixloc's ``load_experiment`` and
class-swizzled at pixtrack/localization/pixloc_pose_refiners.py:71-72; pixtrack then
uses ``opt.run(p3d, F_ref, F_q, T_init, camera, W_ref_query=...) -> (T, failed)``
(through pixloc BaseRefiner, :260-262), ``opt.interpolator(feats, p2d)`` (:351),
``opt.conf.{grad,dt,dR}_stop_criteria`` / ``opt.training`` (pixtrack_optimizer.py:8-14)
and ``opt.logging_fn`` (tracker hook, pixtrack/localization/tracker.py:32-46).

Here the whole iteration loop runs inside ONE persistent HIP kernel
(csrc/pxt_lm.hip via ``pxt_lm_refine``); this class only packs arguments and replays
the per-iteration log into ``logging_fn`` afterwards.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import threading
import re
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .ops import ops
from .geometry import Camera, Pose
from .utils.conf import Conf, merge

LOSS_KINDS = {"squared": 0, "huber": 1, "barron": 2}


def parse_loss_fn(spec: str) -> Tuple[int, float, float]:
    """pixloc names losses by expression: 'squared_loss', 'scaled_loss(huber_loss, a)',
    'scaled_barron(alpha, a)'.  Returns (kind, alpha, scale)."""
    spec = spec.replace(" ", "")
    if spec == "squared_loss":
        return 0, 2.0, 1.0
    m = re.fullmatch(r"scaled_barron\(([-+.\de]+),([-+.\de]+)\)", spec)
    if m:
        return 2, float(m.group(1)), float(m.group(2))
    m = re.fullmatch(r"scaled_huber\(([-+.\de]+)\)", spec)
    if m:
        return 1, 0.0, float(m.group(1))
    if spec == "huber_loss":
        return 1, 0.0, 1.0
    raise ValueError(f"unsupported loss_fn '{spec}'")


def round4(n: int) -> int:
    return (n + 3) // 4 * 4


def cstride_for(C_: int) -> int:
    """Channel stride of the packed HWC records: C descriptor + 1 confidence, padded to 4."""
    return round4(C_ + 1)


@dataclass
class LevelPack:
    """One pyramid level in the layout the kernels read (include/pixtrack_hip.h)."""

    fmap: torch.Tensor  # [h, w, cstride] float32, descriptor normalised, conf at [C]
    fref: torch.Tensor  # [N, cstride]   float32, descriptor normalised, W_ref at [C]
    C: int
    camera: Camera  # query camera scaled to this level
    lambda_: torch.Tensor  # [6] host float32


@dataclass
class LMResult:
    T: Pose
    failed: bool
    iters: List[int]
    costs: List[List[float]]  # per level, per iteration masked-mean cost
    log: torch.Tensor  # host [n_levels, num_iters, 20]
    total_iters: int = 0


class Interpolator:
    """pixloc Interpolator surface: ``obs, mask, grads = interp(tensor[C,h,w], pts[N,2])``.
    Implemented on the sparse-sampling kernel with an identity camera (u = x, v = y)."""

    def __init__(self, mode: str = "linear", pad: int = 4):
        assert mode == "linear", "only bilinear interpolation is on the tracking path"
        self.mode = mode
        self.pad = pad

    def __call__(self, tensor: torch.Tensor, pts: torch.Tensor, return_gradients: bool = False):
        if return_gradients:
            raise NotImplementedError("gradients are only formed inside the fused LM kernel")
        _lib.require_gpu(tensor, "tensor")
        Cc, h, w = tensor.shape
        cs = round4(Cc + 1)
        fmap = torch.zeros(h, w, cs, device=tensor.device, dtype=torch.float32)
        fmap[..., :Cc] = tensor.permute(1, 2, 0)
        # treat ALL Cc channels as "descriptor" channels of a (Cc4 = round4) record
        obs, valid = sample_sparse_points2d(fmap, Cc, pts.to(tensor.device, torch.float32), self.pad)
        # third output (pixloc: gradients, None unless asked for): gradients exist only inside the
        # fused LM kernel; asking for them raises above rather than returning made-up values
        return obs, valid, None


def sample_sparse_points2d(fmap: torch.Tensor, Cc: int, pts: torch.Tensor, pad: int):
    """Bilinear sample an HWC map at pixel coordinates through pxt_sample_sparse."""
    h, w, cs = fmap.shape
    N = pts.shape[0]
    # round the descriptor width down to a multiple of 4 is not possible in general, so
    # sample a padded record: channels [0, C4) are "descriptor", the record is C4+4 wide.
    C4 = round4(Cc)
    if cs < C4 + 4:
        fm = torch.zeros(h, w, C4 + 4, device=fmap.device, dtype=torch.float32)
        fm[..., :cs] = fmap
        fmap, cs = fm, C4 + 4
    p3d = torch.cat([pts, torch.ones(N, 1, device=pts.device)], -1).contiguous()
    out = torch.empty(N, cs, device=pts.device, dtype=torch.float32)
    valid = torch.empty(N, device=pts.device, dtype=torch.uint8)
    # identity pose and a unit camera: (x, y, 1) projects to pixel (x, y)
    ops.sample_sparse(p3d, [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, 0, 0, 0], [fmap], [C4],
                      [float(w), float(h), 1.0, 1.0, 0, 0, 0, 0, 0, 0], [0], int(pad), False, [out], valid)
    return out[:, :Cc], valid.bool()


class DampingNet:
    """Learned constant damping: lambda = 10^(lo + sigmoid(const) (hi - lo))."""

    def __init__(self, conf, num_params: int = 6):
        self.conf = conf
        self.const = torch.zeros(num_params)

    def __call__(self) -> torch.Tensor:
        lo, hi = self.conf.log_range
        return 10.0 ** (lo + torch.sigmoid(self.const.float().cpu()) * (hi - lo))

    forward = __call__


class PixTrackOptimizer:
    """Same name and public surface as the reference class; HIP-only implementation."""

    default_conf = dict(
        num_iters=100,
        loss_fn="scaled_barron(0, 0.1)",
        jacobi_scaling=False,
        normalize_features=False,
        lambda_=0.0,
        interpolation=dict(mode="linear", pad=4),
        # pixtrack passes a TOP-LEVEL optimizer.pad = 1 (pixloc_tracker_r9.py:48).  Decision (DESIGN.md
        # section 4, pinned by tests/test_host_conf.py): it is honoured and becomes interpolation.pad,
        # as pixloc's BaseModel.__init__ does for its own run_*.py configs ("pad" in conf and not
        # in default_conf -> conf.interpolation = {pad: ...}; upstream recall, pixloc is not in the
        # reference tree).  An explicit interpolation.pad wins when the top-level key is absent.
        pad=None,
        grad_stop_criteria=1e-4,
        dt_stop_criteria=5e-3,
        dR_stop_criteria=5e-2,
        damping=dict(type="constant", log_range=[-6, 5]),
        learned_damping=True,
        min_valid=10,
        n_workgroups=0,
        lm_path=0,  # 0: automatic (one-round levels keep their points in registers); 2: always the general path
        spin_limit=0,  # polls before an inter-workgroup wait gives up (0: the library's default; tests force a time-out with 1)
    )

    def __init__(self, conf=None, device: Optional[torch.device] = None):
        self.conf = merge(self.default_conf, conf or {})
        assert not self.conf.jacobi_scaling and not self.conf.normalize_features
        pad = self.conf.pad if self.conf.pad is not None else self.conf.interpolation.pad
        self.interpolator = Interpolator(self.conf.interpolation.mode, int(pad))
        self.dampingnet = DampingNet(self.conf.damping)
        self.training = False
        self.logging_fn: Optional[Callable] = None        # per-iteration hook (pixloc protocol)
        self.level_logging_fn: Optional[Callable] = None  # whole-level record (DebugTracker.record_level)
        self.device = device
        self._ws = None
        self._ws_dev = None

    # ---- pixloc module-ish helpers ---------------------------------------
    def eval(self):
        self.training = False
        return self

    def to(self, device):
        self.device = torch.device(device)
        return self

    def load_state_dict(self, sd):
        self.dampingnet.const = torch.as_tensor(sd["dampingnet.const"]).float().cpu()

    def state_dict(self):
        return {"dampingnet.const": self.dampingnet.const.clone()}

    # ---- stop rule ------------------------------------------------------------
    def early_stop(self, *, i=0, T_delta=None, grad=None, **_unused) -> bool:
        """Host statement of the stop test ``lm_refine_kernel`` applies after EVERY update
        (interface: pixtrack/optimizers/pixtrack_optimizer.py:5-18, which moves pixloc's
        every-10th-iteration check to every iteration).  In inference mode a batch is finished
        once each element either took a step below both step thresholds (translation norm and
        rotation angle in degrees) or has a gradient norm below ``grad_stop_criteria``."""
        if self.training:
            return False
        rot_deg, trans = T_delta.magnitude()
        return self.converged(rot_deg, trans, torch.linalg.vector_norm(grad.detach(), dim=-1))

    def converged(self, rot_deg, trans, grad_norm) -> bool:
        """Same test on plain numbers / arrays, e.g. columns 2..4 of the kernel's iteration log
        (include/pixtrack_hip.h): dR [deg], dt, ||g||."""
        c = self.conf
        tiny_step = (torch.as_tensor(trans) < c.dt_stop_criteria) & (torch.as_tensor(rot_deg) < c.dR_stop_criteria)
        flat = torch.as_tensor(grad_norm) < c.grad_stop_criteria
        return bool(torch.all(tiny_step | flat))

    def log(self, **args):
        if self.logging_fn is not None:
            self.logging_fn(**args)

    # ---- native call ------------------------------------------------------
    def _workspace(self, device):
        if self._ws is None or self._ws_dev != device:
            n = int(_lib.lib().pxt_lm_workspace_bytes())
            self._ws = torch.zeros(n, dtype=torch.uint8, device=device)
            self._ws_dev = device
        return self._ws

    def native_conf(self) -> _lib.LmConf:
        kind, alpha, scale = parse_loss_fn(self.conf.loss_fn)
        c = _lib.LmConf()
        c.num_iters = int(self.conf.num_iters)
        c.pad = int(self.interpolator.pad)
        c.loss, c.loss_alpha, c.loss_scale = kind, alpha, scale
        c.grad_stop = float(self.conf.grad_stop_criteria)
        c.dt_stop = float(self.conf.dt_stop_criteria)
        c.dR_stop = float(self.conf.dR_stop_criteria)
        c.min_valid = int(self.conf.min_valid)
        c.n_workgroups = int(self.conf.n_workgroups)
        c.spin_limit = int(self.conf.get("spin_limit", 0))
        c.path = int(self.conf.get("lm_path", 0))
        return c

    @staticmethod
    def refine_levels(
        p3d: torch.Tensor,
        levels: Sequence[LevelPack],
        T_init: Pose,
        conf: _lib.LmConf,
        workspace: torch.Tensor,
        mask: Optional[torch.Tensor] = None,
        want_log: bool = True,
        camera=None,
    ) -> "PendingLM":
        """Enqueue the fused multi-level refinement (levels in EXECUTION order,
        coarse -> fine).  Returns a handle; ``.result()`` synchronises.
        ``camera`` = (conv27, [renderer camera slots], cam_out pinned record or None): the kernel's epilogue also
        converts the final pose into the NeRF renderer's camera and stores it there (pxt_lm_refine_cam), so that a
        render queued behind this launch needs no conversion launch."""
        _lib.require_gpu(p3d, "p3d")
        dev = p3d.device
        n_levels = len(levels)
        assert 1 <= n_levels <= _lib.PXT_MAX_LEVELS
        cams, lambdas, ndist = [], [], []
        for lp in levels:
            assert lp.fref.shape == (p3d.shape[0], lp.fmap.shape[2])
            cams += lp.camera.as10().tolist()
            ndist.append(int(lp.camera._data.shape[-1] - 6))
            lambdas += lp.lambda_.float().tolist()
        # One buffer for the output record and the iteration log, in pinned host memory: the
        # kernel's (write-only) stores land on the host directly, so the frame's critical path
        # has no device->host copy to enqueue and wait for after the kernel - only the event.
        nh = 16 + _lib.PXT_MAX_LEVELS
        n_log = n_levels * conf.num_iters * _lib.PXT_LM_LOG_STRIDE if want_log else 0
        buf = _pinned_record(nh + n_log)
        buf[:nh].zero_()
        p3d = p3d.to(torch.float32).contiguous()
        if mask is not None:
            mask = mask.to(dev, torch.uint8).contiguous()
        T0 = T_init.as12().detach().cpu().reshape(-1).tolist() if hasattr(T_init, "as12") else [float(x) for x in T_init]
        ops.lm_refine(p3d, mask, [lp.fmap for lp in levels], [lp.fref for lp in levels], [int(lp.C) for lp in levels],
                      cams, ndist, lambdas, T0, conf.num_iters, conf.pad, conf.loss, conf.loss_alpha, conf.loss_scale,
                      conf.grad_stop, conf.dt_stop, conf.dR_stop, conf.min_valid, conf.n_workgroups, buf, workspace,
                      bool(want_log), int(conf.spin_limit),
                      None if camera is None else [float(x) for x in camera[0]],
                      None if camera is None else [int(x) for x in camera[1]],
                      None if camera is None else camera[2], int(conf.path))
        keep = list(levels)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        return PendingLM(buf, want_log, n_levels, conf.num_iters, (p3d, mask, keep, workspace), done)

    @staticmethod
    def refine_levels_batch(problems: Sequence[dict], conf: _lib.LmConf, batch_workspace: torch.Tensor,
                            want_log: bool = True, pool_key=0) -> List["PendingLM"]:
        """K independent refinements in ONE persistent launch (pxt_lm_refine_batch): ``problems`` are
        PoseTrackerRefiner.lm_problem records (ref.p3d / ref.valid, packs, T_init, workspace, camera) of K objects
        tracked in lock-step; ``conf`` is shared (conf.n_workgroups = grid per problem, 0 = 256 / K).  Returns one
        result handle per problem; each problem's result is that of refine_levels with the same grid."""
        K = len(problems)
        assert 1 <= K <= _lib.PXT_LM_MAX_BATCH
        dev = problems[0]["ref"].p3d.device
        nh = 16 + _lib.PXT_MAX_LEVELS
        p3ds, masks, n_levels, fmaps, frefs, chans, cams, ndist, lambdas, T0, recs, wss = [], [], [], [], [], [], [], [], [], [], [], []
        # (a problem may carry the camera record of a render queued behind the launch - the tracker's steady frames - or
        # not: a frame after a failure)
        cam_conv, cam_slots, cam_outs, cam_on = [], [], [], []
        with_cam = any(pr.get("camera") is not None for pr in problems)
        bufs = _pinned_records([nh + (len(pr["packs"]) * conf.num_iters * _lib.PXT_LM_LOG_STRIDE if want_log else 0)
                                for pr in problems], pool_key)
        for pr, buf in zip(problems, bufs):
            ref, levels = pr["ref"], pr["packs"]
            _lib.require_gpu(ref.p3d, "p3d")
            p3ds.append(ref.p3d.to(torch.float32).contiguous())
            masks.append(None if ref.valid is None else ref.valid.to(dev, torch.uint8).contiguous())
            n_levels.append(len(levels))
            for lp in levels:
                assert lp.fref.shape == (ref.p3d.shape[0], lp.fmap.shape[2])
                fmaps.append(lp.fmap)
                frefs.append(lp.fref)
                chans.append(int(lp.C))
                cams += lp.camera.as10().tolist()
                ndist.append(int(lp.camera._data.shape[-1] - 6))
                lambdas += lp.lambda_.float().tolist()
            T = pr["T_init"]
            T0 += T.as12().detach().cpu().reshape(-1).tolist() if hasattr(T, "as12") else [float(x) for x in T]
            buf[:nh].zero_()
            recs.append(buf)
            wss.append(pr["workspace"])
            if with_cam:
                if pr.get("camera") is not None:
                    conv, slots, cam_out = pr["camera"]
                    cam_on.append(1)
                else:
                    conv, slots, cam_out = [0.0] * 27, [0, 0], None
                    cam_on.append(0)
                cam_conv += [float(x) for x in conv]
                cam_slots += ([int(x) for x in slots] + [0, 0])[:2]
                cam_outs.append(cam_out if cam_out is not None else _pinned_record(16))
        ops.lm_refine_batch(p3ds, masks, n_levels, fmaps, frefs, chans, cams, ndist, lambdas, T0, conf.num_iters, conf.pad,
                            conf.loss, conf.loss_alpha, conf.loss_scale, conf.grad_stop, conf.dt_stop, conf.dR_stop,
                            conf.min_valid, conf.n_workgroups, recs, wss, batch_workspace, bool(want_log),
                            int(conf.spin_limit), cam_conv if with_cam else None, cam_slots if with_cam else None,
                            cam_outs if with_cam else None, int(conf.path), cam_on if with_cam else None)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        return [PendingLM(buf, want_log, nl, conf.num_iters, (p3d, mk, list(pr["packs"]), ws), done)
                for buf, nl, p3d, mk, pr, ws in zip(recs, n_levels, p3ds, masks, problems, wss)]

    def run(self, p3D, F_ref, F_query, T_init: Pose, camera: Camera, mask=None, W_ref_query=None):
        """One pyramid level, pixloc calling convention:
        p3D [N,3] (numpy or tensor), F_ref [N,C], F_query [C,h,w], W_ref_query =
        (W_ref [N,1], W_q [1,h,w]).  Returns (T: Pose on F_query's device/dtype, failed)."""
        _lib.require_gpu(F_query, "F_query")
        dev = F_query.device
        if isinstance(p3D, np.ndarray):
            p3D = torch.from_numpy(p3D)
        p3D = p3D.to(dev, torch.float32)
        Cc, h, w = F_query.shape
        N = p3D.shape[0]
        cs = cstride_for(Cc)
        fmap = torch.zeros(h, w, cs, device=dev, dtype=torch.float32)
        fmap[..., :Cc] = F_query.permute(1, 2, 0)
        fref = torch.zeros(N, cs, device=dev, dtype=torch.float32)
        fref[:, :Cc] = F_ref
        if W_ref_query is not None:
            W_ref, W_q = W_ref_query
            fmap[..., Cc] = W_q[0]
            fref[:, Cc] = W_ref[:, 0]
        else:
            fmap[..., Cc] = 1.0
            fref[:, Cc] = 1.0
        lp = LevelPack(fmap, fref, Cc, camera, self.dampingnet())
        res = self.refine_levels(p3D, [lp], T_init, self.native_conf(), self._workspace(dev), mask).result()
        self.replay_log(res, 0, T_init)
        T = Pose(res.T.as12().to(dev, F_query.dtype))
        return T, torch.tensor(res.failed, device=dev)

    _run = run

    def replay_log(self, res: LMResult, level: int, T_init: Pose):
        """Hand the recorded iterations of one level to the attached tracker: in one call when it
        takes whole-level records (DebugTracker.record_level), otherwise iteration by iteration
        through ``logging_fn`` with the kwargs of pixloc's hook (tracker.py:32-46)."""
        n = res.iters[level]
        lg = res.log[level]
        if self.level_logging_fn is not None:
            self.level_logging_fn(T_init, lg[:n, 0].tolist(), lg[:n, 8:20], lg[:n, 3].tolist())
            return
        if self.logging_fn is None:
            return
        T_prev = Pose(T_init.as12().detach().cpu().float())
        for i in range(n):
            T = Pose(lg[i, 8:20].clone())
            self.logging_fn(i=i, T_init=T_init, T=T, T_delta=T @ T_prev.inv(),
                            cost=lg[i, 0:1].clone().reshape(1, 1),  # [batch 1, one already-averaged entry]
                            valid=torch.ones(1, 1), n_valid=int(lg[i, 1]))
            T_prev = T


_PINNED = threading.local()  # per thread: one tracker per thread is the supported concurrency


def _pinned_record(n_floats: int) -> torch.Tensor:
    """Reusable pinned host buffers, a ring of two per size: the record of one refinement stays
    readable while the next one is enqueued; it is recycled by the refinement after that."""
    pool = _PINNED.__dict__.setdefault("pool", {})
    ring = pool.get(n_floats)
    if ring is None:
        ring = pool[n_floats] = [[torch.zeros(n_floats, dtype=torch.float32).pin_memory() for _ in range(2)], 0]
    buf = ring[0][ring[1]]
    ring[1] ^= 1
    return buf


def _pinned_records(sizes: Sequence[int], pool_key=0) -> List[torch.Tensor]:
    """The K records of one batched launch: distinct pinned buffers, two sets alternating per (sizes, pool_key)
    signature (pool_key: the caller's group - launches of different groups are in flight together)."""
    pool = _PINNED.__dict__.setdefault("batch_pool", {})
    key = (pool_key,) + tuple(int(x) for x in sizes)
    ring = pool.get(key)
    if ring is None:
        ring = pool[key] = [[[torch.zeros(n, dtype=torch.float32).pin_memory() for n in key[1:]] for _ in range(2)], 0]
    bufs = ring[0][ring[1]]
    ring[1] ^= 1
    return bufs


class PendingLM:
    """Result handle of an enqueued refinement; .result() waits for the kernel's event and reads
    the record the kernel wrote into pinned host memory."""

    poll = True  # poll the completion word (single tracker per process); False: wait on the event

    def __init__(self, buf, has_log, n_levels, num_iters, keepalive, done):
        self.buf, self.has_log = buf, has_log
        self.n_levels, self.num_iters = n_levels, num_iters
        self._keep, self._done = keepalive, done

    def result(self) -> LMResult:
        # This sits on the frame's critical path (the next frame's render waits for the pose):
        # one blocking copy, then plain numpy/Python on the host record.
        # The kernel's last store (system-scope release) sets out[15]: poll it in the pinned
        # buffer - cheaper than an event wake-up on the frame's critical path.  The event is the
        # fallback for a kernel that never gets there (it then reports through `status`).
        flag = self.buf.numpy()
        spins = 0
        if not PendingLM.poll:  # several trackers on threads: a Python spin would hold the GIL
            self._done.synchronize()
        while flag[15] == 0.0:
            spins += 1
            if spins > 2_000_000 or (spins & 0x3FFF) == 0 and self._done.query():
                self._done.synchronize()
                break
        host = self.buf.clone()  # the pinned buffer is reused two refinements later
        h = host.numpy()
        nh = 16 + _lib.PXT_MAX_LEVELS
        status = int(h[13])
        if status != 0:
            # a timed-out launch may leave granules of epochs it never finished behind: the next launch on this
            # workspace starts from zeroed tags (the kernel also advances its tag base by a margin)
            if self._keep is not None:
                self._done.synchronize()
                self._keep[3].zero_()
            self._keep = None
            raise _lib.PxtError(f"pxt_lm_refine: in-kernel status {status} (spin bound exceeded)")
        iters = [int(h[16 + l]) for l in range(self.n_levels)]
        if self.has_log:
            log = host[nh:].view(self.n_levels, self.num_iters, _lib.PXT_LM_LOG_STRIDE)
            lg = h[nh:].reshape(self.n_levels, self.num_iters, _lib.PXT_LM_LOG_STRIDE)
            costs = [lg[l, : iters[l], 0].tolist() for l in range(self.n_levels)]
        else:
            log, costs = torch.zeros(self.n_levels, 0, _lib.PXT_LM_LOG_STRIDE), []
        self._keep = None
        return LMResult(Pose(host[:12].clone()), bool(h[12] != 0), iters, costs, log, int(h[14]))
