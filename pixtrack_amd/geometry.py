"""Rigid poses and pinhole cameras with the surface pixtrack consumes.

pixtrack never defines these types itself; it imports them from pixloc
(`from pixloc.pixlib.geometry import Camera, Pose`, reference
pixtrack/pose_trackers/pixloc_tracker_r9.py:8, pixtrack/localization/
pixloc_pose_refiners.py:10) and touches: ``Pose.from_Rt / from_colmap / numpy() /
magnitude() / inv() / @ / * / cpu() / to()`` and ``Camera.from_colmap / scale /
size / f / c / world2image``.  These classes keep those
names, argument meanings and conventions:

* ``Pose`` stores ``[..., 12]`` = row-major R (9) followed by t (3); ``T * p``
  maps world points into the camera frame; ``A @ B`` composes (apply B first).
* ``Camera`` stores ``[..., 6 + k]`` = (w, h, fx, fy, cx, cy, dist...), with the
  principal point shifted by -0.5 from COLMAP's pixel-corner origin to a
  pixel-centre origin.

They are thin host-side value types (torch tensors, CPU or device); the
per-iteration arithmetic of the hot path lives in the HIP kernels, which take the
same 12-/10-float encodings (include/pixtrack_hip.h).
"""
from __future__ import annotations

import math
from typing import Tuple, Union

import numpy as np
import torch


def _as_tensor(x, like: torch.Tensor = None, dtype=None) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    elif not torch.is_tensor(x):
        x = torch.as_tensor(x)
    if like is not None:
        x = x.to(device=like.device, dtype=like.dtype)
    elif dtype is not None:
        x = x.to(dtype)
    return x


def skew_symmetric(v: torch.Tensor) -> torch.Tensor:
    """[v]x for ``v`` of shape [..., 3]."""
    z = torch.zeros_like(v[..., 0])
    M = torch.stack(
        [z, -v[..., 2], v[..., 1], v[..., 2], z, -v[..., 0], -v[..., 1], v[..., 0], z],
        dim=-1,
    ).reshape(v.shape[:-1] + (3, 3))
    return M


def so3exp_map(w: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Rodrigues formula with a first-order branch below ``eps`` (Appendix A.2)."""
    theta = w.norm(p=2, dim=-1, keepdim=True)
    small = theta < eps
    div = torch.where(small, torch.ones_like(theta), theta)
    W = skew_symmetric(w / div)
    theta = theta[..., None]
    res = W * torch.sin(theta) + (W @ W) * (1 - torch.cos(theta))
    res = torch.where(small[..., None], W, res)
    return torch.eye(3).to(W) + res


class TensorWrapper:
    """Minimal tensor-backed value type (shape/device/dtype plumbing only)."""

    _data: torch.Tensor

    def __init__(self, data: torch.Tensor):
        self._data = data

    @property
    def shape(self):
        return self._data.shape[:-1]

    @property
    def device(self):
        return self._data.device

    @property
    def dtype(self):
        return self._data.dtype

    def __getitem__(self, index):
        return self.__class__(self._data[index])

    def to(self, *args, **kwargs):
        # pixloc passes a tensor ("T.to(F_q)") to copy device and dtype.
        return self.__class__(self._data.to(*args, **kwargs))

    def cpu(self):
        return self.__class__(self._data.cpu())

    def cuda(self):
        return self.__class__(self._data.cuda())

    def float(self):
        return self.__class__(self._data.float())

    def double(self):
        return self.__class__(self._data.double())

    def detach(self):
        return self.__class__(self._data.detach())

    def __repr__(self):
        return f"{self.__class__.__name__} {tuple(self.shape)} {self.dtype} {self.device}"


class Pose(TensorWrapper):
    def __init__(self, data: torch.Tensor):
        assert data.shape[-1] == 12
        super().__init__(data)

    # ---- constructors -------------------------------------------------
    @classmethod
    def from_Rt(cls, R, t) -> "Pose":
        R = _as_tensor(R)
        t = _as_tensor(t).to(R)
        assert R.shape[-2:] == (3, 3) and t.shape[-1] == 3
        return cls(torch.cat([R.flatten(start_dim=-2), t], -1))

    @classmethod
    def from_aa(cls, aa, t) -> "Pose":
        aa = _as_tensor(aa)
        return cls.from_Rt(so3exp_map(aa), _as_tensor(t).to(aa))

    @classmethod
    def from_4x4mat(cls, T) -> "Pose":
        T = _as_tensor(T)
        return cls.from_Rt(T[..., :3, :3], T[..., :3, 3])

    @classmethod
    def from_colmap(cls, image) -> "Pose":
        return cls.from_Rt(image.qvec2rotmat(), image.tvec)

    # ---- accessors ----------------------------------------------------
    @property
    def R(self) -> torch.Tensor:
        rvec = self._data[..., :9]
        return rvec.reshape(rvec.shape[:-1] + (3, 3))

    @property
    def t(self) -> torch.Tensor:
        return self._data[..., -3:]

    def inv(self) -> "Pose":
        R = self.R.transpose(-1, -2)
        t = -(R @ self.t.unsqueeze(-1)).squeeze(-1)
        return self.__class__.from_Rt(R, t)

    def compose(self, other: "Pose") -> "Pose":
        R = self.R @ other.R
        t = self.t + (self.R @ other.t.unsqueeze(-1)).squeeze(-1)
        return self.__class__.from_Rt(R, t)

    def transform(self, p3d) -> torch.Tensor:
        p3d = _as_tensor(p3d, like=self._data)
        assert p3d.shape[-1] == 3
        return p3d @ self.R.transpose(-1, -2) + self.t.unsqueeze(-2)

    def __mul__(self, p3d) -> torch.Tensor:
        return self.transform(p3d)

    def __matmul__(self, other: "Pose") -> "Pose":
        return self.compose(other)

    def J_transform(self, p3d_out: torch.Tensor) -> torch.Tensor:
        """d(T*p)/d(delta) for the left update T <- exp(delta) T; translation first."""
        J_t = torch.diag_embed(torch.ones_like(p3d_out))
        J_rot = -skew_symmetric(p3d_out)
        return torch.cat([J_t, J_rot], dim=-1)

    def numpy(self) -> Tuple[np.ndarray, np.ndarray]:
        a = self._data.numpy()  # one conversion (this sits on the per-frame critical path)
        return a[..., :9].reshape(a.shape[:-1] + (3, 3)), a[..., 9:]

    def magnitude(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(rotation angle in DEGREES, translation norm)."""
        trace = torch.diagonal(self.R, dim1=-1, dim2=-2).sum(-1)
        cos = torch.clamp((trace - 1) / 2, -1, 1)
        dr = torch.acos(cos).abs() / math.pi * 180
        dt = torch.norm(self.t, dim=-1)
        return dr, dt

    def as12(self) -> torch.Tensor:
        """The raw [..., 12] encoding handed to the C ABI."""
        return self._data


class Camera(TensorWrapper):
    eps = 1e-3

    def __init__(self, data: torch.Tensor):
        assert data.shape[-1] in {6, 8, 10}
        super().__init__(data)

    @classmethod
    def from_colmap(cls, camera) -> "Camera":
        """COLMAP camera (namedtuple / dict / object with model,width,height,params)."""
        if isinstance(camera, tuple) and hasattr(camera, "_asdict"):
            camera = camera._asdict()
        if isinstance(camera, dict):
            model = camera["model"]
            params = np.asarray(camera["params"], dtype=np.float64)
            width, height = camera["width"], camera["height"]
        else:
            model = getattr(camera, "model", None) or getattr(camera, "model_name")
            params = np.asarray(camera.params, dtype=np.float64)
            width, height = camera.width, camera.height
        if model in ["OPENCV", "PINHOLE"]:
            (fx, fy, cx, cy), params = np.split(params, [4])
        elif model in ["SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL"]:
            (f, cx, cy), params = np.split(params, [3])
            fx = fy = f
            if model == "SIMPLE_RADIAL":
                params = np.r_[params, 0.0]
        else:
            raise NotImplementedError(model)
        data = np.r_[width, height, fx, fy, cx - 0.5, cy - 0.5, params]
        return cls(torch.from_numpy(data).float())

    @property
    def size(self) -> torch.Tensor:
        return self._data[..., :2]

    @property
    def f(self) -> torch.Tensor:
        return self._data[..., 2:4]

    @property
    def c(self) -> torch.Tensor:
        return self._data[..., 4:6]

    @property
    def dist(self) -> torch.Tensor:
        return self._data[..., 6:]

    def scale(self, scales: Union[float, int, Tuple[float, float]]) -> "Camera":
        if isinstance(scales, (int, float)):
            scales = (scales, scales)
        # memo per instance: the tracking loop rescales the same (immutable) cameras with the same pyramid
        # factors seven times per frame, ~20 us of small tensor ops each on the host that feeds the GPU
        key = (float(scales[0]), float(scales[1]))
        memo = self.__dict__.setdefault("_scaled", {})
        hit = memo.get(key)
        if hit is None:
            s = self._data.new_tensor(scales)
            data = torch.cat([self.size * s, self.f * s, (self.c + 0.5) * s - 0.5, self.dist], -1)
            hit = memo[key] = self.__class__(data)
        return hit

    def __getstate__(self):
        # the memo of scaled copies is a cache of this process, not part of the camera (poses.pkl stays as small as
        # the reference's)
        return {k: v for k, v in self.__dict__.items() if k != "_scaled"}

    def in_image(self, p2d: torch.Tensor) -> torch.Tensor:
        size = self.size.unsqueeze(-2)
        return torch.all((p2d >= 0) & (p2d <= (size - 1)), -1)

    def project(self, p3d: torch.Tensor):
        z = p3d[..., -1]
        valid = z > self.eps
        z = z.clamp(min=self.eps)
        return p3d[..., :-1] / z.unsqueeze(-1), valid

    def J_project(self, p3d: torch.Tensor) -> torch.Tensor:
        x, y, z = p3d[..., 0], p3d[..., 1], p3d[..., 2]
        zero = torch.zeros_like(z)
        z = z.clamp(min=self.eps)
        J = torch.stack([1 / z, zero, -x / z**2, zero, 1 / z, -y / z**2], dim=-1)
        return J.reshape(p3d.shape[:-1] + (2, 3))

    def undistort(self, pts: torch.Tensor):
        assert self.dist.shape[-1] in (0, 2, 4)
        return undistort_points(pts, self.dist)

    def J_undistort(self, pts: torch.Tensor) -> torch.Tensor:
        return J_undistort_points(pts, self.dist)

    def denormalize(self, p2d: torch.Tensor) -> torch.Tensor:
        return p2d * self.f.unsqueeze(-2) + self.c.unsqueeze(-2)

    def J_denormalize(self) -> torch.Tensor:
        return torch.diag_embed(self.f).unsqueeze(-3)

    def world2image(self, p3d):
        p3d = _as_tensor(p3d, like=self._data)
        p2d, visible = self.project(p3d)
        p2d, mask = self.undistort(p2d)
        p2d = self.denormalize(p2d)
        valid = visible & mask & self.in_image(p2d)
        return p2d, valid

    def J_world2image(self, p3d: torch.Tensor):
        p2d_dist, valid = self.project(p3d)
        J = self.J_denormalize() @ self.J_undistort(p2d_dist) @ self.J_project(p3d)
        return J, valid

    def as10(self) -> torch.Tensor:
        """(w,h,fx,fy,cx,cy,k1,k2,p1,p2) zero-padded: the C ABI camera record."""
        d = self._data
        assert d.dim() == 1
        out = torch.zeros(10, dtype=torch.float32)
        out[: d.shape[-1]] = d.detach().cpu().float()
        return out


def undistort_points(pts: torch.Tensor, dist: torch.Tensor):
    """Apply (k1,k2[,p1,p2]) to normalised coordinates; validity = inside the
    monotone range of the radial polynomial (Appendix A.1)."""
    dist = dist.unsqueeze(-2)
    ndist = dist.shape[-1]
    undist = pts
    valid = torch.ones(pts.shape[:-1], device=pts.device, dtype=torch.bool)
    if ndist > 0:
        k1, k2 = dist[..., :2].split(1, -1)
        r2 = torch.sum(pts**2, -1, keepdim=True)
        radial = k1 * r2 + k2 * r2**2
        undist = undist + pts * radial
        limited = ((k2 > 0) & ((9 * k1**2 - 20 * k2) > 0)) | ((k2 <= 0) & (k1 > 0))
        limit = torch.abs(
            torch.where(
                k2 > 0,
                (torch.sqrt((9 * k1**2 - 20 * k2).clamp(min=0)) - 3 * k1) / (10 * k2),
                1 / (3 * k1),
            )
        )
        valid = valid & torch.squeeze(~limited | (r2 < limit), -1)
        if ndist > 2:
            p12 = dist[..., 2:]
            p21 = p12.flip(-1)
            uv = torch.prod(pts, -1, keepdim=True)
            undist = undist + 2 * p12 * uv + p21 * (r2 + 2 * pts**2)
    return undist, valid


def J_undistort_points(pts: torch.Tensor, dist: torch.Tensor) -> torch.Tensor:
    dist = dist.unsqueeze(-2)
    ndist = dist.shape[-1]
    J_diag = torch.ones_like(pts)
    J_cross = torch.zeros_like(pts)
    if ndist > 0:
        k1, k2 = dist[..., :2].split(1, -1)
        r2 = torch.sum(pts**2, -1, keepdim=True)
        uv = torch.prod(pts, -1, keepdim=True)
        radial = k1 * r2 + k2 * r2**2
        d_radial = 2 * k1 + 4 * k2 * r2
        J_diag = J_diag + radial + (pts**2) * d_radial
        J_cross = J_cross + uv * d_radial
        if ndist > 2:
            p12 = dist[..., 2:]
            p21 = p12.flip(-1)
            J_diag = J_diag + 2 * p12 * pts.flip(-1) + 6 * p21 * pts
            J_cross = J_cross + 2 * p12 * pts + 2 * p21 * pts.flip(-1)
    return torch.diag_embed(J_diag) + torch.diag_embed(J_cross).flip(-1)
