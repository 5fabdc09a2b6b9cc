"""Pose helpers on the tracking path (reference pixtrack/utils/pose_utils.py:8-27)."""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def geodesic_distance_for_rotations(R1: np.ndarray, R2: np.ndarray) -> float:
    """Angle (rad) of the relative rotation R1 R2^T (reference pose_utils.py:8-13 takes the norm
    of scipy's rotation vector).  Same value from atan2(|sin|, cos) of the relative rotation --
    accurate at 0 and pi alike and ~50x cheaper than building a scipy Rotation per call (the
    tracker evaluates it for every covisible reference each frame)."""
    Rd = np.asarray(R1, dtype=np.float64) @ np.asarray(R2, dtype=np.float64).T
    cos = (Rd[0, 0] + Rd[1, 1] + Rd[2, 2] - 1.0) * 0.5
    sx, sy, sz = Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]
    sin = 0.5 * np.sqrt(sx * sx + sy * sy + sz * sz)
    return float(np.arctan2(sin, cos))


def geodesic_distances_to(R1: np.ndarray, R2s: np.ndarray) -> np.ndarray:
    """geodesic_distance_for_rotations(R1, R2s[i]) for a stack R2s [n,3,3] in one pass (the tracker ranks
    every covisible reference each frame, on the frame's critical path): the same atan2(|sin|, cos), with
    the relative rotation's entries written out as sums of three products."""
    A = np.asarray(R1, dtype=np.float64)
    B = np.asarray(R2s, dtype=np.float64)
    Rd = np.matmul(A[None], B.transpose(0, 2, 1))  # A @ B[i].T
    cos = (Rd[:, 0, 0] + Rd[:, 1, 1] + Rd[:, 2, 2] - 1.0) * 0.5
    sx, sy, sz = Rd[:, 2, 1] - Rd[:, 1, 2], Rd[:, 0, 2] - Rd[:, 2, 0], Rd[:, 1, 0] - Rd[:, 0, 1]
    sin = 0.5 * np.sqrt(sx * sx + sy * sy + sz * sz)
    return np.arctan2(sin, cos)


def get_world_in_camera_from_pixpose(pixpose) -> np.ndarray:
    """4x4 world->camera matrix [R|t] of a Pose (reference pose_utils.py:16-21)."""
    R, t = pixpose.cpu().numpy()
    wIc = np.eye(4)
    wIc[:3, :3] = R
    wIc[:3, 3] = t
    return wIc


def get_camera_in_world_from_pixpose(pixpose) -> np.ndarray:
    """4x4 camera->world matrix (reference pose_utils.py:24-27: np.linalg.inv of the 4x4).  The
    matrix is rigid, so the inverse is [R^T | -R^T t]: same result to an ulp of float64, without
    the LU factorisation on the per-frame critical path."""
    wIc = get_world_in_camera_from_pixpose(pixpose)
    Rt = wIc[:3, :3].T
    cIw = np.eye(4, dtype=wIc.dtype)
    cIw[:3, :3] = Rt
    cIw[:3, 3] = -Rt @ wIc[:3, 3]
    return cIw


def get_pixpose_from_world_in_camera(wIc: np.ndarray):
    from ..geometry import Pose

    return Pose.from_Rt(wIc[:3, :3], wIc[:3, 3])


def get_pixpose_from_camera_in_world(cIw: np.ndarray):
    return get_pixpose_from_world_in_camera(np.linalg.inv(cIw))
