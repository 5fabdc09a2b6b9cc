"""Pose helpers on the tracking path (reference pixtrack/utils/pose_utils.py:8-27)."""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def geodesic_distance_for_rotations(R1: np.ndarray, R2: np.ndarray) -> float:
    """Angle (rad) of the relative rotation R1 R2^T (reference pose_utils.py:8-13)."""
    return float(np.linalg.norm(Rotation.from_matrix(np.asarray(R1) @ np.asarray(R2).T).as_rotvec()))


def get_world_in_camera_from_pixpose(pixpose) -> np.ndarray:
    """4x4 world->camera matrix [R|t] of a Pose (reference pose_utils.py:16-21)."""
    R, t = pixpose.cpu().numpy()
    wIc = np.eye(4)
    wIc[:3, :3] = R
    wIc[:3, 3] = t
    return wIc


def get_camera_in_world_from_pixpose(pixpose) -> np.ndarray:
    """4x4 camera->world matrix (reference pose_utils.py:24-27)."""
    return np.linalg.inv(get_world_in_camera_from_pixpose(pixpose))


def get_pixpose_from_world_in_camera(wIc: np.ndarray):
    from ..geometry import Pose

    return Pose.from_Rt(wIc[:3, :3], wIc[:3, 3])


def get_pixpose_from_camera_in_world(cIw: np.ndarray):
    return get_pixpose_from_world_in_camera(np.linalg.inv(cIw))
