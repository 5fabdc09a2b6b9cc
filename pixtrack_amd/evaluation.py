"""Trajectory metrics of the reference's evaluation notebook (reference
notebooks/GetMetrics.ipynb: ``similarity_transform``, ``get_pose_offset``, ``get_metrics``).

The notebook aligns the estimated camera-translation track to the ground-truth one with a
similarity (Umeyama) fit, then reports per-frame mean vertex distance (the ADD figure it
calls ``average_error_vertices``), translation error in cm, and the share of frames inside a
(cm, degree) threshold.  Host-side numpy on a few hundred 4x4 matrices: not hot-path work.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .utils.pose_utils import geodesic_distance_for_rotations


def get_pose_mat_from_tensor(pose) -> np.ndarray:
    """Pose -> 4x4 (notebook cell ``get_pose_mat_from_tensor``)."""
    T = np.eye(4)
    T[:3, :3] = pose.R.cpu().numpy()
    T[:3, 3] = pose.t.cpu().numpy()
    return T


def similarity_transform(from_points: np.ndarray, to_points: np.ndarray):
    """Least-squares (R, c, t) with ``to ~ c R from + t`` (Umeyama 1991, as in the notebook,
    including its reflection rule and the collinearity error)."""
    assert from_points.ndim == 2, "from_points must be a m x n array"
    assert from_points.shape == to_points.shape, "from_points and to_points must have the same shape"
    N, m = from_points.shape
    mean_from, mean_to = from_points.mean(axis=0), to_points.mean(axis=0)
    d_from, d_to = from_points - mean_from, to_points - mean_to
    sigma_from = (d_from * d_from).sum(axis=1).mean()
    cov = d_to.T.dot(d_from) / N
    U, d, Vt = np.linalg.svd(cov, full_matrices=True)
    rank = np.linalg.matrix_rank(cov)
    S = np.eye(m)
    if rank >= m - 1 and np.linalg.det(cov) < 0:
        S[m - 1, m - 1] = -1
    elif rank < m - 1:
        raise ValueError("colinearility detected in covariance matrix:\n{}".format(cov))
    R = U.dot(S).dot(Vt)
    c = (d * S.diagonal()).sum() / sigma_from
    t = mean_to - c * R.dot(mean_from)
    return R, c, t


def get_pose_offset(poses_file: Dict) -> np.ndarray:
    """4x4 rigid part (scale dropped, as the notebook does) of the similarity that maps the
    ground-truth translations onto the refined ones, over successful frames."""
    from_trs, to_trs = [], []
    for key in poses_file:
        if not poses_file[key]["success"]:
            continue
        to_trs.append(poses_file[key]["T_refined"].t.cpu().numpy())
        from_trs.append(poses_file[key]["gt_pose"].t.cpu().numpy())
    R, _, t = similarity_transform(np.array(from_trs, dtype=np.float64), np.array(to_trs, dtype=np.float64))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def get_metrics(poses_file: Dict, vertices: np.ndarray, tr_threshold: float, rot_threshold: float) -> Dict:
    """``vertices``: [V, 4] homogeneous model points in the object frame (the notebook's global
    of that name).  Distances in cm (x100), rotations in degrees; a frame is bad when either
    exceeds its threshold.  Frames with ``success == False`` are skipped, so - as in the
    notebook - they stay in ``total_frames`` but are never counted bad."""
    offset = get_pose_offset(poses_file)
    distances, pose_dists, bad = [], [], 0
    for key in poses_file:
        if not poses_file[key]["success"]:
            continue
        res = get_pose_mat_from_tensor(poses_file[key]["T_refined"])
        gt = get_pose_mat_from_tensor(poses_file[key]["gt_pose"])
        aligned = offset @ res
        tr = np.linalg.norm(gt[:3, 3] - aligned[:3, 3]) * 100
        rot = geodesic_distance_for_rotations(gt[:3, :3], aligned[:3, :3]) * 180 / np.pi
        res_v = (offset @ (res @ vertices.T)).T[:, :3] * 100
        gt_v = (gt @ vertices.T).T[:, :3] * 100
        distances.append(np.mean(np.linalg.norm(gt_v - res_v, axis=1)))
        pose_dists.append(tr)
        if tr > tr_threshold or rot > rot_threshold:
            bad += 1
    n = len(poses_file)
    return {
        "average_error_vertices": float(np.mean(distances)),
        "max_error": float(np.max(distances)),
        "max_translation_error": float(np.max(pose_dists)),
        "average_translation_error_pose": float(np.mean(pose_dists)),
        "bad_count": bad,
        "total_frames": n,
        "accuracy": (1.0 * (n - bad)) / (1.0 * n),
    }


def adds_distance(T_est: np.ndarray, T_gt: np.ndarray, vertices: np.ndarray) -> float:
    """ADD-S (symmetric average closest-point distance, Xiang et al. 2018) between the model
    under two 4x4 poses; ``vertices`` [V, 3|4].  BASELINE config 3 reports it against the
    synthetic ground truth.  O(V^2) in blocks: V is a few thousand."""
    v = np.asarray(vertices, dtype=np.float64)[:, :3]
    a = v @ T_est[:3, :3].T + T_est[:3, 3]
    b = v @ T_gt[:3, :3].T + T_gt[:3, 3]
    best = np.empty(len(b))
    for s in range(0, len(b), 1024):
        d = np.linalg.norm(b[s:s + 1024, None, :] - a[None, :, :], axis=-1)
        best[s:s + 1024] = d.min(axis=1)
    return float(best.mean())
