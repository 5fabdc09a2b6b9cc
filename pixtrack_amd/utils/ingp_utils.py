"""NeRF <-> SfM frame maps and Testbed construction (reference
pixtrack/utils/ingp_utils.py:16-83).  ``nerf2sfm.pkl`` = {up, centroid, avglen, totp, R}
written by colmap2ingp (reference pixtrack/utils/colmap2ingp.py:356-362)."""
from __future__ import annotations

import os
import pickle as pkl

import numpy as np

_FLIP_YZ = np.diag([1.0, -1.0, -1.0, 1.0])


def load_nerf2sfm(path):
    with open(path, "rb") as f:
        return pkl.load(f)


_WARNED_AABB = []


def initialize_ingp(snapshot_path, aabb, background=None, device=None):
    """Same render settings as reference ingp_utils.py:22-44, on the HIP Testbed."""
    from ..ngp import Testbed, TestbedMode

    if background is None:
        background = [255, 255, 255, 0.0]
    testbed = Testbed(TestbedMode.Nerf, device=device)
    testbed.nerf.sharpen = 0.0
    testbed.load_snapshot(snapshot_path)
    testbed.nerf.render_with_camera_distortion = True
    testbed.background_color = background
    testbed.snap_to_pixel_centers = True
    testbed.nerf.rendering_min_transmittance = 1e-7
    testbed.fov_axis = 0
    testbed.shall_train = False
    # config/motor_core.sh writes its y bounds max-first; a box with min > max on an axis contains no point, so
    # instant-ngp would render background only.  The corners are sorted per axis here (builder's decision).
    # PXT_STRICT_AABB=1 keeps the corners as given (instant-ngp's behaviour: an empty box); otherwise a swapped pair is
    # sorted WITH a warning, once per process - a deviation from the reference that must not be silent.
    swapped = [i for i, (a, b) in enumerate(zip(aabb[0], aabb[1])) if float(a) > float(b)]
    if swapped and os.environ.get("PXT_STRICT_AABB", "0") == "1":
        testbed.render_aabb.min = [float(a) for a in aabb[0]]
        testbed.render_aabb.max = [float(b) for b in aabb[1]]
        testbed.exposure = 0.0
        return testbed
    if swapped and not _WARNED_AABB:
        import warnings

        _WARNED_AABB.append(True)
        warnings.warn(f"render box {aabb} has min > max on axis {swapped}: instant-ngp would render background only "
                      "(the box contains no point); the corners are sorted per axis here (PXT_STRICT_AABB=1 keeps them)",
                      RuntimeWarning, stacklevel=2)
    testbed.render_aabb.min = [min(float(a), float(b)) for a, b in zip(aabb[0], aabb[1])]
    testbed.render_aabb.max = [max(float(a), float(b)) for a, b in zip(aabb[0], aabb[1])]
    testbed.exposure = 0.0
    return testbed


def sfm_to_nerf_pose(nerf2sfm, sfm_pose: np.ndarray) -> np.ndarray:
    """camera-in-world (SfM) 4x4 -> NeRF transform_matrix convention: camera y/z flip,
    world x<->y swap and z flip, recentre, scale to 'nerf size', align up, recentre on the
    cameras' centre of attention."""
    p = np.asarray(sfm_pose, dtype=np.float64) @ _FLIP_YZ
    p = p[[1, 0, 2, 3], :]
    p[2, :] *= -1
    p[0:3, 3] -= nerf2sfm["centroid"]
    p[0:3, 3] *= 3.0 / nerf2sfm["avglen"]
    p = nerf2sfm["R"] @ p
    p[0:3, 3] -= nerf2sfm["totp"]
    return p


def nerf_to_sfm_pose(nerf2sfm, nerf_pose: np.ndarray) -> np.ndarray:
    p = np.array(nerf_pose, dtype=np.float64)
    p[0:3, 3] += nerf2sfm["totp"]
    p = np.linalg.inv(nerf2sfm["R"]) @ p
    p[0:3, 3] /= 3.0 / nerf2sfm["avglen"]
    p[0:3, 3] += nerf2sfm["centroid"]
    p[2, :] *= -1
    p = p[[1, 0, 2, 3], :]
    return p @ _FLIP_YZ


def get_nerf_aabb_from_sfm(model3d, nerf2sfm):
    """Render box for the YCB policy (reference ingp_utils.py:86-109, used at
    pixloc_tracker_ycb.py:92): SfM points mapped into NeRF coordinates, DBSCAN(eps 0.1, min 2)
    fitted as the reference does, box = min/max over ALL points (the reference computes the
    cluster extremes but returns the global ones), converted to ngp's unit cube (/3 + 0.5) with
    the (y, z, x) axis cycle."""
    from sklearn.cluster import DBSCAN

    if not isinstance(nerf2sfm, dict):
        nerf2sfm = load_nerf2sfm(nerf2sfm)
    pts = []
    for p in model3d.points3D.values():
        T = np.eye(4)
        T[:3, -1] = p.xyz
        pts.append(sfm_to_nerf_pose(nerf2sfm, T)[:3, -1])
    pts = np.array(pts)
    DBSCAN(eps=0.1, min_samples=2).fit(pts)
    lo = pts.min(axis=0) / 3.0 + 0.5
    hi = pts.max(axis=0) / 3.0 + 0.5
    return [[lo[1], lo[2], lo[0]], [hi[1], hi[2], hi[0]]]


def get_object_center_from_sfm(model3d):
    """Mean of the SfM points (reference ingp_utils.py:112-116)."""
    return np.mean(np.array([p.xyz for p in model3d.points3D.values()]), axis=0)
