"""``get_nerf_image`` -- the one function of the reference's visualisation module that sits on
the tracking path (pixtrack/visualization/run_vis_on_poses.py:28-57; everything else in that
file is offline overlay rendering and out of scope)."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib
from ..ops import ops


def get_nerf_image_device(testbed, nerf_pose, camera, depth: bool = False, alpha_thresh: float = 0.0,
                          spp: int = 8):
    """Renders at ``camera``'s size with fov from fx (cx, cy, fy ignored: reference quirk
    Appendix D.5).  Returns the float32 RGBA frame [H,W,4] left on the device; ``alpha_thresh``
    is applied by ``rgba_to_u8`` (kept in the signature for parity with get_nerf_image)."""
    width, height = camera.size
    width, height = int(width), int(height)
    fl_x = float(camera.f[0])
    angle_x = math.atan(width / (fl_x * 2)) * 2
    testbed.fov = angle_x * 180 / np.pi
    testbed.set_nerf_camera_matrix(np.asarray(nerf_pose)[:3, :])
    if depth:
        testbed.render_mode = testbed.render_mode.Depth
    try:
        rgba = testbed.render_device(width, height, spp, True)
    finally:
        if depth:
            testbed.render_mode = testbed.render_mode.Shade
    return rgba


def rgba_to_u8(rgba: torch.Tensor, alpha_thresh: float = 0.0) -> torch.Tensor:
    """``nerf_img[alpha < thresh] = 0; (nerf_img[:, :, :3] * 255).astype(uint8)`` on the device."""
    H, W = int(rgba.shape[0]), int(rgba.shape[1])
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=rgba.device)
    ops.rgba_to_u8(rgba, float(alpha_thresh), out)
    return out


def get_nerf_image(testbed, nerf_pose, camera, depth: bool = False, alpha_thresh: float = 0.0) -> np.ndarray:
    """Reference contract: uint8 H x W x 3 on the host."""
    rgba = get_nerf_image_device(testbed, nerf_pose, camera, depth, alpha_thresh)
    return rgba_to_u8(rgba, alpha_thresh).cpu().numpy()
