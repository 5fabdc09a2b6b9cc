"""``get_nerf_image`` -- the one function of the reference's visualisation module that sits on
the tracking path (pixtrack/visualization/run_vis_on_poses.py:28-57) -- and, below it, the offline
overlay renderer of the same file (:60-371; SURVEY 8f rank 1, host-side drawing)."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib
from ..ops import ops


def get_nerf_image_device(testbed, nerf_pose, camera, depth: bool = False, alpha_thresh: float = 0.0,
                          spp: int = 8, **render_kw):
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
        rgba = testbed.render_device(width, height, spp, True, **render_kw)
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


# -------------------------------------------------------------------------------------------------
# Offline overlay renderer (reference pixtrack/visualization/run_vis_on_poses.py
# :60-371): for every frame of a poses.pkl, the NeRF render at the tracked pose blended over the
# query frame, the object's axes and centre drawn at the tracked pose, optional reference-image
# inset and pose-error text; written to <out_dir>/results/result_<frame>.  PIL instead of cv2 (not
# in this image); colours are given in RGB for the same pixels the reference's BGR tuples produce.
# -------------------------------------------------------------------------------------------------
AXIS_RGB = ((0, 0, 255), (0, 255, 0), (255, 0, 0))  # x, y, z: cv2 BGR (255,0,0) / (0,255,0) / (0,0,255)


def pinhole_K(camera) -> np.ndarray:
    """Intrinsics the overlays use: fx on both axes, principal point at the image centre (:84-90)."""
    width, height = float(camera.size[0]), float(camera.size[1])
    f = float(camera.f[0])
    return np.array([[f, 0.0, width / 2], [0.0, f, height / 2], [0.0, 0.0, 1.0]])


def project_3d_to_2d(pts_cam: np.ndarray, K: np.ndarray) -> np.ndarray:
    p = (K @ np.asarray(pts_cam, np.float64).T)
    return (p[:2] / p[2]).T


def world_to_camera(points_h: np.ndarray, camera_in_world: np.ndarray) -> np.ndarray:
    """Homogeneous world points [n,4] -> camera coordinates [n,3] for a 4x4 camera-in-world pose."""
    return points_h @ np.linalg.inv(camera_in_world).T[:, :3]


def blend_images(query_image: np.ndarray, nerf_image: np.ndarray, alpha: float = 0.3) -> np.ndarray:
    return (np.asarray(query_image, np.float64) * alpha + np.asarray(nerf_image, np.float64) * (1 - alpha)).astype(np.uint8)


def add_pose_axes(image: np.ndarray, camera, camera_in_world: np.ndarray, axes_center, length: float = 0.025,
                  thickness: int = 2) -> np.ndarray:
    """Three segments from the object centre along +x, -y, -z (the reference's axis triple, :93-104)."""
    from PIL import Image, ImageDraw

    c = np.asarray(list(axes_center)[:3], np.float64)
    ends = [c + [length, 0, 0], c + [0, -length, 0], c + [0, 0, -length]]
    pts = np.array([np.append(p, 1.0) for p in [c] + ends])
    uv = project_3d_to_2d(world_to_camera(pts, camera_in_world), pinhole_K(camera)).astype(np.int16)
    im = Image.fromarray(np.ascontiguousarray(image))
    d = ImageDraw.Draw(im)
    for k in range(3):
        d.line([tuple(map(int, uv[0])), tuple(map(int, uv[1 + k]))], fill=AXIS_RGB[k], width=int(thickness))
    return np.asarray(im)


def add_object_center(image: np.ndarray, camera, camera_in_world: np.ndarray, object_center, thickness: int = 5):
    from PIL import Image, ImageDraw

    p = np.append(np.asarray(object_center, np.float64)[:3], 1.0)[None]
    u, v = project_3d_to_2d(world_to_camera(p, camera_in_world), pinhole_K(camera)).astype(np.int16)[0]
    im = Image.fromarray(np.ascontiguousarray(image))
    r = thickness / 2.0
    ImageDraw.Draw(im).ellipse([u - r, v - r, u + r, v + r], fill=(255, 255, 255))
    return np.asarray(im)


def add_text_lines(image: np.ndarray, lines, origin=(20, 12), color=(0, 0, 255), step: int = 30) -> np.ndarray:
    from PIL import Image, ImageDraw

    im = Image.fromarray(np.ascontiguousarray(image))
    d = ImageDraw.Draw(im)
    for i, text in enumerate(lines):
        d.text((origin[0], origin[1] + i * step), text, fill=color)
    return np.asarray(im)


def add_reference_image(base_image: np.ndarray, reference_image: np.ndarray, name: str, s: float = 0.25) -> np.ndarray:
    """Inset of the current reference image in the top-left corner with its name (:222-243)."""
    from PIL import Image

    scale = base_image.shape[1] * s / reference_image.shape[1]
    dim = (int(reference_image.shape[1] * scale), int(reference_image.shape[0] * scale))
    small = np.asarray(Image.fromarray(np.ascontiguousarray(reference_image)).resize(dim, Image.BOX))
    out = base_image.copy()
    out[: dim[1], : dim[0]] = small
    label = name.split("/")[-1].split(".")[0]
    return add_text_lines(out, ["Reference image: %s" % label], origin=(5, max(dim[1] - 22, 0)))


def render_overlays(pose_stream: dict, testbed, nerf2sfm, object_center, out_dir, no_axes: bool = False,
                    obj_center: bool = False, pose_error: bool = False, read_query=None, reference_lookup=None) -> list:
    """The main loop of the reference script (:297-371).  ``pose_stream`` is a poses.pkl dict;
    ``read_query(path) -> uint8 HxWx3`` defaults to utils.io.read_image; ``reference_lookup(ref_ids)
    -> (image, name)`` enables the reference inset.  Returns the written paths."""
    import os

    from PIL import Image

    from ..utils.ingp_utils import sfm_to_nerf_pose
    from ..utils.io import read_image
    from ..utils.pose_utils import geodesic_distance_for_rotations, get_camera_in_world_from_pixpose

    read_query = read_query or read_image
    results_dir = os.path.join(str(out_dir), "results")
    os.makedirs(results_dir, exist_ok=True)
    written = []
    for name_q, rec in pose_stream.items():
        query_img = np.asarray(read_query(rec["query_path"]))[..., :3].astype(np.uint8)
        camera = rec["camera"]
        if "T_refined" in rec:
            cIw = get_camera_in_world_from_pixpose(rec["T_refined"])
            nerf_img = get_nerf_image(testbed, sfm_to_nerf_pose(nerf2sfm, cIw), camera, alpha_thresh=0.0)
        else:  # the reference keeps the previous frame's pose for the axes here; without one, skip them
            cIw = None
            nerf_img = np.full(query_img.shape, 255, np.uint8)
        result = blend_images(query_img, nerf_img)
        if reference_lookup is not None:
            ref_img, ref_name = reference_lookup(rec["reference_ids"])
            result = add_reference_image(result, ref_img, ref_name)
        if cIw is not None and not no_axes:
            result = add_pose_axes(result, camera, cIw, list(object_center) + [0])
        if cIw is not None and not obj_center:  # (sic: the reference draws the centre unless --obj_center is given)
            result = add_object_center(result, camera, cIw, object_center)
        if pose_error and "T_refined" in rec and "gt_pose" in rec:
            pr_R, pr_T = rec["T_refined"].numpy()
            gt_R, gt_T = rec["gt_pose"].numpy()
            rot = geodesic_distance_for_rotations(gt_R, pr_R) * 180.0 / np.pi
            tra = float(np.linalg.norm(gt_T - pr_T)) * 100.0
            result = add_text_lines(result, [f"Rotation error: {rot:.4f} degrees", f"Translation error: {tra:.4f} cm"])
        path = os.path.join(results_dir, "result_%s" % os.path.basename(str(rec["query_path"])))
        Image.fromarray(result).save(path)
        written.append(path)
    return written


def main(argv=None):
    """python -m pixtrack_amd.visualization.run_vis_on_poses --object_path P --out_dir D [--reference_image 1]
    [--no_axes] [--obj_center] [--pose_error] [--obj_aabb "[[..],[..]]"]  (reference :256-304)."""
    import argparse
    import ast
    from pathlib import Path

    from ..model3d import Model3D
    from ..utils.ingp_utils import get_nerf_aabb_from_sfm, get_object_center_from_sfm, initialize_ingp, load_nerf2sfm
    from ..utils.io import load_reference_pickle, read_image

    ap = argparse.ArgumentParser()
    ap.add_argument("--object_path", type=Path)
    ap.add_argument("--out_dir", type=Path)
    ap.add_argument("--reference_image", default=False)
    ap.add_argument("--no_axes", action="store_true", default=False)
    ap.add_argument("--obj_center", action="store_true", default=False)
    ap.add_argument("--pose_error", action="store_true", default=False)
    ap.add_argument("--obj_aabb", type=str, default="")
    args = ap.parse_args(argv)
    obj = args.object_path
    sfm_dir = obj / "pixtrack/aug_nerf_sfm/aug_sfm"
    model3d = Model3D(str(sfm_dir))
    nerf2sfm = load_nerf2sfm(str(obj / "pixtrack/pixsfm/dataset/nerf2sfm.pkl"))
    aabb = ast.literal_eval(args.obj_aabb) if args.obj_aabb else get_nerf_aabb_from_sfm(model3d, nerf2sfm)
    testbed = initialize_ingp(str(obj / "pixtrack/instant-ngp/snapshots/weights.msgpack"), aabb)
    poses = load_reference_pickle(Path(args.out_dir) / "poses.pkl")
    lookup = None
    if args.reference_image:
        def lookup(ref_ids):
            name = model3d.dbs[ref_ids[0]].name
            return read_image(obj / "pixtrack/aug_nerf_sfm" / name), name
    written = render_overlays(poses, testbed, nerf2sfm, get_object_center_from_sfm(model3d), args.out_dir,
                              no_axes=args.no_axes, obj_center=args.obj_center, pose_error=args.pose_error,
                              reference_lookup=lookup)
    print("wrote %d overlays to %s" % (len(written), Path(args.out_dir) / "results"))


if __name__ == "__main__":
    main()
