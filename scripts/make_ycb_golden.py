"""Generates tests/golden/ycb_640x480.npz: ONE frame of BASELINE configs[2] under the YCB policy
(reference pixtrack/pose_trackers/pixloc_tracker_ycb.py:241-295) computed by the CPU oracle at full
size: unit-cube object box (config/cracker_box.sh:3), 640x480 OPENCV camera with the principal point
at (319.5, 239.5) (pixtrack/utils/io.py:50), reference_scale 0.3, mask every frame, image scale [1],
render box = box of the SfM points (get_nerf_aabb_from_sfm).  Mask and reference cameras differ, so
this is the TWO-render path.  CPU only (~3 min).

    python scripts/make_ycb_golden.py
    python scripts/make_ycb_golden.py --refshape     # tests/golden/ycb_refshape_921.npz

``--refshape``: the same frame with the reference's OWN camera shapes instead of the "query size / 0.3" stand-in:
SfM camera 1 = 3072 x 3072, f 2700, c 1536 (scripts/create_sfm_from_obj.py:154-159) x 0.3 -> a 921 x 921 reference
render and UNet pass (int(921.6): run_vis_on_poses.py:30-32), query 640 x 480 with the YCB-Video intrinsics
fx 1066.778 / fy 1067.487 and the principal point forced to (319.5, 239.5) (pixtrack/utils/io.py:46-50).
PXT_ORACLE_PROCS=n deals the renders' rows to n processes (bit-identical).
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np

from oracle import frame_oracle as FO
from oracle import ngp_oracle as NO
from pixtrack_amd.synthetic import (CRACKER_BOX_AABB, REF_CAMERA_YCB, YCB_QUERY_FXY, make_tracking_assets,
                                    perturb_pose)
from pixtrack_amd.utils.ingp_utils import get_nerf_aabb_from_sfm

REFSHAPE = "--refshape" in sys.argv
OUT = ROOT / "tests" / "golden" / ("ycb_refshape_921.npz" if REFSHAPE else "ycb_640x480.npz")
SEED, W, H, SPP = 1021, 640, 480, 8


def ycb_assets(n_frames=12):
    kw = dict(ref_camera=REF_CAMERA_YCB, query_f=YCB_QUERY_FXY[0]) if REFSHAPE else {}
    return make_tracking_assets(seed=SEED, width=W, height=H, n_frames=n_frames, aabb=CRACKER_BOX_AABB,
                                reference_scale=0.3, n_points=5600, **kw)


def ycb_query_camera(assets):
    f = float(assets["query_camera"]["params"][0])
    fx, fy = YCB_QUERY_FXY if REFSHAPE else (f, f)
    return dict(model="OPENCV", width=W, height=H, params=np.array([fx, fy, 319.5, 239.5]))


def nearest_reference(assets, R):
    from pixtrack_amd.utils.pose_utils import geodesic_distance_for_rotations

    dbs = assets["model3d"].dbs
    return min(dbs, key=lambda r: geodesic_distance_for_rotations(R, dbs[r].qvec2rotmat()))


def main():
    t_all = time.time()
    assets = ycb_assets()
    aabb = get_nerf_aabb_from_sfm(assets["model3d"], assets["nerf2sfm"])
    cam = ycb_query_camera(assets)
    ngp = FO.ngp_model(assets["snapshot"])
    gt = assets["gt_poses"]
    rng = np.random.default_rng(SEED + 77)
    # query = oracle render at the GT pose of frame 1 with the YCB camera, inside the SfM box
    qcam = FO.colmap_camera_to_pix(cam)
    rgba = NO.render(ngp, FO.nerf_view(assets["snapshot"], assets["nerf2sfm"], aabb, gt[1][0], gt[1][1], qcam, 0, SPP))
    u8 = FO.to_u8(rgba).astype(np.float32)
    query = np.clip(np.rint(u8 + rng.normal(size=u8.shape) * 2.0), 0, 255).astype(np.uint8)
    print("query rendered", round(time.time() - t_all, 1), "s", flush=True)
    # the tracker's pose before the frame: GT of frame 0 moved by a small fixed twist
    R0, t0 = perturb_pose(gt[0][0], gt[0][1], np.random.default_rng(SEED + 78), 0.15, 0.001, assets["center"])
    ref_id = nearest_reference(assets, R0)
    keep, tm = {}, {}
    ret = FO.track_frame(assets, R0, t0, query.astype(np.float32), ref_id, multiscale=(1,), use_mask=True, spp=SPP,
                         keep=keep, timings=tm, reference_scale=0.3, aabb=aabb, query_camera=cam)
    print("ycb frame", ret["success"], ret["cost"], ret["iters"], {k: round(v, 1) for k, v in tm.items()}, flush=True)
    assert ret["success"]
    np.savez_compressed(OUT, seed=SEED, width=W, height=H, spp=SPP, aabb=np.asarray(aabb), query=query, R0=R0, t0=t0,
                        ref_id=ref_id, R=ret["R"].numpy(), t=ret["t"].numpy(), cost=ret["cost"], iters=np.array(ret["iters"]),
                        n_points=ret["n_points"], mask_bits=np.packbits(ret["mask"].astype(np.uint8)),
                        mask_sum=int(ret["mask"].sum()), ref_rgba=keep["ref_rgba"].astype(np.float16),
                        ref_wh=np.array(keep["ref_rgba"].shape[1::-1]), query_params=np.asarray(cam["params"]),
                        gt_R=np.stack([g[0] for g in gt]), gt_t=np.stack([g[1] for g in gt]))
    print("wrote", OUT, round(OUT.stat().st_size / 1e6, 2), "MB; total", round(time.time() - t_all, 1), "s")


if __name__ == "__main__":
    main()
