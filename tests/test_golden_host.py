"""Host helpers against golden vectors captured from the reference's own functions
(tests/golden/make_golden.py imports /root/reference/pixtrack/utils/{pose_utils,ingp_utils,
colmap_read_model}.py under stubs; only the data is committed)."""
from pathlib import Path

import numpy as np
import torch

from pixtrack_amd.geometry import Pose
from pixtrack_amd.utils import colmap, ingp_utils, pose_utils

G = Path(__file__).resolve().parent / "golden"


def _n2s(row):
    return {"up": row[0:3], "centroid": row[3:6], "avglen": float(row[6]), "totp": row[7:10], "R": row[10:26].reshape(4, 4)}


def test_geodesic_distance_matches_reference():
    g = np.load(G / "pose_helpers.npz")
    got = np.array([pose_utils.geodesic_distance_for_rotations(a, b) for a, b in zip(g["Ra"], g["Rb"])])
    assert np.allclose(got, g["geodesic"], atol=1e-12)


def test_pixpose_matrices_match_reference():
    g = np.load(G / "pose_helpers.npz")
    for R, t, wIc, cIw in zip(g["Ra"], g["t"], g["wIc"], g["cIw"]):
        p = Pose.from_Rt(R, t)
        assert np.allclose(pose_utils.get_world_in_camera_from_pixpose(p), wIc, atol=1e-14)
        assert np.allclose(pose_utils.get_camera_in_world_from_pixpose(p), cIw, atol=1e-12)
        back = pose_utils.get_pixpose_from_camera_in_world(cIw)
        assert np.allclose(back.numpy()[0], R, atol=1e-12) and np.allclose(back.numpy()[1], t, atol=1e-12)


def test_sfm_nerf_maps_match_reference():
    g = np.load(G / "pose_helpers.npz")
    for row, cIw, nerf, back in zip(g["nerf2sfm"], g["cIw"], g["nerf_pose"], g["sfm_pose_back"]):
        d = _n2s(row)
        assert np.allclose(ingp_utils.sfm_to_nerf_pose(d, cIw), nerf, atol=1e-12)
        assert np.allclose(ingp_utils.nerf_to_sfm_pose(d, nerf), back, atol=1e-12)
        assert np.allclose(back, cIw, atol=1e-9)  # the pair is a round trip


def test_frame_oracle_restatement_matches_reference_too():
    from oracle import frame_oracle as FO

    g = np.load(G / "pose_helpers.npz")
    for row, cIw, nerf in zip(g["nerf2sfm"], g["cIw"], g["nerf_pose"]):
        assert np.allclose(FO.sfm_to_nerf_pose(_n2s(row), cIw), nerf, atol=1e-12)


def test_quaternion_helpers_match_reference():
    g = np.load(G / "pose_helpers.npz")
    for q, R, qb in zip(g["qvec"], g["rotmat"], g["qvec_back"]):
        assert np.allclose(colmap.qvec2rotmat(q), R, atol=1e-14)
        got = colmap.rotmat2qvec(R)
        assert np.allclose(got, qb, atol=1e-9) or np.allclose(got, -qb, atol=1e-9)


def test_colmap_binary_reader_matches_reference_reader():
    e = np.load(G / "colmap_tiny_expected.npz")
    cams, imgs, pts = colmap.read_model(G / "colmap_tiny")
    assert sorted(cams) == [1, 2] and sorted(imgs) == [1, 2, 3, 4] and len(pts) == 12
    for cid, c in cams.items():
        assert c.model == str(e[f"cam{cid}_model"]) and [c.width, c.height] == list(e[f"cam{cid}_wh"])
        assert np.array_equal(c.params, e[f"cam{cid}_params"])
    for iid, im in imgs.items():
        assert np.array_equal(im.qvec, e[f"img{iid}_qvec"]) and np.array_equal(im.tvec, e[f"img{iid}_tvec"])
        assert im.camera_id == int(e[f"img{iid}_cam"]) and im.name == str(e[f"img{iid}_name"])
        assert np.array_equal(im.xys, e[f"img{iid}_xys"]) and np.array_equal(im.point3D_ids, e[f"img{iid}_p3d"])
        assert np.allclose(im.qvec2rotmat(), e[f"img{iid}_R"], atol=1e-14)
    for pid, p in pts.items():
        assert np.array_equal(p.xyz, e[f"pt{pid}_xyz"]) and np.array_equal(p.rgb, e[f"pt{pid}_rgb"])
        assert p.error == float(e[f"pt{pid}_err"])
        assert np.array_equal(p.image_ids, e[f"pt{pid}_img"]) and np.array_equal(p.point2D_idxs, e[f"pt{pid}_idx"])


def test_model3d_queries():
    from pixtrack_amd.model3d import Model3D, extract_covisibility

    m = Model3D(G / "colmap_tiny")
    assert m.name2id["mapping/0002.png"] == 2
    sel = m.get_p3did_to_dbids([1], None, None, "all", 3)
    for pid, dbs in sel.items():
        assert 1 in dbs and len(m.points3D[pid].image_ids) >= 3 and pid in set(m.dbs[1].point3D_ids.tolist())
    inv = m.get_dbid_to_p3dids(sel)
    assert set(inv.get(1, [])) == set(sel)
    cov = extract_covisibility(m)
    assert set(cov) == {1, 2, 3, 4}


def test_oracle_reference_id_policy_equals_the_hosts():
    """oracle/frame_oracle.covisibility + nearest_reference (the restatement of update_reference_ids,
    pixloc_tracker_r9.py:120-143) against the host's extract_covisibility + its vectorised ranking, along an orbit of the
    thin-slab box (config/roncelli_blankk.sh) that crosses from the upright reference to its neighbour - the switch behind
    the stall of profiles/r06_drift_probe_roncelli_blankk.log."""
    from oracle import frame_oracle as FO
    from pixtrack_amd import parallel
    from pixtrack_amd.model3d import extract_covisibility
    from pixtrack_amd.synthetic import make_tracking_assets
    from pixtrack_amd.utils.pose_utils import geodesic_distances_to

    obj = next(o for o in parallel.load_object_configs() if o["name"] == "roncelli_blankk")
    assets = make_tracking_assets(seed=1008, width=160, height=120, n_frames=60, aabb=obj["aabb"], n_points=1500)
    m = assets["model3d"]
    cov_o, cov_h = FO.covisibility(m), extract_covisibility(m)
    assert {k: v for k, v in cov_h.items() if v} == cov_o
    assert all(list(cov_h[k]) == list(cov_o[k]) for k in cov_o)  # (same insertion order: the ranking's ties keep it)
    ref_o = ref_h = m.name2id[assets["upright_ref_img"]]
    seen = set()
    for Rg, _ in assets["gt_poses"]:
        ref_o = FO.nearest_reference(m, cov_o, ref_o, Rg)
        ids = list(dict.fromkeys([ref_h] + [k for k, v in cov_h[ref_h].items() if v > 50]))
        ref_h = ids[int(np.argmin(geodesic_distances_to(Rg, np.stack([m.dbs[r].qvec2rotmat() for r in ids]))))]
        assert ref_o == ref_h
        seen.add(ref_o)
    assert len(seen) >= 2  # the orbit does leave the upright reference
