"""Generates the golden vectors under tests/golden/ by IMPORTING the reference's own pure
helpers in the dev container: /root/reference/pixtrack/utils/
{pose_utils,ingp_utils,colmap_read_model}.py, with stub modules for the third-party imports
they do not need for these functions (pycolmap, cv2, pixloc, commentjson, pyngp, common,
scenes, sklearn).  /root/reference does not exist on the GPU box, so only the produced
.npz / .bin DATA is committed; this script is the provenance record.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT.parent.parent))


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _StubPose:  # what pose_utils needs from pixloc's Pose: .cpu().numpy() -> (R, t)
    def __init__(self, R, t):
        self.R_, self.t_ = R, t

    @classmethod
    def from_Rt(cls, R, t):
        return cls(np.asarray(R), np.asarray(t))

    def cpu(self):
        return self

    def numpy(self):
        return self.R_, self.t_


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    os.environ.setdefault("PROJECT_ROOT", "/tmp")
    stub("pycolmap")
    stub("cv2")
    stub("commentjson")
    stub("pyngp")
    stub("common", np=np, ROOT_DIR="/tmp")
    stub("scenes", scenes_nerf={})
    stub("sklearn")
    stub("sklearn.cluster", DBSCAN=object)
    stub("pixloc")
    stub("pixloc.pixlib")
    stub("pixloc.pixlib.geometry", Pose=_StubPose)
    stub("pixtrack")
    stub("pixtrack.utils")
    pose_utils = load(REF / "pixtrack/utils/pose_utils.py", "pixtrack.utils.pose_utils")
    sys.modules["pixtrack.utils.pose_utils"] = pose_utils
    ingp_utils = load(REF / "pixtrack/utils/ingp_utils.py", "pixtrack.utils.ingp_utils")
    crm = load(REF / "pixtrack/utils/colmap_read_model.py", "colmap_read_model")

    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(20260928)
    n = 32
    Ra = Rotation.random(n, random_state=1).as_matrix()
    Rb = Rotation.random(n, random_state=2).as_matrix()
    # include near-identity and near-pi pairs
    Rb[0] = Ra[0]
    Rb[1] = Ra[1] @ Rotation.from_rotvec([1e-9, 0, 0]).as_matrix()
    Rb[2] = Ra[2] @ Rotation.from_rotvec([0, np.pi - 1e-6, 0]).as_matrix()
    geo = np.array([pose_utils.geodesic_distance_for_rotations(a, b) for a, b in zip(Ra, Rb)])
    t = rng.normal(size=(n, 3)) * 3
    wIc = np.stack([pose_utils.get_world_in_camera_from_pixpose(_StubPose(R, tt)) for R, tt in zip(Ra, t)])
    cIw = np.stack([pose_utils.get_camera_in_world_from_pixpose(_StubPose(R, tt)) for R, tt in zip(Ra, t)])
    # sfm <-> nerf maps with random nerf2sfm records (colmap2ingp.py:356-362 schema)
    n2s, nerf, back = [], [], []
    for i in range(n):
        Rr = np.eye(4)
        Rr[:3, :3] = Rotation.random(random_state=100 + i).as_matrix()
        d = {"up": rng.normal(size=3), "centroid": rng.normal(size=3), "avglen": float(rng.uniform(0.5, 6)),
             "totp": rng.normal(size=3), "R": Rr}
        p = ingp_utils.sfm_to_nerf_pose(d, cIw[i].copy())
        q = ingp_utils.nerf_to_sfm_pose(d, p.copy())
        n2s.append(np.concatenate([d["up"], d["centroid"], [d["avglen"]], d["totp"], Rr.ravel()]))
        nerf.append(p)
        back.append(q)
    q = Rotation.random(n, random_state=3).as_quat()  # x,y,z,w
    qvec = np.concatenate([q[:, 3:], q[:, :3]], 1)  # COLMAP order w,x,y,z
    rot = np.stack([crm.qvec2rotmat(v) for v in qvec])
    qback = np.stack([crm.rotmat2qvec(r) for r in rot])
    np.savez(OUT / "pose_helpers.npz", Ra=Ra, Rb=Rb, geodesic=geo, t=t, wIc=wIc, cIw=cIw, nerf2sfm=np.stack(n2s),
             nerf_pose=np.stack(nerf), sfm_pose_back=np.stack(back), qvec=qvec, rotmat=rot, qvec_back=qback)

    # a tiny COLMAP binary model: written by THIS repo's writer, read back by the reference's reader
    from pixtrack_amd.utils.colmap import ColmapCamera, ColmapImage, ColmapPoint3D, write_model_binary

    cams = {1: ColmapCamera(1, "SIMPLE_RADIAL", 640, 480, np.array([768.0, 320.0, 240.0, -0.01])),
            2: ColmapCamera(2, "OPENCV", 320, 240, np.array([300.0, 310.0, 160.0, 120.0, 0.01, -0.02, 0.001, 0.002]))}
    imgs, pts = {}, {}
    for i in range(1, 5):
        m = 6 + i
        ids = rng.integers(-1, 12, size=m).astype(np.int64)
        imgs[i] = ColmapImage(i, qvec[i], t[i], 1 + (i % 2), f"mapping/{i:04d}.png", rng.uniform(0, 600, size=(m, 2)), ids)
    for p in range(12):
        k = 2 + p % 3
        pts[p] = ColmapPoint3D(p, rng.normal(size=3), rng.integers(0, 255, size=3), float(rng.uniform(0, 2)),
                               rng.integers(1, 5, size=k).astype(np.int64), rng.integers(0, 6, size=k).astype(np.int64))
    model_dir = OUT / "colmap_tiny"
    write_model_binary(model_dir, cams, imgs, pts)
    rc, ri, rp = crm.read_model(str(model_dir), ext=".bin")
    rec = {}
    for cid, c in rc.items():
        rec[f"cam{cid}_model"] = np.array(c.model)
        rec[f"cam{cid}_wh"] = np.array([c.width, c.height])
        rec[f"cam{cid}_params"] = np.asarray(c.params)
    for iid, im in ri.items():
        rec[f"img{iid}_qvec"] = im.qvec
        rec[f"img{iid}_tvec"] = im.tvec
        rec[f"img{iid}_cam"] = np.array(im.camera_id)
        rec[f"img{iid}_name"] = np.array(im.name)
        rec[f"img{iid}_xys"] = im.xys
        rec[f"img{iid}_p3d"] = im.point3D_ids
        rec[f"img{iid}_R"] = im.qvec2rotmat()
    for pid, p in rp.items():
        rec[f"pt{pid}_xyz"] = p.xyz
        rec[f"pt{pid}_rgb"] = p.rgb
        rec[f"pt{pid}_err"] = np.array(p.error)
        rec[f"pt{pid}_img"] = p.image_ids
        rec[f"pt{pid}_idx"] = p.point2D_idxs
    np.savez(OUT / "colmap_tiny_expected.npz", **rec)
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
