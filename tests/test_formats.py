"""Data formats either side of the hot path: pixloc-path pickles,
pixloc checkpoints, instant-ngp snapshot layout, the GetMetrics evaluation.  CPU only."""
import io
import pickle
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from pixtrack_amd import evaluation, ngp
from pixtrack_amd.geometry import Camera, Pose
from pixtrack_amd.synthetic import make_synthetic_nerf, rodrigues
from pixtrack_amd.unet import from_pixloc_state_dict, load_weights, make_synthetic_unet_weights, to_pixloc_state_dict
from pixtrack_amd.utils.io import dump_reference_pickle, load_reference_pickle


# ------------------------------------------------------------------ pickles
def _history():
    T = Pose.from_Rt(torch.eye(3), torch.tensor([0.1, 0.2, 0.3]))
    cam = Camera.from_colmap(dict(model="SIMPLE_RADIAL", width=64, height=48, params=[70.0, 32.0, 24.0, 0.0]))
    return {"frame0.png": {"success": True, "T_refined": T, "camera": cam, "reference_ids": [1],
                           "query_path": "frame0.png", "diff_R": 0.0}}


def test_pickle_uses_pixloc_class_path_and_round_trips(tmp_path):
    path = tmp_path / "poses.pkl"
    dump_reference_pickle(_history(), path)
    raw = path.read_bytes()
    assert b"pixloc.pixlib.geometry.wrappers" in raw and b"pixtrack_amd" not in raw
    assert "pixloc" not in sys.modules  # the stand-in modules are gone again
    back = load_reference_pickle(path)
    T = back["frame0.png"]["T_refined"]
    assert isinstance(T, Pose) and isinstance(back["frame0.png"]["camera"], Camera)
    assert torch.equal(T._data, _history()["frame0.png"]["T_refined"]._data)


def test_pickle_loads_in_a_process_that_only_has_pixloc(tmp_path):
    """A reader with pixloc's class at that path and *without* this package gets the tensor."""
    path = tmp_path / "poses.pkl"
    dump_reference_pickle(_history(), path)
    reader = textwrap.dedent(f"""
        import sys, types, pickle
        for name in ("pixloc", "pixloc.pixlib", "pixloc.pixlib.geometry", "pixloc.pixlib.geometry.wrappers"):
            sys.modules[name] = types.ModuleType(name)
        class Pose:
            pass
        class Camera:
            pass
        Pose.__module__ = Camera.__module__ = "pixloc.pixlib.geometry.wrappers"
        sys.modules["pixloc.pixlib.geometry.wrappers"].Pose = Pose
        sys.modules["pixloc.pixlib.geometry.wrappers"].Camera = Camera
        d = pickle.load(open({str(path)!r}, "rb"))
        T = d["frame0.png"]["T_refined"]
        assert type(T) is Pose and tuple(T._data.shape) == (12,), T.__dict__
        assert abs(float(T._data[9]) - 0.1) < 1e-7
        assert "pixtrack_amd" not in sys.modules
        print("ok")
    """)
    out = subprocess.run([sys.executable, "-c", reader], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_plain_pickle_from_this_package_also_loads():
    buf = io.BytesIO()
    pickle.dump(_history(), buf)
    buf.seek(0)
    assert isinstance(load_reference_pickle(buf)["frame0.png"]["T_refined"], Pose)


# ------------------------------------------------------------------ pixloc checkpoint
def test_pixloc_checkpoint_round_trip(tmp_path):
    w = make_synthetic_unet_weights(3)
    for i in range(3):
        w[f"optimizer.{i}.dampingnet.const"] = torch.full((6,), -2.0 + i)
    sd = to_pixloc_state_dict(w)
    assert "extractor.encoder.0.0.weight" in sd and "extractor.encoder.1.1.weight" in sd
    assert "extractor.encoder.2.5.bias" in sd and "extractor.decoder.3.layers.1.running_var" in sd
    assert "extractor.adaptation.2.0.weight" in sd and "extractor.uncertainty.0.0.bias" in sd
    path = tmp_path / "checkpoint_best.tar"
    torch.save({"model": sd, "conf": {"name": "two_view_refiner"}, "epoch": 7}, path)
    back = load_weights(path)
    assert set(back) == set(w)
    for k in w:
        assert torch.equal(back[k], w[k]), k
    assert set(from_pixloc_state_dict(sd)) == {k for k in w if not k.startswith("optimizer.")}
    flat = tmp_path / "flat.pt"
    torch.save(w, flat)
    assert set(load_weights(flat)) == set(w)


# ------------------------------------------------------------------ instant-ngp snapshot layout
@pytest.fixture(scope="module")
def snap():
    return make_synthetic_nerf(seed=21)


def test_morton_table_known_answers():
    m = ngp._morton_index_table()  # [z, y, x]
    assert m[0, 0, 1] == 1 and m[0, 1, 0] == 2 and m[1, 0, 0] == 4
    assert m[0, 0, 3] == 0b001001 and m[0, 2, 0] == 0b010000 and m[5, 0, 0] == 0b100000100
    assert m[127, 127, 127] == 128**3 - 1
    assert np.array_equal(np.sort(m.reshape(-1)), np.arange(128**3))


def test_instant_ngp_param_split_and_round_trip(snap, tmp_path):
    import msgpack

    d = ngp.to_instant_ngp(snap)
    assert d["snapshot"]["n_params"] == 13074912 + 10240  # tcnn: total_encoding_params + total_network_params
    path = tmp_path / "weights.msgpack"
    path.write_bytes(msgpack.packb(d, use_bin_type=True))
    back = ngp.load_snapshot_file(str(path))
    assert np.array_equal(back.grid, snap.grid) and np.array_equal(back.mlp, snap.mlp)
    assert (back.n_levels, back.log2_hashmap, back.base_res, back.cascades) == (16, 19, 16, 3)
    assert back.aabb_scale == 4.0 and back.cone_angle == 1.0 / 256.0 and back.scale == snap.scale
    a = np.unpackbits(snap.occupancy, bitorder="little")
    b = np.unpackbits(back.occupancy, bitorder="little")
    assert not np.any(a & ~b & 1)  # every occupied cell survives
    # the only additions are the max-pool of a finer cascade into the centre half of the next one
    extra = (b & ~a & 1).reshape(3, 128, 128, 128)
    assert extra[0].sum() == 0 and extra[:, :32].sum() == 0 and extra[:, 96:].sum() == 0
    fine = a.reshape(3, 128, 128, 128)
    for c in (1, 2):
        zs, ys, xs = np.nonzero(extra[c])
        for z, y, x in zip(zs, ys, xs):
            blk = fine[c - 1, 2 * (z - 32):2 * (z - 32) + 2, 2 * (y - 32):2 * (y - 32) + 2, 2 * (x - 32):2 * (x - 32) + 2]
            assert blk.any()


def test_snapshot_colour_space_flag(snap, tmp_path):
    """instant-ngp's shade kernel converts a finished ray's colour sRGB -> linear unless the snapshot was trained in
    linear colours (training.linear_colors = dataset.is_hdr).  Default (LDR images, pixtrack's case): convert."""
    import dataclasses

    d = ngp.to_instant_ngp(snap)
    assert d["snapshot"]["nerf"]["dataset"]["is_hdr"] is False and ngp.from_instant_ngp(d).linear_colors is False
    del d["snapshot"]["nerf"]["dataset"]["is_hdr"]  # a snapshot without the key: convert
    assert ngp.from_instant_ngp(d).linear_colors is False
    d["snapshot"]["nerf"]["dataset"]["is_hdr"] = True
    assert ngp.from_instant_ngp(d).linear_colors is True
    d["snapshot"]["nerf"]["linear_colors"] = False  # an explicit flag wins over the dataset's
    assert ngp.from_instant_ngp(d).linear_colors is False
    hdr = dataclasses.replace(snap, linear_colors=True)
    assert ngp.from_instant_ngp(ngp.to_instant_ngp(hdr)).linear_colors is True
    path = tmp_path / "own.msgpack"
    ngp.save_snapshot(str(path), hdr)
    assert ngp.load_snapshot_file(str(path)).linear_colors is True
    ngp.save_snapshot(str(path), snap)
    assert ngp.load_snapshot_file(str(path)).linear_colors is False


def test_instant_ngp_density_threshold_and_dtype(snap):
    d = ngp.to_instant_ngp(snap)
    cells = 128**3
    dens = np.zeros((3, cells), np.float32)
    m = ngp._morton_index_table()
    dens[0, m[10, 20, 30]] = 0.5      # above min(0.01, mean)
    dens[0, m[11, 20, 30]] = 1e-9     # below the mean-derived threshold
    dens[1, m[64, 64, 64]] = -1.0     # negative: never occupied
    d["snapshot"]["density_grid_binary"] = dens.tobytes()  # fp32 flavour
    occ = np.unpackbits(ngp.from_instant_ngp(d).occupancy, bitorder="little").reshape(3, 128, 128, 128)
    assert occ[0, 10, 20, 30] == 1 and occ[0, 11, 20, 30] == 0 and occ[1, 64, 64, 64] == 0
    assert occ[1, 32 + 5, 32 + 10, 32 + 15] == 1  # pooled into cascade 1 ...
    assert occ[2, 32 + 18, 32 + 21, 32 + 23] == 1  # ... and cascade 2
    assert occ.sum() == 3


def test_instant_ngp_derives_per_level_scale_and_rejects_bad_sizes(snap):
    d = ngp.to_instant_ngp(snap)
    del d["encoding"]["per_level_scale"]
    assert abs(ngp.from_instant_ngp(d).per_level_scale - 1.51572) < 1e-5  # (2048*4/16)^(1/15)
    d["snapshot"]["params_binary"] = d["snapshot"]["params_binary"][:-2]
    with pytest.raises(Exception, match="params_binary"):
        ngp.from_instant_ngp(d)
    d = ngp.to_instant_ngp(snap)
    d["snapshot"]["density_grid_binary"] = d["snapshot"]["density_grid_binary"][:-6]
    with pytest.raises(Exception, match="density_grid_binary"):
        ngp.from_instant_ngp(d)
    d = ngp.to_instant_ngp(snap)
    d["rgb_network"]["n_hidden_layers"] = 3
    with pytest.raises(Exception, match="MLP shape"):
        ngp.from_instant_ngp(d)


# ------------------------------------------------------------------ GetMetrics
def _pose(R, t):
    return Pose.from_Rt(torch.from_numpy(R), torch.from_numpy(t))


def test_similarity_transform_recovers_known_map():
    rng = np.random.default_rng(0)
    P = rng.normal(size=(40, 3))
    R = rodrigues(np.array([0.3, -0.2, 0.5]))
    Q = 1.7 * P @ R.T + np.array([0.4, -1.0, 2.0])
    R2, c2, t2 = evaluation.similarity_transform(P, Q)
    assert np.allclose(R2, R, atol=1e-10) and abs(c2 - 1.7) < 1e-10 and np.allclose(t2, [0.4, -1.0, 2.0], atol=1e-10)
    with pytest.raises(ValueError):
        evaluation.similarity_transform(np.outer(np.arange(5.0), [1, 0, 0]), np.outer(np.arange(5.0), [1, 0, 0]))


def test_get_metrics_counts_bad_frames():
    rng = np.random.default_rng(1)
    verts = np.c_[rng.uniform(-0.05, 0.05, size=(200, 3)), np.ones(200)]
    poses = {}
    for i in range(12):
        R = rodrigues(rng.normal(size=3) * 0.4)
        t = np.array([0.0, 0.0, 0.6]) + rng.normal(size=3) * 0.05
        Re, te = R, t.copy()
        if i == 5:
            te = te + np.array([0.03, 0.0, 0.0])  # 3 cm off
        poses[f"f{i}"] = {"success": i != 9, "T_refined": _pose(Re, te), "gt_pose": _pose(R, t)}
    m = evaluation.get_metrics(poses, verts, tr_threshold=1.0, rot_threshold=2.0)
    assert m["total_frames"] == 12 and m["bad_count"] == 1
    assert abs(m["accuracy"] - 11 / 12) < 1e-12
    assert 2.0 < m["max_translation_error"] < 3.5 and m["average_error_vertices"] < 1.0


def test_adds_is_zero_under_a_model_symmetry():
    g = np.linspace(-1, 1, 5)
    verts = np.array([[x, y, z] for x in g for y in g for z in g])  # cube lattice: 90 deg symmetric
    T0 = np.eye(4)
    T1 = np.eye(4)
    T1[:3, :3] = rodrigues(np.array([0.0, 0.0, np.pi / 2]))
    assert evaluation.adds_distance(T1, T0, verts) < 1e-12
    T1[:3, 3] = [0.1, 0, 0]
    assert abs(evaluation.adds_distance(T1, T0, verts) - 0.1) < 1e-12


def test_overlay_drawing_helpers():
    """Host-side pieces of the overlay renderer (reference run_vis_on_poses.py:60-253): projection with the
    centred pinhole, blend weights, axes / centre / text drawn where the pose puts them."""
    from pixtrack_amd.geometry import Camera
    from pixtrack_amd.visualization import run_vis_on_poses as V

    cam = Camera.from_colmap(dict(model="SIMPLE_RADIAL", width=64, height=48, params=np.array([80.0, 32.0, 24.0, 0.0])))
    K = V.pinhole_K(cam)
    assert np.allclose(K, [[80, 0, 32], [0, 80, 24], [0, 0, 1]])
    assert np.allclose(V.project_3d_to_2d(np.array([[0.0, 0.0, 2.0], [0.5, -0.25, 2.0]]), K), [[32, 24], [52, 14]])
    q, n = np.full((48, 64, 3), 200, np.uint8), np.full((48, 64, 3), 100, np.uint8)
    assert (V.blend_images(q, n) == 130).all()  # 0.3 * query + 0.7 * render
    cIw = np.eye(4)
    cIw[:3, 3] = [0.0, 0.0, -2.0]  # camera 2 units in front of the origin, looking along +z
    img = V.add_pose_axes(np.zeros((48, 64, 3), np.uint8), cam, cIw, [0, 0, 0, 0], length=0.25, thickness=1)
    assert tuple(img[24, 37]) == (0, 0, 255)   # +x axis: to the right of the centre, the reference's BGR (255,0,0)
    assert tuple(img[19, 32]) == (0, 255, 0)   # -y axis: upwards
    assert img[30:, :].sum() == 0
    dot = V.add_object_center(np.zeros((48, 64, 3), np.uint8), cam, cIw, [0.0, 0.0, 0.0])
    assert tuple(dot[24, 32]) == (255, 255, 255) and dot[:15].sum() == 0
    txt = V.add_text_lines(np.zeros((48, 64, 3), np.uint8), ["Rotation error: 0.1000 degrees"], origin=(2, 2))
    assert txt[..., 2].sum() > 0 and txt[..., 0].sum() == 0
    inset = V.add_reference_image(np.zeros((48, 64, 3), np.uint8), np.full((40, 80, 3), 90, np.uint8), "mapping/0001.png")
    assert (inset[0, :16] == 90).all() and (inset[20:, 20:] == 0).all()


def test_reference_feature_cache_h5_branch_with_an_api_mock(tmp_path, monkeypatch):
    """the `.h5` branch of read_features / write_features (reference
    pixloc_pose_refiners.py:175-198).  h5py is not in this image, so the branch runs against a small
    in-memory mock of the h5py calls it makes (File context manager, nested groups, create_dataset, keys):
    this executes the hierarchy logic f[ref_id][scale]["p3dids"] / [level]["p3did_to_feat"]; real HDF5
    I/O stays untested here."""
    import sys
    import types

    store = {}

    class Node(dict):
        def create_dataset(self, key, data=None):
            node = self
            parts = key.split("/")
            for p_ in parts[:-1]:
                node = node.setdefault(p_, Node())
            node[parts[-1]] = np.asarray(data)

    class File(Node):
        def __init__(self, path, mode="r"):
            super().__init__()
            self.path, self.mode = path, mode
            if mode == "r":
                self.update(store[path])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            if self.mode == "w":
                store[self.path] = Node(self)
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    from pixtrack_amd import refiner as RF
    from pixtrack_amd.unet import OUTPUT_DIMS

    class P3D:
        def __init__(self, xyz):
            self.xyz = xyz

    r = RF.PoseTrackerRefiner.__new__(RF.PoseTrackerRefiner)
    r.paths = {"dumps": tmp_path}
    r.conf = types.SimpleNamespace(multiscale=[1])
    r.device = torch.device("cpu")
    r.model3d = types.SimpleNamespace(points3D={i: P3D(np.array([i, 2 * i, 3 * i], float)) for i in range(7)})
    rng = np.random.default_rng(0)
    ids = [5, 1, 3, 6]
    packed = []
    for c in OUTPUT_DIMS:
        rec = torch.zeros(len(ids), RF.cstride_for(c))
        d = torch.from_numpy(rng.normal(size=(len(ids), c)).astype(np.float32))
        rec[:, :c] = d / d.norm(dim=1, keepdim=True)
        rec[:, c] = torch.from_numpy(rng.uniform(size=len(ids)).astype(np.float32))
        packed.append(rec)
    feats = RF.SparseReferenceFeatures(packed, torch.tensor([1, 1, 0, 1], dtype=torch.uint8), ids, torch.zeros(4, 3), OUTPUT_DIMS)
    written = r.write_features({9: {"1": feats}})
    assert written.endswith("reference_features.h5") and written in store
    (tmp_path / "reference_features.h5").write_bytes(b"")  # the reader looks for the file's existence
    back = r.read_features(9)["1"]
    assert back.p3dids_all == [5, 1, 6]  # the invalid point was dropped by the writer
    for lvl, c in enumerate(OUTPUT_DIMS):
        assert torch.allclose(back.packed[lvl][:, : c + 1], packed[lvl][[0, 1, 3], : c + 1], atol=1e-6)
    assert np.allclose(back.p3d.numpy(), [[5, 10, 15], [1, 2, 3], [6, 12, 18]])
