"""python -m pixtrack_amd.doctor: the pre-flight for real assets.  On the output of
synthetic.write_object_dir (the reference's on-disk layout, both third-party importers exercised) every check passes;
deliberately broken inputs each produce their OWN message, naming the assumption that breaks."""
import io
import os
import pickle
from contextlib import redirect_stdout

import msgpack
import numpy as np
import pytest
import torch

from pixtrack_amd import doctor
from pixtrack_amd.synthetic import make_tracking_assets, write_object_dir


@pytest.fixture(scope="module")
def object_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("obj")
    assets = make_tracking_assets(seed=1002, width=160, height=120, n_frames=2, n_points=800)
    write_object_dir(assets, str(d))
    return d, assets


def _run(d, assets, *extra, env=None):
    old = {k: os.environ.get(k) for k in ("OBJ_AABB", "UPRIGHT_REF_IMG", "PIXTRACK_WEIGHTS")}
    os.environ["OBJ_AABB"] = str(assets["aabb"])
    os.environ["UPRIGHT_REF_IMG"] = assets["upright_ref_img"]
    os.environ.pop("PIXTRACK_WEIGHTS", None)
    os.environ.update(env or {})
    buf = io.StringIO()
    try:
        with redirect_stdout(buf):
            rc = doctor.main(["--object_path", str(d), *extra])
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return rc, buf.getvalue()


def test_a_good_object_directory_passes(object_dir):
    d, assets = object_dir
    rc, out = _run(d, assets, "--no-gpu")
    assert rc == 0 and "PASSED" in out and "FAIL" not in out, out
    assert "13,074,912 + 10,240" in out and "keeps" in out and "16 image(s)" in out


def test_broken_inputs_name_what_breaks(object_dir, tmp_path):
    import shutil

    d, assets = object_dir
    snap = "pixtrack/instant-ngp/snapshots/weights.msgpack"

    def variant(name):
        v = tmp_path / name
        shutil.copytree(d, v)
        return v

    # (1) truncated params_binary
    v = variant("trunc")
    raw = msgpack.unpackb((v / snap).read_bytes(), raw=False, strict_map_key=False)
    raw["snapshot"]["params_binary"] = raw["snapshot"]["params_binary"][:-4096]
    (v / snap).write_bytes(msgpack.packb(raw, use_bin_type=True))
    rc, out = _run(v, assets, "--no-gpu")
    assert rc == 1 and "FAIL  snapshot import" in out and "params_binary holds" in out, out
    # (2) a density grid of the wrong size
    v = variant("grid")
    raw = msgpack.unpackb((v / snap).read_bytes(), raw=False, strict_map_key=False)
    raw["snapshot"]["density_grid_binary"] = raw["snapshot"]["density_grid_binary"][: 100 * 100 * 100 * 2]
    (v / snap).write_bytes(msgpack.packb(raw, use_bin_type=True))
    rc, out = _run(v, assets, "--no-gpu")
    assert rc == 1 and "density_grid_binary has" in out, out
    # (3) a checkpoint with a renamed key
    v = variant("ckpt")
    ck = torch.load(v / "pixtrack/pixloc_megadepth.pt", map_location="cpu", weights_only=False)
    ck["model"]["extractor.encoder.2.9.weight"] = ck["model"].pop("extractor.encoder.2.1.weight")
    torch.save(ck, v / "pixtrack/pixloc_megadepth.pt")
    rc, out = _run(v, assets, "--no-gpu")
    assert rc == 1 and "FAIL  checkpoint keys" in out and "encoder.2.1.weight" in out and "torchvision VGG16 index" in out, out
    # (4) a render box that misses the object, (5) an upright image that does not exist, (6) nerf2sfm without a key
    rc, out = _run(d, assets, "--no-gpu", env={"OBJ_AABB": "[[0.9, 0.9, 0.9], [0.95, 0.95, 0.95]]"})
    assert rc == 1 and "holds no occupied cell" in out, out
    rc, out = _run(d, assets, "--no-gpu", env={"UPRIGHT_REF_IMG": "mapping/IMG_0000.png"})
    assert rc == 1 and "is not an image of the model" in out, out
    v = variant("n2s")
    p = v / "pixtrack/pixsfm/dataset/nerf2sfm.pkl"
    n2s = pickle.loads(p.read_bytes())
    del n2s["totp"]
    p.write_bytes(pickle.dumps(n2s))
    rc, out = _run(v, assets, "--no-gpu")
    assert rc == 1 and "key 'totp' is missing" in out, out


@pytest.mark.gpu
def test_doctor_on_the_gpu(object_dir):
    d, assets = object_dir
    rc, out = _run(d, assets)
    assert rc == 0 and "thumbnail render" in out and "fp16 activations" in out and "FAIL" not in out, out
