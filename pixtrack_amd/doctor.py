"""Pre-flight for REAL assets:  python -m pixtrack_amd.doctor --object_path P [--query DIR] [--no-gpu]

Reads every file the tracker's CLI reads (pixtrack/pose_trackers/pixloc_tracker_r9.py:77-86, 288-301; setup.sh:13-20) and
reports, without tracking, what it found and - where a file does not look like what this package's importers assume - WHICH
assumption breaks.  The importers for the two third-party formats follow their layouts from recall (instant-ngp snapshots,
pixloc checkpoints: neither project's source is in the reference tree), so a first user with real assets
needs to be told what is wrong, not that "something" is:

  COLMAP model   aug_nerf_sfm/aug_sfm/{cameras,images,points3D}.bin: counts, camera models, the upright reference image
  nerf2sfm.pkl   keys up / centroid / avglen / totp / R (colmap2ingp.py:356-362)
  snapshot       instant-ngp weights.msgpack: parameter count against the network configuration (13,074,912 + 10,240 for
                 the reference's shape, notebooks/Render YCB GT Poses .ipynb:147-150), density-grid size, aabb_scale,
                 colour-space flag, lens record; $OBJ_AABB against the occupied cells (how many it keeps)
  checkpoint     pixloc_megadepth checkpoint: key coverage (missing / unexpected tensors), shapes, damping constants
  on the GPU     (unless --no-gpu) one thumbnail render with its coverage, UNet.activation_stats on that render with the fp16
                 verdict

Exit status 0 = nothing fatal; 1 = at least one check failed (each failure is one line starting with "FAIL").
"""
from __future__ import annotations

import argparse
import ast
import os
import sys
from pathlib import Path
from typing import List

import numpy as np

REF_GRID_PARAMS, REF_MLP_PARAMS = 13_074_912, 10_240  # hash grid / the two MLPs of the reference's network shape


class Report:
    def __init__(self):
        self.lines: List[str] = []
        self.failed = 0

    def ok(self, what, detail=""):
        self.lines.append(f"ok    {what}" + (f": {detail}" if detail else ""))

    def note(self, what, detail=""):
        self.lines.append(f"note  {what}" + (f": {detail}" if detail else ""))

    def fail(self, what, detail, assumption):
        self.failed += 1
        self.lines.append(f"FAIL  {what}: {detail}\n      assumption that breaks: {assumption}")

    def dump(self, out=None):
        out = out if out is not None else sys.stdout
        for l in self.lines:
            print(l, file=out)
        print(f"{'FAILED' if self.failed else 'PASSED'}: {self.failed} failing check(s)", file=out)


def check_colmap(rep: Report, sfm_dir: Path, upright: str | None):
    from .utils.colmap import read_model

    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        if not (sfm_dir / name).is_file():
            rep.fail("COLMAP model", f"{sfm_dir / name} is missing",
                     "the reference reads P/pixtrack/aug_nerf_sfm/aug_sfm/*.bin (pixloc_tracker_r9.py:299-301, Model3D)")
            return None
    try:
        cameras, images, points3D = read_model(str(sfm_dir))
    except Exception as e:
        rep.fail("COLMAP model", f"unreadable: {e!r}", "COLMAP binary model format (colmap_read_model.py:298-308)")
        return None
    models = sorted({c.model for c in cameras.values()})
    rep.ok("COLMAP model", f"{len(cameras)} camera(s) {models}, {len(images)} image(s), {len(points3D)} point(s)")
    supported = {"SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV"}
    bad = [m for m in models if m not in supported]
    if bad:
        rep.fail("camera model", f"{bad} is not implemented", f"geometry.Camera.from_colmap handles {sorted(supported)} (SURVEY A.1)")
    if 1 not in cameras:
        rep.fail("reference camera", "no camera with id 1",
                 "the reference image is rendered with model3d.cameras[1] x reference_scale (pixloc_tracker_r9.py:145-152)")
    else:
        c = cameras[1]
        rep.ok("reference camera", f"{c.model} {c.width} x {c.height}, params {np.round(c.params, 3).tolist()} "
                                   f"(x 0.5 -> a {c.width // 2} x {c.height // 2} reference render per frame)")
    long_tracks = sum(1 for p in points3D.values() if len(p.image_ids) >= 3)
    if long_tracks < 10:
        rep.fail("tracks", f"only {long_tracks} points have track length >= 3",
                 "the refiner samples points with track length >= 3 and fails below 10 (pixloc_pose_refiners.py:148-158)")
    else:
        rep.ok("tracks", f"{long_tracks} points with track length >= 3")
    names = {im.name for im in images.values()}
    if upright is None:
        rep.note("UPRIGHT_REF_IMG", "not set in the environment: the CLI needs it (pixloc_tracker_r9.py:77-78)")
    elif upright not in names:
        rep.fail("UPRIGHT_REF_IMG", f"{upright!r} is not an image of the model (e.g. {sorted(names)[:3]})",
                 "the cold start begins at the pose of model3d.name2id[$UPRIGHT_REF_IMG] (pixloc_tracker_r9.py:77-78, 95-106)")
    else:
        rep.ok("UPRIGHT_REF_IMG", upright)
    return cameras, images, points3D


def check_nerf2sfm(rep: Report, path: Path):
    import pickle

    if not path.is_file():
        rep.fail("nerf2sfm.pkl", f"{path} is missing", "P/pixtrack/pixsfm/dataset/nerf2sfm.pkl (pixloc_tracker_r9.py:79-80)")
        return None
    try:
        with open(path, "rb") as f:
            d = pickle.load(f)
    except Exception as e:
        rep.fail("nerf2sfm.pkl", f"unreadable: {e!r}", "a pickled dict (colmap2ingp.py:356-362)")
        return None
    want = {"centroid": (3,), "avglen": (), "totp": (3,), "R": (4, 4)}
    for k, shp in want.items():
        if k not in d:
            rep.fail("nerf2sfm.pkl", f"key {k!r} is missing (has {sorted(d)})", "keys up / centroid / avglen / totp / R (colmap2ingp.py:356-362)")
            return None
        if tuple(np.asarray(d[k]).shape) != shp:
            rep.fail("nerf2sfm.pkl", f"{k} has shape {np.asarray(d[k]).shape}, expected {shp}", "sfm_to_nerf_pose's arithmetic (ingp_utils.py:47-63)")
            return None
    rep.ok("nerf2sfm.pkl", f"avglen {float(d['avglen']):.4f}, centroid {np.round(np.asarray(d['centroid'], float), 3).tolist()}")
    return d


def check_snapshot(rep: Report, path: Path, aabb):
    from . import _lib
    from .ngp import _MLP_PARAMS, _grid_entries, load_snapshot_file

    if not path.is_file():
        rep.fail("snapshot", f"{path} is missing", "P/pixtrack/instant-ngp/snapshots/weights.msgpack (ingp_utils.py:27, train_ingp_nerf.sh)")
        return None
    try:
        import msgpack

        with open(path, "rb") as f:
            raw = msgpack.unpackb(f.read(), raw=False, strict_map_key=False)
    except Exception as e:
        rep.fail("snapshot", f"not a msgpack document: {e!r}", "instant-ngp writes snapshots with msgpack (testbed.cu save_snapshot)")
        return None
    for k in ("encoding", "snapshot"):
        if k not in raw:
            rep.fail("snapshot", f"top-level key {k!r} is missing (has {sorted(raw)})",
                     "the network configuration JSON (encoding / network / rgb_network) + snapshot.{params_binary, density_grid_binary, nerf}")
            return None
    enc, s = raw["encoding"], raw["snapshot"]
    if "params_binary" in s:
        n = len(s["params_binary"]) // 2
        n_grid = _grid_entries(int(enc.get("n_levels", 16)), int(enc.get("log2_hashmap_size", 19)), int(enc.get("base_resolution", 16)),
                               float(enc.get("per_level_scale", 0.0)) or 1.0) * int(enc.get("n_features_per_level", 2))
        detail = f"{n} fp16 parameters"
        if n == REF_GRID_PARAMS + REF_MLP_PARAMS:
            rep.ok("snapshot parameters", detail + " = 13,074,912 + 10,240: the reference's network shape (L 16, F 2, T 2^19, 64-wide MLPs)")
        else:
            rep.note("snapshot parameters", detail + f" (the reference's shape has 13,074,912 + 10,240; this encoding config asks for {_MLP_PARAMS} + {n_grid})")
    try:
        snap = load_snapshot_file(str(path))
    except _lib.PxtError as e:
        rep.fail("snapshot import", str(e), "layout of instant-ngp snapshots as ngp.from_instant_ngp states it (params = density MLP | rgb MLP | "
                                            "hash grid, fp16; density grid [cascades][128^3] in Morton order)")
        return None
    except Exception as e:
        rep.fail("snapshot import", repr(e), "ngp.from_instant_ngp's recalled layout (see its comment block)")
        return None
    occ = np.unpackbits(snap.occupancy, bitorder="little")[: snap.cascades * 128**3].reshape(snap.cascades, 128, 128, 128)
    frac0 = float(occ[0].mean())
    rep.ok("snapshot import", f"levels {snap.n_levels}, T 2^{snap.log2_hashmap}, aabb_scale {snap.aabb_scale:g} ({snap.cascades} cascades), "
                              f"scale {snap.scale:g}, k1 {snap.k1:g}, colours {'linear (HDR set)' if snap.linear_colors else 'sRGB -> linear at the end of a ray'}, "
                              f"{100 * frac0:.1f} % of cascade 0 occupied")
    if not (0.0005 < frac0 < 0.9):
        rep.fail("occupancy grid", f"{100 * frac0:.2f} % of cascade 0 is occupied",
                 "density_grid_binary holds densities in Morton order and a cell is occupied above min(0.01, mean) "
                 "(ngp.occupancy_from_density_grid): an almost empty or almost full grid usually means another order or dtype")
    if aabb is None:
        rep.note("OBJ_AABB", "not set in the environment: the CLI needs it (pixloc_tracker_r9.py:85-86)")
    else:
        lo = np.minimum(np.asarray(aabb[0], float), np.asarray(aabb[1], float))
        hi = np.maximum(np.asarray(aabb[0], float), np.asarray(aabb[1], float))
        if (np.asarray(aabb[0], float) > np.asarray(aabb[1], float)).any():
            rep.note("OBJ_AABB", f"{aabb} has min > max on an axis: sorted per axis here, an empty box in instant-ngp (PXT_STRICT_AABB=1 keeps it)")
        zs, ys, xs = np.nonzero(occ[0])  # cascade 0 covers the unit cube, cell centres at (i + 0.5) / 128
        c = (np.stack([xs, ys, zs], 1) + 0.5) / 128.0
        inside = ((c >= lo) & (c <= hi)).all(1)
        kept = int(inside.sum())
        rep.ok("OBJ_AABB", f"{aabb}: keeps {kept} of {len(c)} occupied cells of cascade 0 ({100.0 * kept / max(len(c), 1):.1f} %)")
        if kept == 0:
            rep.fail("OBJ_AABB", "the box holds no occupied cell: every render would be background",
                     "OBJ_AABB is given in instant-ngp's unit-cube coordinates (render_aabb, ingp_utils.py:41-42), the object inside it")
    return snap


def check_checkpoint(rep: Report, path: Path):
    import torch

    from .unet import conv_layer_dims, conv_layer_names, load_weights

    if not path.is_file():
        rep.fail("checkpoint", f"{path} is missing", "$PIXTRACK_WEIGHTS or P/pixtrack/pixloc_megadepth.pt (pixloc's checkpoint_best.tar: "
                                                     "pixloc_pose_refiners.py:49-60 loads experiment pixloc_megadepth)")
        return None
    try:
        blob = torch.load(path, map_location="cpu", weights_only=False)
    except Exception as e:
        rep.fail("checkpoint", f"torch.load failed: {e!r}", "a torch checkpoint: {'model': state_dict, ...} or a flat tensor dict")
        return None
    sd = blob["model"] if isinstance(blob, dict) and isinstance(blob.get("model"), dict) else blob
    try:
        w = load_weights(str(path))
    except KeyError as e:
        have = sorted(k for k in sd if "encoder" in k or "decoder" in k)[:6]
        rep.fail("checkpoint keys", f"tensor {e.args[0]!r} is missing (the file has e.g. {have})",
                 "pixloc's UNet state_dict names: extractor.encoder.<block>.<torchvision VGG16 index>.{weight,bias}, "
                 "extractor.decoder.<d>.layers.{0,1}.*, extractor.adaptation.<k>.0.*, extractor.uncertainty.<k>.0.* (unet.from_pixloc_state_dict, SURVEY A.5)")
        return None
    problems = []
    for name, (cin, cout) in zip(conv_layer_names(), conv_layer_dims()):
        t = w.get(f"{name}.weight")
        if t is None:
            problems.append(f"{name}.weight missing")
        elif tuple(t.shape[:2]) != (cout, cin):
            problems.append(f"{name}.weight is {tuple(t.shape)}, expected ({cout}, {cin}, 3, 3)")
    if problems:
        rep.fail("checkpoint shapes", "; ".join(problems[:4]), "VGG16 encoder up to the 5th pool + decoder [64, 64, 64, 32] + heads 32 / 128 / 128 (SURVEY A.5)")
        return None
    used = {k for k in sd if k.startswith("extractor.") or k.startswith("optimizer.")}
    unexpected = sorted(k for k in sd if k not in used)[:5]
    damp = [k for k in w if k.endswith("dampingnet.const")]
    rep.ok("checkpoint", f"{len(w)} tensors, {len(damp)} damping constant set(s)" + (f"; ignored keys e.g. {unexpected}" if unexpected else ""))
    if len(damp) < 3:
        rep.fail("damping constants", f"found {sorted(damp)}", "one learned optimizer.<level>.dampingnet.const [6] per pyramid level (SURVEY A.3 step 9)")
    return w


def gpu_checks(rep: Report, snap, weights, aabb, cams):
    import torch

    if not torch.cuda.is_available():
        rep.note("GPU checks", "skipped: no ROCm device visible")
        return
    from .ngp import Testbed
    from .unet import UNet
    from .utils.ingp_utils import initialize_ingp

    dev = torch.device("cuda:0")
    tb = initialize_ingp(snap, aabb if aabb is not None else [[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]], device=dev)
    lo, hi = np.asarray(tb.render_aabb.min, float), np.asarray(tb.render_aabb.max, float)
    c, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
    from .synthetic import look_at_pose

    eye = c + np.array([0.9, 0.5, 0.3]) / np.linalg.norm([0.9, 0.5, 0.3]) * ext * 1.2
    R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
    tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
    tb.fov = 45.0
    out = tb.render_frame_device(320, 240, 4, mode=2)
    torch.cuda.synchronize()
    cover = float(out["depth_nz"].float().mean())
    rep.ok("thumbnail render", f"320 x 240 at 1.2 box diagonals: {100 * cover:.1f} % of the pixels hit the object") if cover > 0.002 else \
        rep.fail("thumbnail render", "no pixel hits the object", "the snapshot's density field is non-empty inside $OBJ_AABB (see the occupancy checks above)")
    if weights is not None:
        net = UNet(weights, dev)
        stats = net.activation_stats(out["rgb_u8"])
        worst = max((s[0] for s in stats if s is not None), default=0.0)
        bad = sum(s[1] for s in stats if s is not None)
        if bad or worst > 3.0e4:
            rep.fail("fp16 activations", f"largest |activation| {worst:.0f}, {bad} non-finite value(s) on the thumbnail",
                     "activations stay inside fp16 (max 65504): pixloc runs this network in fp32; unet.auto_rescale_for_fp16(weights, device, images) "
                     "returns exactly rescaled weights")
        else:
            rep.ok("fp16 activations", f"largest |activation| {worst:.0f} on the thumbnail render (fp16 max 65504)")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--object_path", type=Path, required=True)
    ap.add_argument("--query", type=Path, default=None)
    ap.add_argument("--no-gpu", action="store_true")
    args = ap.parse_args(argv)
    rep = Report()
    root = args.object_path / "pixtrack"
    aabb = None
    if os.environ.get("OBJ_AABB"):
        try:
            aabb = ast.literal_eval(os.environ["OBJ_AABB"])
            assert len(aabb) == 2 and len(aabb[0]) == 3 and len(aabb[1]) == 3
        except Exception:
            rep.fail("OBJ_AABB", f"{os.environ['OBJ_AABB']!r} does not parse", "a Python literal [[x0, y0, z0], [x1, y1, z1]] (config/*.sh, pixloc_tracker_r9.py:85)")
            aabb = None
    cams = check_colmap(rep, root / "aug_nerf_sfm" / "aug_sfm", os.environ.get("UPRIGHT_REF_IMG"))
    check_nerf2sfm(rep, root / "pixsfm" / "dataset" / "nerf2sfm.pkl")
    snap = check_snapshot(rep, root / "instant-ngp" / "snapshots" / "weights.msgpack", aabb)
    weights = check_checkpoint(rep, Path(os.environ.get("PIXTRACK_WEIGHTS", str(root / "pixloc_megadepth.pt"))))
    if args.query is not None:
        from .utils.io import ImageIterator

        try:
            it = ImageIterator(str(args.query), 2)
            first = next(iter(it))
            img = first[1]
            rep.ok("query frames", f"{args.query}: first frame {tuple(img.shape)}")
        except Exception as e:
            rep.fail("query frames", repr(e), "a directory of images readable by utils.io.ImageIterator (io.py, sorted by name)")
    if not args.no_gpu and snap is not None:
        try:
            gpu_checks(rep, snap, weights, aabb, cams)
        except Exception as e:
            rep.fail("GPU checks", repr(e), "the HIP library loads and a 320 x 240 render + one UNet pass run on cuda:0")
    rep.dump()
    return 1 if rep.failed else 0


if __name__ == "__main__":
    sys.exit(main())
