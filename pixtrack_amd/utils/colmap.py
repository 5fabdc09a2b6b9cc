"""COLMAP sparse-model records + binary reader (what pixloc's Model3D wraps).

The reference vendors COLMAP's own reader (pixtrack/utils/colmap_read_model.py); this is
an independent, minimal restatement of the documented binary format (cameras.bin,
images.bin, points3D.bin) with the attribute names pixtrack touches:
``image.qvec2rotmat() / .tvec / .name / .camera_id / .point3D_ids``,
``point.xyz / .image_ids``, ``camera.model / .width / .height / .params``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict

import numpy as np

# model_id -> (name, n_params)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5),
                 4: ("OPENCV", 8), 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5),
                 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
MODEL_IDS = {v[0]: k for k, v in CAMERA_MODELS.items()}


def qvec2rotmat(q) -> np.ndarray:
    w, x, y, z = q
    return np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
        [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
        [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def rotmat2qvec(R) -> np.ndarray:
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R).flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0],
                  [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0], [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


@dataclass
class ColmapCamera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray

    def _asdict(self):
        return dict(id=self.id, model=self.model, width=self.width, height=self.height, params=self.params)


@dataclass
class ColmapImage:
    id: int
    qvec: np.ndarray
    tvec: np.ndarray
    camera_id: int
    name: str
    xys: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    point3D_ids: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int64))

    def qvec2rotmat(self) -> np.ndarray:
        return qvec2rotmat(self.qvec)


@dataclass
class ColmapPoint3D:
    id: int
    xyz: np.ndarray
    rgb: np.ndarray
    error: float
    image_ids: np.ndarray
    point2D_idxs: np.ndarray


def _read(f, fmt):
    return struct.unpack("<" + fmt, f.read(struct.calcsize("<" + fmt)))


def read_cameras_binary(path) -> Dict[int, ColmapCamera]:
    cams = {}
    with open(path, "rb") as f:
        (n,) = _read(f, "Q")
        for _ in range(n):
            cid, mid, w, h = _read(f, "iiQQ")
            name, npar = CAMERA_MODELS[mid]
            cams[cid] = ColmapCamera(cid, name, int(w), int(h), np.array(_read(f, "d" * npar)))
    return cams


def read_images_binary(path) -> Dict[int, ColmapImage]:
    imgs = {}
    with open(path, "rb") as f:
        (n,) = _read(f, "Q")
        for _ in range(n):
            vals = _read(f, "idddddddi")
            iid, q, t, cid = vals[0], np.array(vals[1:5]), np.array(vals[5:8]), vals[8]
            name = b""
            while True:
                c = f.read(1)
                if c == b"\x00":
                    break
                name += c
            (m,) = _read(f, "Q")
            raw = np.frombuffer(f.read(24 * m), dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]))
            imgs[iid] = ColmapImage(iid, q, t, cid, name.decode(), np.stack([raw["x"], raw["y"]], 1) if m else np.zeros((0, 2)),
                                    raw["id"].astype(np.int64))
    return imgs


def read_points3D_binary(path) -> Dict[int, ColmapPoint3D]:
    pts = {}
    with open(path, "rb") as f:
        (n,) = _read(f, "Q")
        for _ in range(n):
            vals = _read(f, "QdddBBBd")
            (tl,) = _read(f, "Q")
            tr = np.frombuffer(f.read(8 * tl), dtype="<i4").reshape(-1, 2)
            pts[vals[0]] = ColmapPoint3D(vals[0], np.array(vals[1:4]), np.array(vals[4:7]), vals[7],
                                         tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64))
    return pts


def write_model_binary(path, cameras, images, points3D) -> None:
    path = Path(path)
    path.mkdir(parents=True, exist_ok=True)
    with open(path / "cameras.bin", "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras.values():
            f.write(struct.pack("<iiQQ", c.id, MODEL_IDS[c.model], c.width, c.height))
            f.write(struct.pack("<" + "d" * len(c.params), *c.params))
    with open(path / "images.bin", "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images.values():
            f.write(struct.pack("<idddddddi", im.id, *im.qvec, *im.tvec, im.camera_id))
            f.write(im.name.encode() + b"\x00")
            f.write(struct.pack("<Q", len(im.point3D_ids)))
            for (x, y), pid in zip(im.xys, im.point3D_ids):
                f.write(struct.pack("<ddq", x, y, int(pid)))
    with open(path / "points3D.bin", "wb") as f:
        f.write(struct.pack("<Q", len(points3D)))
        for p in points3D.values():
            f.write(struct.pack("<QdddBBBd", p.id, *p.xyz, *[int(v) for v in p.rgb], p.error))
            f.write(struct.pack("<Q", len(p.image_ids)))
            for a, b in zip(p.image_ids, p.point2D_idxs):
                f.write(struct.pack("<ii", int(a), int(b)))


def read_model(path):
    path = Path(path)
    return (read_cameras_binary(path / "cameras.bin"), read_images_binary(path / "images.bin"),
            read_points3D_binary(path / "points3D.bin"))
