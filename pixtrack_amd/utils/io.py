"""Frame iterators (reference pixtrack/utils/io.py:13-128)."""
from __future__ import annotations

import glob
import os

import numpy as np


def read_image(path) -> np.ndarray:
    """RGB uint8 H x W x 3 (pixloc read_image reads BGR with cv2 and flips to RGB)."""
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB"))


class ImagePathIterator:
    def __init__(self, path, max_frames=10):
        assert os.path.isdir(path)
        paths = sorted(glob.glob(os.path.join(path, "*.jpg")) + glob.glob(os.path.join(path, "*.png")))
        self.image_paths = paths[:max_frames]
        self.idx = 0

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.image_paths)

    def __next__(self):
        if self.idx > len(self) - 1:
            raise StopIteration
        self.idx += 1
        return self.image_paths[self.idx - 1]


class ImageIterator:
    """Preloads every frame as float32 RGB, yields (path, image)."""

    def __init__(self, path, max_frames=100):
        assert os.path.isdir(path)
        paths = sorted(glob.glob(os.path.join(path, "*.jpg")) + glob.glob(os.path.join(path, "*.png")))
        if max_frames is not None and np.isfinite(max_frames):
            paths = paths[: int(max_frames)]
        self.image_paths = paths
        print("Reading query images")
        print(len(paths))
        self.images = [read_image(p).astype(np.float32) for p in paths]
        self.idx = 0

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.image_paths)

    def __next__(self):
        if self.idx > len(self) - 1:
            raise StopIteration
        self.idx += 1
        return self.image_paths[self.idx - 1], self.images[self.idx - 1]


class ArrayIterator:
    """In-memory (name, image) frames; images may be numpy or device tensors (HWC, 0..255)."""

    def __init__(self, names, images, max_frames=None):
        n = len(names) if max_frames is None or not np.isfinite(max_frames) else min(len(names), int(max_frames))
        self.image_paths, self.images = list(names[:n]), list(images[:n])
        self.idx = 0

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.image_paths)

    def __next__(self):
        if self.idx > len(self) - 1:
            raise StopIteration
        self.idx += 1
        return self.image_paths[self.idx - 1], self.images[self.idx - 1]


# ---------------------------------------------------------------------------------------------
# poses.pkl / trackers.pkl interchange with the reference
#
# The reference pickles pixloc objects (reference pixloc_tracker_r9.py:281-284,314-316) and its
# tools unpickle them again (reference run_vis_on_poses.py:289-304, notebooks/GetMetrics.ipynb),
# so the class path inside the pickle is ``pixloc.pixlib.geometry.wrappers.{Pose,Camera}`` with
# the single instance attribute ``_data``.  Our Pose/Camera keep that attribute, so the two
# directions are a module-path rename at (un)pickling time.
_PIXLOC_WRAPPERS = "pixloc.pixlib.geometry.wrappers"
_PIXLOC_ALIASES = ("pixloc.pixlib.geometry.wrappers", "pixloc.pixlib.geometry")


def load_reference_pickle(path_or_file):
    """Reads a poses.pkl / trackers.pkl written by the reference *or* by this package; pixloc
    Pose/Camera instances come back as pixtrack_amd.geometry types."""
    import pickle

    from ..geometry import Camera, Pose

    table = {"Pose": Pose, "Camera": Camera}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module in _PIXLOC_ALIASES and name in table:
                return table[name]
            return super().find_class(module, name)

    if hasattr(path_or_file, "read"):
        return _Unpickler(path_or_file).load()
    with open(path_or_file, "rb") as f:
        return _Unpickler(f).load()


def dump_reference_pickle(obj, path_or_file, to_cpu: bool = True) -> None:
    """Writes ``obj`` so that a process with pixloc installed (and without this package)
    unpickles Pose/Camera as pixloc's own classes.  ``to_cpu`` moves their tensors to the host
    first, so the file loads on a machine without a ROCm device."""
    import copyreg
    import pickle
    import sys
    import types

    from ..geometry import Camera, Pose

    # Stand-in classes that pickle *by reference* under pixloc's module path.  They exist in
    # sys.modules only while dumping, and only if pixloc itself is not importable.
    created = []
    try:
        import importlib

        wrappers = importlib.import_module(_PIXLOC_WRAPPERS)
        targets = {Pose: wrappers.Pose, Camera: wrappers.Camera}
    except Exception:
        parts = _PIXLOC_WRAPPERS.split(".")
        for i in range(1, len(parts) + 1):
            name = ".".join(parts[:i])
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                created.append(name)
        mod = sys.modules[_PIXLOC_WRAPPERS]
        targets = {}
        for cls in (Pose, Camera):
            stub = type(cls.__name__, (), {})
            stub.__module__, stub.__qualname__ = _PIXLOC_WRAPPERS, cls.__name__
            setattr(mod, cls.__name__, stub)
            targets[cls] = stub

    class _Pickler(pickle.Pickler):
        def reducer_override(self, o):
            tgt = targets.get(type(o))
            if tgt is None:
                return NotImplemented
            data = o._data.detach()
            # copyreg._reconstructor(cls, object, None) == object.__new__(cls); state -> __dict__
            return copyreg._reconstructor, (tgt, object, None), {"_data": data.cpu() if to_cpu else data}

    try:
        if hasattr(path_or_file, "write"):
            _Pickler(path_or_file, protocol=pickle.HIGHEST_PROTOCOL).dump(obj)
        else:
            with open(path_or_file, "wb") as f:
                _Pickler(f, protocol=pickle.HIGHEST_PROTOCOL).dump(obj)
    finally:
        for name in created:
            sys.modules.pop(name, None)


# ---------------------------------------------------------------------------------------------
# YCB-Video (reference pixtrack/utils/io.py:13-72).  The reference goes through the `ycbvideo`
# package; here the dataset's files are read directly: <root>/data/<seq:04d>/<frame:06d>-color.png,
# -label.png and -meta.mat (scipy.io.loadmat: intrinsic_matrix 3x3, cls_indexes [n,1], poses [3,4,n]).
# ---------------------------------------------------------------------------------------------
YCB_CLASS_MAP = {"003_cracker_box": 2, "004_sugar_box": 3, "006_mustard_bottle": 5, "021_bleach_cleanser": 12,
                 "035_power_drill": 15}  # io.py:21-27


def _parse_slice(spec: str, n: int):
    if spec in ("", "*"):
        return list(range(n))
    if ":" not in spec:
        return [int(spec)]
    parts = [int(p) if p else None for p in spec.split(":")]
    return list(range(n))[slice(*parts)]


def parse_ycb_expression(expression: str, ycb_root) -> list:
    """'<sequences>/<frames>' frame selection of ycbvideo.Loader.frames, the subset pixtrack uses:
    '7/:20' (sequence 7, first 20 frames), '7' or '7/*' (all of it), '7:10' (sequences 7, 8, 9).
    Frame positions index the sequence's sorted frame list.  Returns [(sequence dir, frame stem)]."""
    from pathlib import Path

    seq_spec, _, frame_spec = str(expression).partition("/")
    root = Path(ycb_root) / "data"
    seqs = sorted(p.name for p in root.iterdir() if p.is_dir()) if root.is_dir() else []
    if ":" in seq_spec:
        lo, _, hi = seq_spec.partition(":")
        wanted = [f"{i:04d}" for i in range(int(lo or 0), int(hi) if hi else (int(seqs[-1]) + 1 if seqs else 0))]
    else:
        wanted = [f"{int(seq_spec):04d}"]
    out = []
    for sq in wanted:
        d = root / sq
        if not d.is_dir():
            raise FileNotFoundError(f"YCB-Video sequence {d} not found")
        stems = sorted(p.name[: -len("-color.png")] for p in d.glob("*-color.png"))
        out += [(sq, stems[i]) for i in _parse_slice(frame_spec, len(stems))]
    return out


class YCBVideoIterator:
    """Yields (path, image float32 HxWx3, gt_pose: Pose, gt_camera: Camera) per frame, as the
    reference's iterator does: OPENCV camera from the frame's intrinsic matrix with the principal
    point FORCED to (319.5, 239.5) (io.py:50) and zero distortion, pose = the object's [R|t] of
    `meta["poses"]` at the index of its class id."""

    def __init__(self, object_path, expression="7/:20", ycb_path="/data/ycb/"):
        from pathlib import Path

        self.ycb_root = Path(ycb_path)
        self.object_id = YCB_CLASS_MAP[Path(object_path).name]
        self.frames = parse_ycb_expression(expression, self.ycb_root)
        self.idx = 0

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.frames)

    def __next__(self):
        from scipy.io import loadmat

        from ..geometry import Camera, Pose
        from .colmap import ColmapCamera

        if self.idx > len(self) - 1:
            raise StopIteration
        sequence, frame = self.frames[self.idx]
        base = self.ycb_root / "data" / sequence
        path = base / f"{frame}-color.png"
        query_image = read_image(path).astype(np.float32)
        meta = loadmat(str(base / f"{frame}-meta.mat"))
        K = np.asarray(meta["intrinsic_matrix"], np.float64)
        fx, fy = K[0, 0], K[1, 1]
        cx, cy = 319.5, 239.5  # the dataset's principal point is overridden (io.py:46-50)
        H, W = query_image.shape[:2]
        cls = np.asarray(meta["cls_indexes"]).reshape(-1)
        pose_idx = int(np.argwhere(cls == self.object_id).squeeze())
        pose = np.asarray(meta["poses"], np.float64)[:, :, pose_idx]
        pixpose = Pose.from_Rt(pose[:, :3], pose[:, 3])
        camera = ColmapCamera(1, "OPENCV", W, H, np.array([fx, fy, cx, cy]))
        self.idx += 1
        return path, query_image, pixpose, Camera.from_colmap(camera)


def write_ycb_sequence(ycb_root, sequence: int, frames, gt_poses, K: np.ndarray, class_id: int, first_frame: int = 1):
    """Writes frames (HWC 0..255) + per-frame meta in the YCB-Video layout (synthetic test data):
    <root>/data/<seq:04d>/<frame:06d>-{color.png, label.png, meta.mat}."""
    from pathlib import Path

    from PIL import Image
    from scipy.io import savemat

    d = Path(ycb_root) / "data" / f"{int(sequence):04d}"
    d.mkdir(parents=True, exist_ok=True)
    for i, (fr, (R, t)) in enumerate(zip(frames, gt_poses)):
        a = fr.detach().cpu().numpy() if hasattr(fr, "detach") else np.asarray(fr)
        stem = f"{first_frame + i:06d}"
        Image.fromarray(np.clip(np.rint(a), 0, 255).astype(np.uint8)).save(d / f"{stem}-color.png")
        Image.fromarray(np.full(a.shape[:2], class_id, np.uint8)).save(d / f"{stem}-label.png")
        # a second, unrelated object comes first: the reader must pick the pose by class id
        poses = np.stack([np.eye(3, 4), np.concatenate([np.asarray(R), np.asarray(t)[:, None]], 1)], -1)
        savemat(str(d / f"{stem}-meta.mat"), {"intrinsic_matrix": np.asarray(K, np.float64),
                                               "cls_indexes": np.array([[1], [class_id]], np.uint8), "poses": poses,
                                               "factor_depth": np.array([[10000]], np.uint16)})
