"""Frame iterators (reference pixtrack/utils/io.py:75-128)."""
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
