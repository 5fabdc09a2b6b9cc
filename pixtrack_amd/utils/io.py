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
