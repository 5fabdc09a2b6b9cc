"""PixTrackFeatureExtractor (reference pixtrack/localization/feature_extractor.py:9-59).

Same name, constructor and ``__call__(image, scale_image) -> (features, scales, confidences)``
contract.  ``extract_packed`` is the device-resident path the refiner uses: the image stays on
the GPU (float32 / uint8 HWC 0..255), the optional query mask is multiplied inside the first
convolution, and the pyramid comes back as the HWC records the LM kernel reads.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .ops import ops
from .unet import OUTPUT_DIMS
from .utils.conf import merge


class PixTrackFeatureExtractor:
    default_conf: Dict = dict(resize=1024, resize_by="max")

    def __init__(self, model, device: torch.device, conf: Optional[Dict] = None):
        self.conf = merge(self.default_conf, conf or {})
        self.device = torch.device(device)
        self.model = model
        self._staged = None   # (image, scale_image, mask, normalize) announced by stage()
        self._ready = None    # (image, scale_image, mask, normalize, maps, scales) computed alongside
        # The frame's two passes run side by side; the announced (query) pass may still be running when the reference
        # maps are handed back - the refiner samples the reference points from them meanwhile - and is joined when ITS
        # maps are asked for (round 4; PXT_UNET_DEFER_JOIN=0: join inside the call, as before).
        import os

        self.defer_join = os.environ.get("PXT_UNET_DEFER_JOIN", "1") != "0" and hasattr(model, "set_defer_join")
        self._join_pending = False
        self._preloaded = []  # [(image, scale_image, mask, normalize, maps, scales)] handed over by preload()
        self.last_input_wh = None  # (w, h) the UNet ran the last extract_packed image at (after the resize rule)
        assert hasattr(self.model, "scales")
        assert self.conf.resize_by in ["max", "max_force"], self.conf.resize_by

    def eval(self):
        return self

    # ---- sizing rule of pixloc `resize(image, size, max, "linear")` -----------
    def target_size(self, h: int, w: int, scale_image: int) -> Tuple[int, int, Tuple[float, float]]:
        scale_resize = (1.0, 1.0)
        if self.conf.resize is not None:
            target = self.conf.resize // scale_image
            if max(h, w) > target or self.conf.resize_by == "max_force":
                s = target / max(h, w)
                h_new, w_new = int(round(h * s)), int(round(w * s))
                # pixloc's resize() returns the UNROUNDED factor for the int/max branch (its own
                # TODO says it should recompute from the rounded size; it does not), and the level
                # cameras are scaled with what it returns (feature_extractor.py:45,56-57)
                return h_new, w_new, (s, s)
        return h, w, scale_resize

    def _to_device_hwc(self, image) -> torch.Tensor:
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image))
        if image.dtype not in (torch.float32, torch.uint8):
            image = image.float()
        # frames handed over in pinned host memory (ImageIterator's role) are uploaded asynchronously
        return image.to(self.device, non_blocking=True).contiguous()

    def stage(self, image, scale_image: int = 1, mask: Optional[torch.Tensor] = None,
              normalize: bool = False) -> None:
        """Announces an extraction that WILL be requested next with exactly these arguments (the
        frame's masked query, known before the reference render is encoded).  The next
        extract_packed then runs both images through the UNet side by side
        (pxt_unet_forward_batch, or pxt_unet_forward_pair when their sizes differ) and keeps the staged result for the announced call.  Purely
        a scheduling hint: results are those of separate calls up to fp32 summation order."""
        self._join()
        self._staged = (image, scale_image, mask, normalize)
        self._ready = None

    def preload(self, image, scale_image: int, mask, normalize: bool, maps, scales) -> None:
        """Hands over the pyramid of an extraction that WILL be requested with exactly these arguments - computed
        elsewhere, in a batched pass over the images of several trackers (pose_trackers/multi_object_tracker.py: the
        reference renders and masked queries of K objects in one pxt_unet_forward_batch).  The matching extract_packed
        call returns it; anything else is computed as usual.  Cleared by unstage()."""
        self._preloaded.append((image, scale_image, mask, normalize, maps, scales))

    def prepared(self, image, scale_image: int = 1, mask=None):
        """(HWC device image at the size the UNet runs it, mask still to apply, per-level scales): what a batched pass
        needs to run this extraction elsewhere (see preload)."""
        a_img, a_mask, a_sr = self._prepare(image, scale_image, mask)
        return a_img, a_mask, [(a_sr[0] / s, a_sr[1] / s) for s in self.model.scales]

    def unstage(self) -> None:
        self._join()
        self._staged = self._ready = None
        self._preloaded.clear()

    def _join(self) -> None:
        if self._join_pending:
            self.model.join()
            self._join_pending = False

    def _prepare(self, image, scale_image, mask):
        """-> (HWC image on the device at the size the UNet runs it, mask still to apply, scale_resize).  An image above
        the resize target is masked first and resized here (the reference multiplies the mask into the image before its
        `resize`, pixloc_pose_refiners.py:240-249 + feature_extractor.py:41-45)."""
        img = self._to_device_hwc(image)
        H, W = int(img.shape[0]), int(img.shape[1])
        h_new, w_new, scale_resize = self.target_size(H, W, scale_image)
        if (h_new, w_new) != (H, W):
            # where the resized image can be non-zero (the UNet's constant-tile skipping needs to know: a resized image is
            # float and carries no mask any more): from the mask it is multiplied by, or from a uint8 render's zeros.
            # Handed on as the UNet call's mask - a multiplication by exactly 0 or 1 there.
            active = None
            if mask is not None:
                mask = mask.to(self.device, torch.uint8).contiguous()
                active = torch.empty(h_new, w_new, dtype=torch.uint8, device=self.device)
                ops.resize_activity(mask, None, H, W, active)
                img = img.float() * mask[..., None].float()
            elif img.dtype == torch.uint8:
                active = torch.empty(h_new, w_new, dtype=torch.uint8, device=self.device)
                ops.resize_activity(None, img, H, W, active)
            mask = active
            src = img.float().contiguous()
            dst = torch.empty(h_new, w_new, 3, device=self.device, dtype=torch.float32)
            ops.resize_linear(src, dst)
            img = dst
        return img, mask, scale_resize

    def extract_packed(self, image, scale_image: int = 1, mask: Optional[torch.Tensor] = None,
                       normalize: bool = False):
        """-> (maps [h,w,cstride] x3 on device, scales [(sx,sy)] x3)."""
        for k, r in enumerate(self._preloaded):
            if r[0] is image and r[1] == scale_image and r[2] is mask and r[3] == normalize:
                del self._preloaded[k]
                self.last_input_wh = (int(r[4][0].shape[1]), int(r[4][0].shape[0]))
                return r[4], r[5]
        if self._ready is not None:
            r = self._ready
            self._ready = None
            self._join()  # (the announced pass may have been left running beside the caller's work)
            if r[0] is image and r[1] == scale_image and r[2] is mask and r[3] == normalize:
                return r[4], r[5]
        a_img, a_mask, a_sr = self._prepare(image, scale_image, mask)
        self.last_input_wh = (int(a_img.shape[1]), int(a_img.shape[0]))
        scales = [(a_sr[0] / s, a_sr[1] / s) for s in self.model.scales]
        if self._staged is not None:
            st_image, st_scale, st_mask, st_norm = self._staged
            self._staged = None
            if st_image is not image:
                # equal sizes: one batched call; two sizes: the pair entry - both images side by side either way
                b_img, b_mask, b_sr = self._prepare(st_image, st_scale, st_mask)
                if self.defer_join:
                    self.model.set_defer_join(True)
                try:
                    both = self.model.forward_packed_batch([(a_img, a_mask, normalize), (b_img, b_mask, st_norm)])
                finally:
                    if self.defer_join:
                        self.model.set_defer_join(False)
                        self._join_pending = True
                # (b_img / b_mask ride along: the side stream may still be reading them until the join)
                self._ready = (st_image, st_scale, st_mask, st_norm, both[1],
                               [(b_sr[0] / s, b_sr[1] / s) for s in self.model.scales], b_img, b_mask)
                return both[0], scales
        return self.model.forward_packed(a_img, a_mask, normalize=normalize), scales

    @torch.no_grad()
    def __call__(self, image: np.ndarray, scale_image: int = 1):
        """Reference contract: features [C,h,w] x3, scales, confidences [1,h,w] x3 (device)."""
        maps, scales = self.extract_packed(image, scale_image, None, normalize=False)
        features = [m[..., :c].permute(2, 0, 1) for m, c in zip(maps, OUTPUT_DIMS)]
        confidences = [m[..., c : c + 1].permute(2, 0, 1) for m, c in zip(maps, OUTPUT_DIMS)]
        assert len(self.model.scales) == len(features)
        return features, scales, confidences
