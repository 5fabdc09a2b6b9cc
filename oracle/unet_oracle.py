"""CPU ORACLE (test infrastructure, NOT product code) for the UNet feature pyramid.

PARITY UNPINNED: the network is pixloc's ``UNet`` with the ``pixloc_megadepth``
configuration (cvg/pixloc, un-vendored and un-pinned: /root/reference/.gitmodules:4-6);
neither its source nor its checkpoint exists under /root/reference.
This restates the published architecture in plain
PyTorch-CPU fp32 with ``torch.nn.functional`` ops, anchored on the reference call
sites pixtrack/localization/feature_extractor.py:34-59 (prepare_input, model call,
scales) and pixtrack/localization/pixloc_pose_refiners.py:49-60 (experiment load).

Architecture: VGG16 ``features`` up to (excluding) the 5th max-pool, split into 5 blocks
at each pool (the pool OPENS the next block); decoder blocks = bilinear x2 upsample
(align_corners=False), crop skip, concat [upsampled, skip], conv3x3(pad 1, no bias) +
BatchNorm(eval) + ReLU with outputs [64, 64, 64, 32]; 1x1 adaptation heads at output
scales [0, 2, 4] -> dims [32, 128, 128]; 1x1 uncertainty heads with
confidence = sigmoid(-x); ImageNet mean/std normalisation inside the model.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
ENC_BLOCKS = [[(3, 64), (64, 64)], [(64, 128), (128, 128)], [(128, 256), (256, 256), (256, 256)],
              [(256, 512), (512, 512), (512, 512)], [(512, 512), (512, 512), (512, 512)]]
SKIP_DIMS = [64, 128, 256, 512, 512]
DECODER = [64, 64, 64, 32]
OUTPUT_SCALES = [0, 2, 4]
OUTPUT_DIMS = [32, 128, 128]
BN_EPS = 1e-5


def head_inputs() -> List[int]:
    """Channel count feeding each adaptation/uncertainty head."""
    n_enc = len(ENC_BLOCKS)
    return [SKIP_DIMS[i] if i == n_enc - 1 else DECODER[-1 - i] for i in OUTPUT_SCALES]


def unet_forward(weights: Dict[str, torch.Tensor], image_chw01: torch.Tensor) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """image: [3,H,W] float in [0,1] (numpy_image_to_torch output).  Returns
    (feature_maps [C_l,h,w] x3, confidences [1,h,w] x3), fine -> coarse."""
    w = weights
    x = image_chw01[None].float()
    mean = x.new_tensor(MEAN)[None, :, None, None]
    std = x.new_tensor(STD)[None, :, None, None]
    x = (x - mean) / std
    skips = []
    for b, convs in enumerate(ENC_BLOCKS):
        if b > 0:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
        for i, _ in enumerate(convs):
            x = F.relu(F.conv2d(x, w[f"enc{b}_{i}.weight"], w[f"enc{b}_{i}.bias"], padding=1))
        skips.append(x)
    pre = [skips[-1]]
    for d, skip in enumerate(skips[:-1][::-1]):
        up = F.interpolate(pre[-1], scale_factor=2, mode="bilinear", align_corners=False)
        hu, wu = up.shape[-2:]
        assert hu <= skip.shape[-2] and wu <= skip.shape[-1]
        y = torch.cat([up, skip[:, :, :hu, :wu]], 1)
        y = F.conv2d(y, w[f"dec{d}.weight"], None, padding=1)
        y = F.batch_norm(y, w[f"dec{d}.bn_mean"], w[f"dec{d}.bn_var"], w[f"dec{d}.bn_weight"],
                         w[f"dec{d}.bn_bias"], training=False, eps=BN_EPS)
        pre.append(F.relu(y))
    pre = pre[::-1]  # fine -> coarse: dec3, dec2, dec1, dec0, enc4
    feats, confs = [], []
    for k, i in enumerate(OUTPUT_SCALES):
        feats.append(F.conv2d(pre[i], w[f"adapt{k}.weight"], w[f"adapt{k}.bias"])[0])
        unc = F.conv2d(pre[i], w[f"unc{k}.weight"], w[f"unc{k}.bias"])[0]
        confs.append(torch.sigmoid(-unc))
    return feats, confs


def extractor_call(weights, image_hwc: np.ndarray, scale_image: int = 1, resize_max: int = 1024):
    """PixTrackFeatureExtractor.__call__ (feature_extractor.py:34-59): optional resize so
    that max side == resize_max // scale_image (bilinear, cv2.INTER_LINEAR convention),
    HWC->CHW, /255, forward; scales = resize_scale / model.scales."""
    img = np.asarray(image_hwc, dtype=np.float32)
    scale_resize = (1.0, 1.0)
    target = resize_max // scale_image
    if max(img.shape[:2]) > target:
        h, w = img.shape[:2]
        s = target / max(h, w)
        h_new, w_new = int(round(h * s)), int(round(w * s))
        img = cv2_resize_linear(img, w_new, h_new)
        scale_resize = (s, s)  # pixloc resize(): the int/max branch returns the unrounded factor (its TODO)
    x = torch.from_numpy(img).permute(2, 0, 1) / 255.0
    feats, confs = unet_forward(weights, x)
    scales = [(scale_resize[0] / s, scale_resize[1] / s) for s in (1, 4, 16)]
    return feats, scales, confs


def cv2_resize_linear(img: np.ndarray, w_new: int, h_new: int) -> np.ndarray:
    """cv2.resize(img, (w_new, h_new), interpolation=cv2.INTER_LINEAR) for float32 HWC
    (half-pixel centres, edge clamp, no antialias)."""
    h, w = img.shape[:2]
    sx, sy = w / w_new, h / h_new
    fx = (np.arange(w_new, dtype=np.float32) + 0.5) * np.float32(sx) - 0.5
    fy = (np.arange(h_new, dtype=np.float32) + 0.5) * np.float32(sy) - 0.5
    x0 = np.floor(fx).astype(np.int64)
    y0 = np.floor(fy).astype(np.int64)
    ax = (fx - x0).astype(np.float32)
    ay = (fy - y0).astype(np.float32)
    ax[x0 < 0] = 0
    x0[x0 < 0] = 0
    ax[x0 >= w - 1] = 0
    x0[x0 >= w - 1] = w - 1
    ay[y0 < 0] = 0
    y0[y0 < 0] = 0
    ay[y0 >= h - 1] = 0
    y0[y0 >= h - 1] = h - 1
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    img = img.astype(np.float32)
    ax_ = ax[None, :, None]
    ay_ = ay[:, None, None]
    top = img[y0][:, x0] * (1 - ax_) + img[y0][:, x1] * ax_
    bot = img[y1][:, x0] * (1 - ax_) + img[y1][:, x1] * ax_
    return (top * (1 - ay_) + bot * ay_).astype(np.float32)
