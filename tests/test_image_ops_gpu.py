"""Small image kernels vs numpy restatements of the cv2/numpy calls they replace."""
import numpy as np
import pytest
import torch

from oracle import unet_oracle as UO
from pixtrack_amd import _lib

pytestmark = pytest.mark.gpu


def np_morph(img, erode):
    H, W = img.shape
    pad = np.full((H + 4, W + 4), 255 if erode else 0, dtype=np.uint8)
    pad[2:-2, 2:-2] = img
    stack = [pad[dy:dy + H, dx:dx + W] for dy in range(5) for dx in range(5)]
    return (np.min if erode else np.max)(np.stack(stack, 0), axis=0)


@pytest.mark.parametrize("H,W,ne,nd,y0,x0", [(60, 84, 1, 5, 15, 20), (60, 84, 1, 5, 0, 44), (97, 150, 2, 3, 67, 0),
                                              (60, 84, 2, 7, 15, 20), (40, 70, 0, 2, 10, 30)])
def test_depth_mask_matches_cv2_semantics(device, H, W, ne, nd, y0, x0):
    """(1, 5) is the tracker's setting (single fused pass); (2, 7) exceeds the fused kernel's halo
    and takes the iterated kernels; blobs touching the border exercise OpenCV's border rule."""
    rng = np.random.default_rng(1)
    depth = np.zeros((H, W, 4), np.float32)
    blob = np.zeros((H, W), np.float32)
    blob[y0:y0 + 30, x0:x0 + 40] = rng.uniform(0.5, 2.0, size=(30, 40))
    blob[rng.uniform(size=(H, W)) > 0.97] = 1.0   # speckles the erosion must remove
    blob[30, 40] = 0.0                            # pin-hole the erosion must widen
    blob[5, 5] = 0.003                            # < 1/255 -> uint8 0
    blob[6, 6] = 256.0 / 255.0                    # wraps to 0 in uint8 (Appendix D.6)
    depth[..., :3] = blob[..., None]
    ref = ((depth[..., 0] * 255.0).astype(np.int64) & 255) != 0
    ref = ref.astype(np.uint8)
    for _ in range(ne):
        ref = np_morph(ref, True)
    for _ in range(nd):
        ref = np_morph(ref, False)
    d = torch.from_numpy(depth).to(device)
    out = torch.zeros(H, W, dtype=torch.uint8, device=device)
    tmp = torch.zeros(2 * H * W, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().pxt_depth_mask(d.data_ptr(), H, W, ne, nd, out.data_ptr(), tmp.data_ptr(),
                                         _lib.stream_ptr(device)), "pxt_depth_mask")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
    assert 0 < ref.sum() < H * W


def test_rgba_to_u8(device):
    rng = np.random.default_rng(2)
    rgba = rng.uniform(0, 1, size=(33, 47, 4)).astype(np.float32)
    ref = rgba.copy()
    ref[ref[:, :, 3] < 0.25] = 0.0
    ref = (ref[:, :, :3] * 255.0).astype(np.uint8)
    d = torch.from_numpy(rgba).to(device)
    out = torch.zeros(33, 47, 3, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().pxt_rgba_to_u8(d.data_ptr(), 33, 47, 0.25, out.data_ptr(), _lib.stream_ptr(device)), "u8")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.parametrize("H,W,Ho,Wo", [(480, 640, 192, 256), (100, 75, 33, 25), (48, 64, 96, 128)])
def test_resize_linear(device, H, W, Ho, Wo):
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 255, size=(H, W, 3)).astype(np.float32)
    ref = UO.cv2_resize_linear(img, Wo, Ho)
    d = torch.from_numpy(img).to(device)
    out = torch.zeros(Ho, Wo, 3, device=device)
    _lib.check(_lib.lib().pxt_resize_linear(d.data_ptr(), H, W, 3, out.data_ptr(), Ho, Wo, _lib.stream_ptr(device)), "rs")
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-3
