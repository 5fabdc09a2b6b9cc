"""The reference pass on a WINDOW of the reference render (round 5).  The reference computes the reference image's dense
maps only to sample them at the projected 3-D points (pixtrack/localization/pixloc_pose_refiners.py:282-290, `del
features_ref_dense` :236,316); with the reference's own camera shapes the render is 921 x 921 and mostly background
(scripts/create_sfm_from_obj.py:154-159).  Three facts carry the window: (1) the sampler addresses a window of a level
exactly as it addresses the full level; (2) a texel of the pyramid depends on the input only within the dependency
radius (scripts/unet_dependency_radius.py: 135 px at stride 1), so the maps inside the points' bounding box do not
change - bit for bit - when everything farther away than PoseTrackerRefiner.WINDOW_MARGIN changes; (3) end to end, the
sparse reference features of a windowed pass equal the full pass's up to the layers' fp32 summation order (another image
size takes another tile / split-K plan)."""
import numpy as np
import pytest
import torch

from pixtrack_amd.ops import ops
from pixtrack_amd.refiner import PoseTrackerRefiner
from pixtrack_amd.unet import OUTPUT_DIMS, UNet, make_synthetic_unet_weights

pytestmark = pytest.mark.gpu


def test_sampler_reads_a_window_as_the_full_level(device):
    g = torch.Generator().manual_seed(5)
    H, W, x0, y0, ww, wh = 240, 336, 64, 32, 208, 160  # window in stride-1 pixels, multiples of 16
    n = 1500
    p3d = torch.cat([torch.rand(n, 1, generator=g) * (W + 40) - 20, torch.rand(n, 1, generator=g) * (H + 40) - 20,
                     torch.ones(n, 1)], 1).to(device)
    fulls, wins, cams, chans, windows = [], [], [], [], []
    for c, s in zip(OUTPUT_DIMS, (1, 4, 16)):
        cs = (c + 1 + 3) // 4 * 4
        fm = torch.randn(H // s, W // s, cs, generator=g).to(device)
        fulls.append(fm)
        wins.append(fm[y0 // s:(y0 + wh) // s, x0 // s:(x0 + ww) // s].contiguous())
        # identity pose + a camera that maps (x, y, 1) to pixel (x, y) / s of the level
        cams += [float(W // s), float(H // s), 1.0 / s, 1.0 / s, 0.0, 0.0, 0, 0, 0, 0]
        chans.append(c)
        windows += [x0 // s, y0 // s, W // s, H // s]
    T = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, 0, 0, 0]
    outs_f = [torch.zeros(n, f.shape[2], device=device) for f in fulls]
    outs_w = [torch.zeros(n, f.shape[2], device=device) for f in fulls]
    vf, vw = torch.zeros(n, dtype=torch.uint8, device=device), torch.zeros(n, dtype=torch.uint8, device=device)
    ops.sample_sparse(p3d, T, fulls, chans, cams, [0, 0, 0], 1, True, outs_f, vf)
    ops.sample_sparse(p3d, T, wins, chans, cams, [0, 0, 0], 1, True, outs_w, vw, windows)
    assert torch.equal(vf, vw) and 0 < int(vf.sum()) < n  # validity is the FULL level's
    # points whose 2 x 2 texels lie inside the window on every level: the same bits
    x, y = p3d[:, 0], p3d[:, 1]
    inside = (x >= x0 + 16) & (x <= x0 + ww - 17) & (y >= y0 + 16) & (y <= y0 + wh - 17) & (vf != 0)
    assert int(inside.sum()) > 200
    for a, b in zip(outs_f, outs_w):
        assert torch.equal(a[inside], b[inside])


def test_maps_do_not_depend_on_pixels_beyond_the_margin(device):
    """Same image size (so the same layer plans): noise everywhere OUTSIDE [box +- WINDOW_MARGIN] must leave every level's
    maps INSIDE the box bit-identical - the dependency radius holds with the margin the refiner uses."""
    net = UNet(make_synthetic_unet_weights(7), device)
    H = W = 608
    m = PoseTrackerRefiner.WINDOW_MARGIN
    bx0, bx1, by0, by1 = 224, 384, 240, 368  # the points' bounding box (stride-1 pixels)
    g = torch.Generator().manual_seed(11)
    a = torch.zeros(H, W, 3)
    a[by0 - 40:by1 + 40, bx0 - 40:bx1 + 40] = torch.rand(by1 - by0 + 80, bx1 - bx0 + 80, 3, generator=g) * 255
    b = torch.rand(H, W, 3, generator=g) * 255
    keep = torch.zeros(H, W, dtype=torch.bool)
    keep[max(0, by0 - m):by1 + m, max(0, bx0 - m):bx1 + m] = True
    b[keep] = a[keep]
    assert not torch.equal(a, b)
    ma = net.forward_packed(a.to(device).contiguous(), None, False)
    mb = net.forward_packed(b.to(device).contiguous(), None, False)
    for fa, fb, s in zip(ma, mb, (1, 4, 16)):
        ra = fa[by0 // s:-(-by1 // s) + 1, bx0 // s:-(-bx1 // s) + 1]
        rb = fb[by0 // s:-(-by1 // s) + 1, bx0 // s:-(-bx1 // s) + 1]
        assert torch.equal(ra, rb), s
        assert not torch.equal(fa, fb)  # (the noise did reach the maps elsewhere)
    # ... and a margin 48 pixels shorter is NOT enough (the radius is not grossly over-estimated)
    c = torch.rand(H, W, 3, generator=g) * 255
    keep2 = torch.zeros(H, W, dtype=torch.bool)
    keep2[by0 - (m - 48):by1 + (m - 48), bx0 - (m - 48):bx1 + (m - 48)] = True
    c[keep2] = a[keep2]
    mc = net.forward_packed(c.to(device).contiguous(), None, False)
    assert not torch.equal(ma[0][by0:by1 + 1, bx0:bx1 + 1], mc[0][by0:by1 + 1, bx0:bx1 + 1])


def test_windowed_reference_features_equal_the_full_pass(device):
    """End to end on the YCB object with the reference's own 921 x 921 reference camera: extract_reference_features with
    and without the window - same valid points, descriptors equal to fp16-storage / summation-order noise."""
    from pixtrack_amd.geometry import Pose
    from pixtrack_amd.pose_trackers import pixloc_tracker_ycb as ycb
    from pixtrack_amd.synthetic import CRACKER_BOX_AABB, REF_CAMERA_YCB, YCB_QUERY_FXY, make_tracking_assets

    assets = make_tracking_assets(seed=1005, width=640, height=480, n_frames=4, aabb=CRACKER_BOX_AABB, reference_scale=0.3,
                                  n_points=5600, ref_camera=REF_CAMERA_YCB, query_f=YCB_QUERY_FXY[0])
    tr = ycb.PixLocPoseTrackerYCB("", "", "/tmp", "003_cracker_box", device=device, assets=assets)
    refiner = tr.localizer.refiner
    pose = Pose.from_Rt(*assets["gt_poses"][1])
    ref_u8 = tr.get_reference_image(pose)
    assert tuple(ref_u8.shape[:2]) == (921, 921)
    refiner.conf.multiscale = [1]
    dbids = [sorted(refiner.model3d.dbs)[0]]  # (the YCB policy picks its reference inside refine(); any image's points do)
    image, window = refiner.reference_window(dbids, pose, ref_u8)
    assert window is not None and image.shape[0] * image.shape[1] <= 0.65 * 921 * 921, (window, image.shape)
    assert window[0] % 16 == 0 and window[1] % 16 == 0 and window[2:] == (921, 921)
    got = refiner.extract_reference_features(dbids, pose, ref_u8)["1"]
    refiner.conf.reference_window = False
    refiner._window_memo = None
    want = refiner.extract_reference_features(dbids, pose, ref_u8)["1"]
    assert torch.equal(got.valid, want.valid) and int(want.valid.sum()) > 1000
    keep = want.valid.bool()
    for a, b, c in zip(got.packed, want.packed, OUTPUT_DIMS):
        cos = (a[keep, :c] * b[keep, :c]).sum(1)
        assert float(cos.min()) > 0.9999, float(cos.min())
        assert float((a[keep, c] - b[keep, c]).abs().max()) < 2e-3
