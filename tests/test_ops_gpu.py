"""`torch.ops.pixtrack.*` called directly with torch.Tensors on the ROCm device: the LM
op against the CPU oracle, the sampling / image ops against numpy, the render ops against the
oracle renderer.  The host classes route through these same ops (optimizer.py, refiner.py, unet.py,
ngp.py), so this is the boundary the reference-side binding of INTEGRATION.md section B uses."""
import numpy as np
import pytest
import torch

from oracle import frame_oracle as FO
from oracle import lm_oracle as O
from oracle import ngp_oracle as NO
from pixtrack_amd import _lib, ops  # noqa: F401  (registers the ops)
from pixtrack_amd.synthetic import make_lm_scene, make_synthetic_nerf, PREMIER_PROTEIN_AABB
from tests.test_lm_gpu import CONSTS, lambdas, pack_level

pytestmark = pytest.mark.gpu
P = torch.ops.pixtrack


def test_lm_refine_op_matches_oracle(device):
    sc = make_lm_scene(seed=1001, width=320, height=240, n_points=2048, sigma_px=2.0)
    lam = lambdas(CONSTS)
    conf = O.LMConf()
    ref = O.refine_pose_using_features(sc.feats_query, sc.scales, sc.camera._data, torch.from_numpy(sc.R_init),
                                       torch.from_numpy(sc.t_init), sc.feats_ref, torch.from_numpy(sc.p3d), lam, conf)
    fmaps, frefs, chans, cams, ndist, lams = [], [], [], [], [], []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        fmaps.append(fmap), frefs.append(fref), chans.append(Cc)
        cams += cam.as10().tolist()
        ndist.append(int(cam._data.shape[-1] - 6))
        lams += lam[level].tolist()
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    nh = 16 + _lib.PXT_MAX_LEVELS
    record = torch.zeros(nh + 3 * conf.num_iters * _lib.PXT_LM_LOG_STRIDE, device=device)  # device record this time
    T0 = sc.T_init.as12().reshape(-1).tolist()
    out = P.lm_refine(p3d, None, fmaps, frefs, chans, cams, ndist, lams, T0, conf.num_iters, conf.pad, 2, 0.0, 0.1,
                      1e-4, 5e-3, 5e-2, 10, 0, record, ws, True)
    assert out is None
    rec = record.cpu()
    assert rec[15] == 1.0 and rec[13] == 0.0 and rec[12] == 0.0
    T = rec[:12].double()
    assert O.rotation_angle_rad(T[:9].reshape(3, 3), ref["R"]) < 1e-3
    assert float((T[9:] - ref["t"]).norm()) < 1e-3
    iters = [int(rec[16 + l]) for l in range(3)]
    log = rec[nh:].view(3, conf.num_iters, _lib.PXT_LM_LOG_STRIDE)
    assert all(1 <= n <= conf.num_iters for n in iters) and float(log[0, 0, 0]) > 0
    # argument errors are exceptions, not silent failures
    with pytest.raises(_lib.PxtError):
        P.lm_refine(p3d, None, fmaps, frefs[:2], chans, cams, ndist, lams, T0, 150, 1, 2, 0.0, 0.1, 1e-4, 5e-3, 5e-2,
                    10, 0, record, ws, True)


def test_sample_and_image_ops(device):
    rng = np.random.default_rng(3)
    h, w, Cc = 30, 40, 8
    fmap = torch.from_numpy(rng.normal(size=(h, w, 12)).astype(np.float32)).to(device)
    pts = torch.tensor([[5.0, 7.0, 1.0], [10.5, 3.25, 1.0], [0.2, 0.2, 1.0], [38.0, 28.0, 1.0]], device=device)
    out = torch.empty(4, 12, device=device)
    valid = torch.empty(4, dtype=torch.uint8, device=device)
    P.sample_sparse(pts, [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, 0, 0, 0], [fmap], [Cc],
                    [float(w), float(h), 1.0, 1.0, 0, 0, 0, 0, 0, 0], [0], 1, False, [out], valid)
    f = fmap.cpu().numpy()
    assert valid.cpu().tolist() == [1, 1, 0, 1]  # (0.2, 0.2) is inside the pad-1 border
    assert np.allclose(out[0, :Cc].cpu().numpy(), f[7, 5, :Cc], atol=1e-6)  # integer coordinates: the texel
    want = 0.5 * (0.75 * f[3, 10, :Cc] + 0.25 * f[4, 10, :Cc]) + 0.5 * (0.75 * f[3, 11, :Cc] + 0.25 * f[4, 11, :Cc])
    assert np.allclose(out[1, :Cc].cpu().numpy(), want, atol=1e-5)

    depth = torch.zeros(24, 32, 4, device=device)
    depth[8:16, 10:20, 0] = 0.5
    mask = torch.empty(24, 32, dtype=torch.uint8, device=device)
    P.depth_mask(depth, 1, 5, mask, torch.empty(2 * 24 * 32, dtype=torch.uint8, device=device))
    assert np.array_equal(mask.cpu().numpy(), FO.depth_mask(depth.cpu().numpy()))

    rgba = torch.from_numpy(rng.uniform(0, 1.2, size=(9, 11, 4)).astype(np.float32)).to(device)
    u8 = torch.empty(9, 11, 3, dtype=torch.uint8, device=device)
    P.rgba_to_u8(rgba, 0.0, u8)
    assert np.array_equal(u8.cpu().numpy(), FO.to_u8(rgba.cpu().numpy()))

    from oracle.unet_oracle import cv2_resize_linear

    src = torch.from_numpy(rng.uniform(0, 255, size=(48, 64, 3)).astype(np.float32)).to(device)
    dst = torch.empty(18, 24, 3, device=device)
    P.resize_linear(src, dst)
    assert np.allclose(dst.cpu().numpy(), cv2_resize_linear(src.cpu().numpy(), 24, 18), atol=2e-3)


def test_render_ops_match_oracle(device):
    from pixtrack_amd.ngp import Testbed, TestbedMode

    snap = make_synthetic_nerf(seed=21)
    tb = Testbed(TestbedMode.Nerf, device=device)
    tb.load_snapshot(snap)
    lo, hi = PREMIER_PROTEIN_AABB
    cam = np.array([[1.0, 0, 0, 0.49], [0, 1.0, 0, -0.02], [0, 0, 1.0, -0.9]], np.float32)  # looks along +z at the box
    W, H, spp, focal = 64, 48, 2, 80.0
    view = [float(x) for x in cam.reshape(-1)] + [focal, 0.0] + list(map(float, lo)) + list(map(float, hi)) + \
        [1.0, 1.0, 1.0, 0.0, 1e-7]
    out = torch.empty(H, W, 4, device=device)
    stats = torch.zeros(4, dtype=torch.int64, device=device)
    P.ngp_render(tb._ctx_int(), view, W, H, spp, 0, out, stats)
    m = FO.ngp_model(snap)
    want, st = NO.render(m, NO.View(cam=cam, focal=focal, width=W, height=H, spp=spp, k1=0.0, aabb_min=tuple(lo),
                                    aabb_max=tuple(hi), mode=0, background=(1.0, 1.0, 1.0, 0.0)), return_stats=True)
    got = out.cpu().numpy()
    assert stats[0].item() == st["samples"] and stats[1].item() == st["rays_hit"] and st["samples"] > 1000
    assert np.abs(got - want).max() < 1e-2 and np.abs(got - want).mean() < 5e-4
    rgba, depth = torch.empty(H, W, 4, device=device), torch.empty(H, W, 4, device=device)
    P.ngp_render_both(tb._ctx_int(), view, W, H, spp, rgba, depth, None)
    assert torch.equal(rgba, out)
    d1 = torch.empty(H, W, 4, device=device)
    P.ngp_render(tb._ctx_int(), view, W, H, spp, 1, d1, None)
    assert torch.equal(depth, d1)
    with pytest.raises(_lib.PxtError):
        P.ngp_render(tb._ctx_int(), view[:-1], W, H, spp, 0, out, None)
