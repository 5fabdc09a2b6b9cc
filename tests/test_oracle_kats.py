"""Known-answer tests of the CPU oracles themselves:
the reference pins nothing at this boundary, so the oracle's own arithmetic is checked
against identities, finite differences and closed forms before the HIP path is compared to it."""
import math

import numpy as np
import pytest
import torch

from oracle import lm_oracle as O
from oracle import ngp_oracle as NO
from oracle import unet_oracle as UO
from pixtrack_amd import geometry as G
from pixtrack_amd.synthetic import make_lm_scene


def test_kat1_so3exp_identities():
    assert torch.equal(O.so3exp(torch.zeros(3, dtype=torch.float64)), torch.eye(3, dtype=torch.float64))
    g = torch.Generator().manual_seed(0)
    w = torch.randn(50, 3, generator=g, dtype=torch.float64)
    R = O.so3exp(w)
    assert torch.allclose(R.transpose(-1, -2) @ R, torch.eye(3, dtype=torch.float64).expand(50, 3, 3), atol=1e-12)
    assert torch.allclose(torch.linalg.det(R), torch.ones(50, dtype=torch.float64), atol=1e-12)
    # continuity across the small-angle branch at 1e-7
    d = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    d = d / d.norm()
    a, b = O.so3exp(d * 0.99e-7), O.so3exp(d * 1.01e-7)
    assert (a - b).abs().max() < 1e-8
    # product and host class agree with the oracle
    assert torch.allclose(G.so3exp_map(w), R)


def _fd(fn, x, eps=1e-6):
    J = []
    for i in range(x.numel()):
        d = torch.zeros_like(x)
        d[i] = eps
        J.append((fn(x + d) - fn(x - d)) / (2 * eps))
    return torch.stack(J, -1)


@pytest.mark.parametrize("dist", [(0.0, 0.0), (-0.12, 0.03), (-0.1, 0.02, 0.003, -0.002)])
def test_kat2_projection_jacobians_vs_finite_differences(dist):
    cam = torch.tensor([640.0, 480.0, 700.0, 690.0, 319.5, 239.5, *dist], dtype=torch.float64)
    g = torch.Generator().manual_seed(1)
    p = torch.randn(20, 3, generator=g, dtype=torch.float64) * 0.4 + torch.tensor([0.0, 0.0, 4.0], dtype=torch.float64)
    J = O.J_world2image(cam, p)
    for k in range(p.shape[0]):
        Jfd = _fd(lambda q: O.world2image(cam, q[None])[0][0], p[k])
        assert torch.allclose(J[k], Jfd, rtol=1e-6, atol=1e-6)
    # left-update Jacobian: d(exp(delta) T p)/d(delta) = [I | -[p]x], translation first
    R0, t0 = O.so3exp(torch.tensor([0.1, -0.2, 0.3], dtype=torch.float64)), torch.tensor([0.3, -0.1, 2.0], dtype=torch.float64)
    x = torch.tensor([0.2, 0.5, -0.3], dtype=torch.float64)
    pc = O.pose_transform(R0, t0, x[None])[0]

    def f(delta):
        Rd = O.so3exp(delta[3:])
        Rn, tn = O.pose_compose(Rd, delta[:3], R0, t0)
        return O.pose_transform(Rn, tn, x[None])[0]

    assert torch.allclose(O.J_transform(pc[None])[0], _fd(f, torch.zeros(6, dtype=torch.float64)), atol=1e-8)
    # host Camera/Pose classes agree with the oracle functions
    C = G.Camera(cam.float())
    p2, v2 = C.world2image(p.float())
    p1, v1 = O.world2image(cam.float(), p.float())
    assert torch.allclose(p1, p2) and torch.equal(v1, v2)
    assert torch.allclose(C.J_world2image(p.float())[0], O.J_world2image(cam.float(), p.float()))


def test_kat3_bilinear_sampler():
    g = torch.Generator().manual_seed(2)
    F = torch.randn(5, 9, 11, generator=g, dtype=torch.float64)
    pts = torch.tensor([[3.0, 4.0], [7.0, 2.0], [1.0, 1.0]], dtype=torch.float64)
    val, _ = O.interpolate_bilinear(F, pts)
    assert torch.allclose(val, torch.stack([F[:, 4, 3], F[:, 2, 7], F[:, 1, 1]]), atol=1e-12)
    ys, xs = torch.meshgrid(torch.arange(9.0, dtype=torch.float64), torch.arange(11.0, dtype=torch.float64), indexing="ij")
    ramp = (0.7 * xs - 0.3 * ys)[None]
    pts = torch.tensor([[3.3, 4.6], [6.5, 2.2]], dtype=torch.float64)
    val, grad = O.interpolate_bilinear(ramp, pts, return_gradients=True)
    assert torch.allclose(val[:, 0], 0.7 * pts[:, 0] - 0.3 * pts[:, 1], atol=1e-12)
    assert torch.allclose(grad[:, 0], torch.tensor([[0.7, -0.3]] * 2, dtype=torch.float64), atol=1e-12)
    assert O.mask_in_image(torch.tensor([[1.0, 1.0], [0.5, 3.0], [9.0, 7.0], [9.5, 7.0]]), 11, 9, 1).tolist() == [True, False, True, False]


def test_kat4_damped_solve_vs_numpy():
    rng = np.random.default_rng(3)
    A = rng.normal(size=(6, 6))
    H = torch.from_numpy(A @ A.T + 0.1 * np.eye(6))
    g = torch.from_numpy(rng.normal(size=6))
    lam = torch.from_numpy(10.0 ** rng.uniform(-6, 0, size=6))
    delta = O.optimizer_step(g, H, lam, ok=True)
    Hd = H.numpy() + np.diag(np.maximum(np.diag(H.numpy()) * lam.numpy(), 1e-6))
    assert np.allclose(delta.numpy(), -np.linalg.solve(Hd, g.numpy()), rtol=1e-10)
    assert torch.equal(O.optimizer_step(g, H, lam, ok=False), torch.zeros(6, dtype=torch.float64))
    assert np.allclose(O.damping_lambda(torch.zeros(6)).numpy(), 10 ** -0.5)


def test_kat5_synthetic_scene_pose_recovery():
    """Smooth random fields, N=2000 points, (3 deg, 2 cm) perturbation -> recovered to the
    ground truth well inside the parity tolerance (fp64 oracle)."""
    sc = make_lm_scene(seed=1001, width=320, height=240, n_points=2000, sigma_px=2.0)
    lam = [O.damping_lambda(torch.full((6,), -2.0)) for _ in range(3)]
    conf = O.LMConf(dt_stop_criteria=1e-6, dR_stop_criteria=1e-5, grad_stop_criteria=0.0, num_iters=40)
    log = O.LMLog()
    ret = O.refine_pose_using_features(sc.feats_query, sc.scales, sc.camera._data, torch.from_numpy(sc.R_init),
                                       torch.from_numpy(sc.t_init), sc.feats_ref, torch.from_numpy(sc.p3d), lam, conf,
                                       log=log, dtype=torch.float64)
    assert ret["success"]
    assert O.rotation_angle_rad(ret["R"], torch.from_numpy(sc.R_gt)) < 1e-3
    assert float((ret["t"] - torch.from_numpy(sc.t_gt)).norm()) < 1e-3
    assert all(c[-1] <= c[0] * 1.0001 for c in log.costs)  # cost never ends above where it started


def test_losses_closed_forms():
    x = torch.tensor([0.0, 0.01, 0.5, 4.0])
    l, w = O.make_loss("barron", 0.0, 0.1)(x)
    assert torch.allclose(l, 0.01 * 2 * torch.log1p(0.5 * x / 0.01)) and torch.allclose(w, 2 / (x / 0.01 + 2))
    l, w = O.make_loss("squared")(x)
    assert torch.equal(l, x) and torch.equal(w, torch.ones(4))
    l2, w2 = O.make_loss("barron", 2.0, 0.3)(x)
    assert torch.allclose(l2, x) and torch.allclose(w2, torch.ones(4))
    lh, wh = O.make_loss("huber", 0.0, 1.0)(x)
    assert torch.allclose(lh, torch.tensor([0.0, 0.01, 0.5, 3.0])) and torch.allclose(wh, torch.tensor([1.0, 1.0, 1.0, 0.5]))


def test_kat7_hash_grid_layout_and_indices():
    m = NO.NgpModel()
    lay = NO.grid_level_layout(m)
    assert sum(n for _, _, _, n, _ in lay) * 2 == 13074912  # notebooks/Render YCB GT Poses .ipynb:147
    assert [h for *_, h in lay][:5] == [False, False, False, False, True]
    assert lay[0][1] == 16 and lay[15][1] == 8193
    # dense level: value at a grid vertex is the stored entry (x = (g - 0.5) / scale)
    grid = np.zeros((sum(n for _, _, _, n, _ in lay), 2), np.float16)
    scale, res, off, size, hashed = lay[2]
    gx, gy, gz = 5, 7, 3
    grid[off + gx + gy * res + gz * res * res] = [0.75, -0.5]
    m.grid = grid
    x = np.array([[(gx - 0.5) / scale, (gy - 0.5) / scale, (gz - 0.5) / scale]], np.float32) + 1e-7
    enc = NO.hash_grid_encode(m, x).astype(np.float32)
    assert np.allclose(enc[0, 4:6], [0.75, -0.5], atol=2e-3) and np.abs(enc[0, :4]).max() == 0
    # hashed level index of a known coordinate
    c = np.array([[17, 33, 65]], np.uint32)
    want = ((c[:, 0] * np.uint32(1)) ^ (c[:, 1] * np.uint32(2654435761)) ^ (c[:, 2] * np.uint32(805459861))) % np.uint32(1 << 19)
    assert int(want[0]) == ((17 ^ ((33 * 2654435761) & 0xFFFFFFFF) ^ ((65 * 805459861) & 0xFFFFFFFF)) % (1 << 19))


def test_kat7_constant_density_slab_transmittance():
    """A fully occupied grid whose MLP returns a constant density: alpha of a ray that crosses
    the unit cube equals 1 - exp(-sigma * path length) up to the step quantisation."""
    lay = NO.grid_level_layout(NO.NgpModel())
    n = sum(k for _, _, _, k, _ in lay)
    grid = np.zeros((n, 2), np.float16)
    grid[:, 0] = 1.0  # every level's feature 0 is 1 everywhere -> interpolates to exactly 1
    d1 = np.zeros((64, 32), np.float16)
    d1[0, 0] = 1.0
    d2 = np.zeros((16, 64), np.float16)
    sigma = 2.0
    d2[0, 0] = np.float16(math.log(sigma))
    mlp = dict(d1=d1, d2=d2, c1=np.zeros((64, 32), np.float16), c2=np.zeros((64, 64), np.float16), c3=np.zeros((16, 64), np.float16))
    occ = np.full(128**3 // 8, 255, np.uint8)
    m = NO.NgpModel(cascades=1, aabb_scale=1.0, cone_angle=0.0, grid=grid, mlp=mlp, occupancy=occ)
    cam = np.array([[1.0, 0, 0, 0.5], [0, 1.0, 0, 0.5], [0, 0, 1.0, -1.0]])  # looks down +z through the cube
    v = NO.View(cam=cam, focal=1e6, width=2, height=2, spp=1, min_transmittance=1e-12, background=(0, 0, 0, 0))
    img = NO.render(m, v)
    sig16 = float(np.exp(np.float32(np.float16(math.log(sigma)))))
    assert np.allclose(img[..., 3], 1 - math.exp(-sig16 * 1.0), atol=3e-3)
    # sigmoid(0) = 0.5, premultiplied, then instant-ngp's srgb_to_linear on the finished ray (Shade, LDR training set)
    assert np.allclose(img[..., :3], NO.srgb_to_linear(0.5 * img[..., 3:4]), atol=1e-6)
    m.linear_colors = True  # a snapshot trained in linear colours: no conversion
    assert np.allclose(NO.render(m, v)[..., :3], 0.5 * img[..., 3:4], atol=1e-6)
    m.linear_colors = False
    vd = NO.View(cam=cam, focal=1e6, width=2, height=2, spp=1, min_transmittance=1e-12, background=(0, 0, 0, 0), mode=1)
    dimg = NO.render(m, vd)  # Depth mode is never converted: expected depth of a uniform slab entered at z = 1
    assert np.all(dimg[..., 0] > dimg[..., 3] * 1.0) and np.all(dimg[..., 0] < dimg[..., 3] * 2.0)
    # an opaque ray: the early-out renormalises to exactly sigmoid(0) = 0.5 -> srgb_to_linear(0.5) = 0.2140
    d2[0, 0] = np.float16(math.log(400.0))
    vo = NO.View(cam=cam, focal=1e6, width=2, height=2, spp=1, min_transmittance=1e-4, background=(0, 0, 0, 0))
    oimg = NO.render(m, vo)
    assert np.allclose(oimg[..., 3], 1.0, atol=1e-6) and np.allclose(oimg[..., :3], 0.21404, atol=2e-5)


def test_kat7_srgb_to_linear_known_values():
    x = np.array([0.0, 0.04045, 0.0405, 0.5, 1.0], np.float32)
    want = [0.0, 0.04045 / 12.92, ((0.0405 + 0.055) / 1.055) ** 2.4, ((0.5 + 0.055) / 1.055) ** 2.4, 1.0]
    assert np.allclose(NO.srgb_to_linear(x), want, rtol=2e-6, atol=1e-9)
    assert abs(float(NO.srgb_to_linear(np.float32(0.5))) - 0.21404114) < 1e-6


def test_kat6_unet_shapes_and_resize():
    from pixtrack_amd.unet import make_synthetic_unet_weights

    w = make_synthetic_unet_weights(seed=1)
    feats, confs = UO.unet_forward(w, torch.rand(3, 75, 100))
    assert [tuple(f.shape) for f in feats] == [(32, 64, 96), (128, 16, 24), (128, 4, 6)]
    assert all(((c > 0) & (c < 1)).all() for c in confs)
    n_params = sum(v.numel() for k, v in w.items() if "bn_mean" not in k and "bn_var" not in k)
    assert n_params == 15_712_387  # "15.7 M params" 
    img = np.arange(6 * 8 * 3, dtype=np.float32).reshape(6, 8, 3)
    assert np.array_equal(UO.cv2_resize_linear(img, 8, 6), img)
    small = UO.cv2_resize_linear(img, 4, 3)
    assert np.allclose(small[0, 0], (img[0, 0] + img[0, 1] + img[1, 0] + img[1, 1]) / 4)


def test_kat8_tracker_policy_cost_gate():
    """pixloc_tracker_r9.py:258-268: the threshold is frozen at 1.1 x the first frame's cost."""
    costs = [0.010, 0.0105, 0.0111, 0.009, float("nan"), 0.0109]
    opt_ok = [True, True, True, False, True, True]
    thr, out = None, []
    for c, ok in zip(costs, opt_ok):
        if thr is None:
            thr = c + 0.1 * c
        out.append(bool(ok and c <= thr))
    assert out == [True, True, False, False, False, True] and abs(thr - 0.011) < 1e-12


def test_kat7_tiled_render_is_the_serial_render():
    """oracle.ngp_oracle.render_parallel (rows dealt to worker processes: fixture generators, bench.py's cpu_baseline)
    returns the serial render bit for bit, counts included, whatever the tiling."""
    from oracle import frame_oracle as FO
    from pixtrack_amd.synthetic import make_tracking_assets

    a = make_tracking_assets(seed=1002, width=96, height=72, n_frames=1, n_points=600)
    ngp = FO.ngp_model(a["snapshot"])
    R, t = a["gt_poses"][0]
    qcam = FO.colmap_camera_to_pix(a["query_camera"])
    for mode in (0, 1):
        v = FO.nerf_view(a["snapshot"], a["nerf2sfm"], a["aabb"], R, t, qcam, mode, 2)
        img, st = NO.render(ngp, v, True)
        assert st["samples"] > 1000 and img[..., 3].max() > 0.5
        for procs, step in ((2, 0), (3, 7)):
            img2, st2 = NO.render_parallel(ngp, v, procs, True, rows_per_job=step)
            assert np.array_equal(img, img2) and st == st2
        part = NO.render(ngp, v, rows=(30, 41))
        assert np.array_equal(part, img[30:41])


def test_render_box_with_swapped_corners_warns_once(monkeypatch):
    """config/motor_core.sh writes its y bounds max-first.  The corners are sorted per axis - with a warning,
    never silently; PXT_STRICT_AABB=1 keeps them as given (instant-ngp: an empty box)."""
    import warnings

    from pixtrack_amd.utils import ingp_utils

    class _TB:  # the part of Testbed that initialize_ingp touches, without a device
        def __init__(self, *a, **k):
            import types

            self.nerf = types.SimpleNamespace()
            self.render_aabb = types.SimpleNamespace(min=None, max=None)

        def load_snapshot(self, p):
            pass

    import pixtrack_amd.ngp as ngp_mod

    monkeypatch.setattr(ngp_mod, "Testbed", _TB)
    monkeypatch.setattr(ingp_utils, "_WARNED_AABB", [])
    box = [[0.1, 0.9, 0.2], [0.6, 0.3, 0.8]]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        tb = ingp_utils.initialize_ingp("x", box)
        tb2 = ingp_utils.initialize_ingp("x", box)
    assert tb.render_aabb.min == [0.1, 0.3, 0.2] and tb.render_aabb.max == [0.6, 0.9, 0.8] and tb2.render_aabb.min == tb.render_aabb.min
    assert len([x for x in w if "min > max" in str(x.message)]) == 1
    monkeypatch.setenv("PXT_STRICT_AABB", "1")
    tb3 = ingp_utils.initialize_ingp("x", box)
    assert tb3.render_aabb.min == [0.1, 0.9, 0.2] and tb3.render_aabb.max == [0.6, 0.3, 0.8]


def test_kat6_power_of_two_rescale_leaves_the_network_unchanged():
    """pixtrack_amd.unet.rescale_unet_weights: storing layer l's activations divided by a power of two c[l] (filters, bias /
    BatchNorm and every consumer's input channels compensated) is the SAME function - checked bit for bit in fp32 on the
    oracle, skip connections, up-sampling and all three heads included."""
    from pixtrack_amd.unet import make_synthetic_unet_weights, rescale_unet_weights

    w = make_synthetic_unet_weights(7, bn_trivial=False)
    img = torch.rand(3, 75, 100, generator=torch.Generator().manual_seed(2))
    f0, c0 = UO.unet_forward(w, img)
    c = [2.0 ** ((i * 7) % 5 - 2) for i in range(17)]  # 1/4 .. 4, different for every layer
    f1, c1 = UO.unet_forward(rescale_unet_weights(w, c), img)
    for a, b in zip(f0 + c0, f1 + c1):
        assert torch.equal(a, b)
    ones = rescale_unet_weights(w, [1.0] * 17)
    assert all(torch.equal(ones[k], w[k]) for k in w)
