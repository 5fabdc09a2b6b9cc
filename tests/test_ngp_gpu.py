"""HIP NeRF renderer vs the numpy oracle on a seeded synthetic hash-grid NeRF.

The ray march (jitter, cone steps, occupancy skipping) is arithmetic-for-arithmetic the
same in both, so both visit the same samples; differences come only from the fp32
accumulation order inside the MLPs (MFMA vs numpy matmul) around fp16 roundings."""
import numpy as np
import pytest
import torch

from oracle import ngp_oracle as NO
from pixtrack_amd.ngp import RenderMode, Testbed, nerf_matrix_to_ngp
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def snap():
    return make_synthetic_nerf(11)


def oracle_model(snap):
    return NO.NgpModel(grid=snap.grid, mlp=snap.mlp_dict(), occupancy=snap.occupancy, cascades=snap.cascades,
                       aabb_scale=snap.aabb_scale, cone_angle=snap.cone_angle, depth_scale=1.0 / snap.scale,
                       linear_colors=snap.linear_colors)


def ngp_camera(offset_dir, dist, up=(0.0, 1.0, 0.0)):
    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    c = 0.5 * (lo + hi)
    eye = c + np.asarray(offset_dir, float) / np.linalg.norm(offset_dir) * dist
    R, _ = look_at_pose(eye, c, up=np.asarray(up, float))
    return np.concatenate([R.T, eye[:, None]], 1)  # camera-to-world in ngp coordinates


def make_testbed(snap, device):
    tb = Testbed(device=device)
    tb.load_snapshot(snap)
    tb.nerf.render_with_camera_distortion = True
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.fov_axis = 0
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    return tb


@pytest.mark.parametrize("mode,W,H,spp,dirn,dist,k1", [
    (0, 64, 48, 2, (0.9, 0.5, 0.3), 1.2, 0.0),
    (1, 64, 48, 2, (0.9, 0.5, 0.3), 1.2, 0.0),
    (0, 50, 37, 3, (-0.4, 0.2, 1.0), 0.9, 0.0),   # ragged tile edges, close-up (box leaves the frame)
    (0, 40, 32, 1, (0.2, -1.0, 0.4), 1.5, -0.08),  # lens distortion
])
def test_render_matches_oracle(device, snap, mode, W, H, spp, dirn, dist, k1):
    cam = ngp_camera(dirn, dist)
    focal = 1.2 * W
    m = oracle_model(snap)
    v = NO.View(cam=cam, focal=focal, width=W, height=H, spp=spp, k1=k1, aabb_min=tuple(PREMIER_PROTEIN_AABB[0]),
                aabb_max=tuple(PREMIER_PROTEIN_AABB[1]), mode=mode)
    ref, st = NO.render(m, v, return_stats=True)

    tb = make_testbed(snap, device)
    tb._snap.k1 = k1
    tb._cam_ngp = cam  # already in ngp coordinates
    import math
    tb.fov = math.degrees(2 * math.atan(W / (2 * focal)))
    tb.render_mode = RenderMode(mode)
    out = tb.render_device(W, H, spp, True, collect_stats=True).cpu().numpy()
    stats = tb.read_stats()
    assert out.shape == (H, W, 4) and np.isfinite(out).all()
    # same rays hit the box; the same samples are visited (bit-identical march)
    assert stats["rays_hit"] == st["rays_hit"]
    assert abs(stats["samples"] - st["samples"]) <= max(4, st["samples"] // 2000)
    scale = max(1.0, float(np.abs(ref[..., :3]).max()))
    diff = np.abs(out - ref) / scale
    assert diff.max() < 1e-2, diff.max()
    assert diff.mean() < 5e-4, diff.mean()
    assert (ref[..., 3] > 0.99).mean() > 0.05  # the object is really there


def test_set_nerf_camera_matrix_convention(device, snap):
    """set_nerf_camera_matrix applies instant-ngp's nerf->ngp map (scale 0.33, offset 0.5,
    y/z flip, axis cycle): rendering through it equals rendering with the ngp matrix."""
    tb = make_testbed(snap, device)
    cam_ngp = ngp_camera((0.9, 0.5, 0.3), 1.2)
    # invert the map to get the NeRF-convention matrix
    n = cam_ngp[[2, 0, 1], :].copy()
    n[:, 3] = (n[:, 3] - 0.5) / 0.33
    n[:, 1] *= -1
    n[:, 2] *= -1
    assert np.allclose(nerf_matrix_to_ngp(n, 0.33, 0.5), cam_ngp)
    tb.fov = 45.0
    tb.set_nerf_camera_matrix(n)
    a = tb.render_device(32, 24, 1, True).cpu().numpy()
    tb._cam_ngp = cam_ngp
    b = tb.render_device(32, 24, 1, True).cpu().numpy()
    assert np.array_equal(a, b)
    host = tb.render(32, 24, 1, True)  # pyngp contract: host float32 HxWx4
    assert isinstance(host, np.ndarray) and host.dtype == np.float32 and np.array_equal(host, b)


def test_background_and_miss(device, snap):
    """A view that looks away from the box returns the (premultiplied) background."""
    tb = make_testbed(snap, device)
    cam = ngp_camera((0.9, 0.5, 0.3), 1.2)
    cam[:, 2] *= -1  # look the other way
    cam[:, 0] *= -1
    tb._cam_ngp = cam
    tb.fov = 45.0
    out = tb.render_device(24, 16, 2, True).cpu().numpy()
    assert np.all(out == 0.0)
    tb.background_color = [0.5, 0.25, 1.0, 1.0]
    out = tb.render_device(24, 16, 2, True).cpu().numpy()
    assert np.allclose(out, np.array([0.5, 0.25, 1.0, 1.0], np.float32))


def test_network_query_matches_oracle(device, snap):
    """KAT-7: hash-grid encode + both MLPs at given points (dense random weights so every
    row/column of every layer matters; an A=I style structured test would hide transposes)."""
    import ctypes as C

    from pixtrack_amd import _lib
    from pixtrack_amd.ngp import NerfSnapshot

    rg = np.random.default_rng(5)
    shapes = dict(d1=(64, 32), d2=(16, 64), c1=(64, 32), c2=(64, 64), c3=(16, 64))
    rnd = {k: (rg.normal(size=v) * (2.0 / v[1]) ** 0.5).astype(np.float16) for k, v in shapes.items()}
    s2 = NerfSnapshot(grid=snap.grid, mlp=np.concatenate([rnd[k].ravel() for k in ("d1", "d2", "c1", "c2", "c3")]),
                      occupancy=snap.occupancy)
    n = 1000  # not a multiple of 64: the tail wave is partially filled
    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    x = rg.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rg.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tb = Testbed(device=device)
    tb.load_snapshot(s2)
    out = torch.zeros(n, 4, device=device)
    xd, dd = torch.from_numpy(x).to(device), torch.from_numpy(d).to(device)
    _lib.check(_lib.lib().pxt_ngp_query(tb._ctx, xd.data_ptr(), dd.data_ptr(), n, out.data_ptr(),
                                        _lib.stream_ptr(device)), "pxt_ngp_query")
    torch.cuda.synchronize()
    m = NO.NgpModel(grid=s2.grid, mlp=s2.mlp_dict(), occupancy=s2.occupancy, cascades=3, aabb_scale=4.0)
    unit = ((x - np.float32(0.5 - 2.0)) * np.float32(0.25)).astype(np.float32)
    den, rgb = NO.network(m, unit, d)
    got = out.cpu().numpy()
    assert np.abs(got[:, 0] - np.log(den)).max() < 5e-3
    assert np.abs(got[:, 1:] - rgb).max() < 5e-3


def test_render_both_equals_two_renders(device, snap):
    """One march producing Shade and Depth == the two separate renders, bit for bit."""
    tb = make_testbed(snap, device)
    tb._cam_ngp = ngp_camera((0.9, 0.5, 0.3), 1.2)
    tb.fov = 45.0
    W, H, spp = 72, 50, 3
    tb.render_mode = RenderMode.Shade
    a = tb.render_device(W, H, spp, True)
    tb.render_mode = RenderMode.Depth
    b = tb.render_device(W, H, spp, True)
    tb.render_mode = RenderMode.Shade
    c, d = tb.render_both_device(W, H, spp)
    assert torch.equal(a, c) and torch.equal(b, d)
    assert float(b[..., 0].max()) > 0


def test_render_sequence_with_scratch_growth_and_pipeline_changes(device, snap):
    """A render's last kernel zeroes the round counters for the next one (no memset in front of a render); a
    (re)allocated scratch or another pipeline count must not see stale counters: the same view renders to the
    same bits before and after renders of other sizes and pipeline counts."""
    tb = make_testbed(snap, device)
    tb._cam_ngp = ngp_camera((0.9, 0.5, 0.3), 1.69)
    tb.fov = 45.0
    big = (640, 480, 8)     # 2.4 M rays: two pipelines by default
    ref_rgba, ref_depth = (t.clone() for t in tb.render_both_device(*big))
    small = tuple(t.clone() for t in tb.render_both_device(72, 50, 3))
    for _ in range(2):
        a, b = tb.render_both_device(*big)
        assert torch.equal(a, ref_rgba) and torch.equal(b, ref_depth)
    tb.set_pipelines(1)
    a, b = tb.render_both_device(*big)
    assert torch.equal(a, ref_rgba) and torch.equal(b, ref_depth)
    tb.set_pipelines(3)
    a, b = tb.render_both_device(*big)
    assert torch.equal(a, ref_rgba) and torch.equal(b, ref_depth)
    tb.set_pipelines(0)
    c, d = tb.render_both_device(72, 50, 3)
    assert torch.equal(c, small[0]) and torch.equal(d, small[1])
    a, b = tb.render_both_device(*big)
    assert torch.equal(a, ref_rgba) and torch.equal(b, ref_depth)


def test_shade_colour_space_flag_on_the_device(device, snap):
    """`linear_colors` on the device: with ONE pass per pixel the Shade image of a snapshot trained on LDR images is
    instant-ngp's srgb_to_linear of the image of the same snapshot flagged as trained in linear colours (the conversion
    acts on the finished ray, before the spp mean); alpha and the Depth image do not depend on the flag; both agree with
    the oracle's."""
    import dataclasses
    import math

    W, H = 96, 72
    cam = ngp_camera((0.9, 0.5, 0.3), 1.3)
    imgs = {}
    for flag in (False, True):
        s2 = dataclasses.replace(snap, linear_colors=flag, k1=0.0)  # (other tests of this module set a lens on the shared snapshot)
        tb = make_testbed(s2, device)
        tb._cam_ngp = cam
        tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
        tb.render_mode = RenderMode.Shade
        shade = tb.render_device(W, H, 1, True).cpu().numpy()
        tb.render_mode = RenderMode.Depth
        depth = tb.render_device(W, H, 1, True).cpu().numpy()
        imgs[flag] = (shade, depth)
        m = oracle_model(s2)
        v = NO.View(cam=cam, focal=1.2 * W, width=W, height=H, spp=1, aabb_min=tuple(PREMIER_PROTEIN_AABB[0]),
                    aabb_max=tuple(PREMIER_PROTEIN_AABB[1]), mode=0)
        ref = NO.render(m, v)
        assert np.abs(shade - ref).max() < 1e-2 and np.abs(shade - ref).mean() < 5e-4
    lin, conv = imgs[True][0], imgs[False][0]
    assert lin[..., 3].max() > 0.9 and np.array_equal(lin[..., 3], conv[..., 3])
    assert np.array_equal(imgs[True][1], imgs[False][1])  # Depth: never converted
    want = NO.srgb_to_linear(lin[..., :3])
    assert np.abs(conv[..., :3] - want).max() < 2e-6 + 1e-5 * want.max()
    assert float(np.abs(conv[..., :3] - lin[..., :3]).max()) > 0.05  # (and the flag does something)
