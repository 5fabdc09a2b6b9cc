"""The next frame's render enqueued behind the LM launch (pxt_ngp_render_both_from_pose + the tracker's render_ahead):
the device-side pose -> camera conversion must give the bits of the host chain, the render those of a plain
render_both of that camera, and a tracked sequence the same poses with and without it."""
import math

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from pixtrack_amd.geometry import Pose
from pixtrack_amd.ngp import Testbed, nerf_matrix_to_ngp
from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, make_synthetic_nerf, make_tracking_assets, render_query_frames
from pixtrack_amd.utils.ingp_utils import sfm_to_nerf_pose
from pixtrack_amd.utils.pose_utils import get_camera_in_world_from_pixpose

pytestmark = pytest.mark.gpu


def test_device_camera_and_render_equal_the_host_path(device):
    tb = Testbed(device=device)
    tb.load_snapshot(make_synthetic_nerf(11))
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    W, H = 160, 120
    tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
    rng = np.random.default_rng(4)
    n2s = {"centroid": rng.normal(size=3) * 0.2, "avglen": 2.3, "R": np.eye(4), "totp": rng.normal(size=3) * 0.1}
    n2s["R"][:3, :3] = Rotation.random(random_state=3).as_matrix()
    conv = tb.pose_conversion(n2s)
    same = 0
    for k in range(24):
        R = Rotation.random(random_state=10 + k).as_matrix().astype(np.float32)
        t = (rng.normal(size=3) * 0.5 + np.array([0, 0, 2.0])).astype(np.float32)
        rec = torch.zeros(32, dtype=torch.float32).pin_memory()
        rec[:9] = torch.from_numpy(R.reshape(-1))
        rec[9:12] = torch.from_numpy(t)
        rgba, depth, cam_out = tb.render_both_from_pose_device(W, H, 2, rec, conv)
        torch.cuda.synchronize()
        got = cam_out.numpy()
        assert got[12] == 1.0
        pose = Pose.from_Rt(R.astype(np.float64), t.astype(np.float64))
        tb.set_nerf_camera_matrix(np.asarray(sfm_to_nerf_pose(n2s, get_camera_in_world_from_pixpose(pose)))[:3, :])
        want = np.asarray(tb._cam_ngp, np.float32).reshape(-1)
        assert np.allclose(got[:12], want, rtol=0, atol=1e-6)
        if np.array_equal(got[:12].view(np.uint32), want.view(np.uint32)):
            same += 1
            r2, d2 = tb.render_both_device(W, H, 2)
            assert torch.equal(rgba, r2) and torch.equal(depth, d2)
    assert same >= 22  # the two chains may differ in a last float64 bit (BLAS vs sequential products); the tracker checks


@pytest.mark.parametrize("coincide", [True, False])
def test_tracking_is_identical_with_and_without_render_ahead(device, coincide):
    """coincide False: mask and reference image are two renders of different cameras (the real-asset case), queued as ONE
    chain behind the LM launch, each context's camera from its own slot."""
    n = 14
    assets = make_tracking_assets(seed=1002, width=320, height=240, n_frames=n)
    hist = {}
    for ahead in (False, True):
        tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
        tr.fuse_identical_views = coincide
        tr.render_ahead = bool(ahead)
        frames = render_query_frames(assets, tr.testbed)
        for i in range(n):
            tr.run_single_frame((f"{i:06d}.png", frames[i]))
        hist[ahead] = np.stack([np.concatenate([a.ravel() for a in tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()])
                                for i in range(n)])
        if ahead:
            assert tr.renders_ahead_used >= n - 3
    assert np.array_equal(hist[False], hist[True])


def test_a_queued_render_is_not_used_after_the_view_settings_changed(device):
    """The render queued behind frame N's LM launch baked in spp / render box / background as they were during
    frame N; if the user changes one before frame N+1, the tracker must render afresh (and count it), and the
    poses must be those of a tracker that never rendered ahead."""
    n = 8
    assets = make_tracking_assets(seed=1002, width=160, height=120, n_frames=n)
    hist, counters = {}, {}
    for ahead in (False, True):
        tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
        tr.render_ahead = ahead
        tr.spp = 4
        frames = render_query_frames(assets, tr.testbed)
        for i in range(n):
            if i == 4:
                tr.spp = 2  # between two steady frames
            if i == 6:
                tr.testbed.nerf.rendering_min_transmittance = 0.02
            tr.run_single_frame((f"{i:06d}.png", frames[i]))
        hist[ahead] = np.stack([np.concatenate([a.ravel() for a in tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()])
                                for i in range(n)])
        counters[ahead] = (tr.renders_ahead_used, tr.renders_ahead_stale, tr.renders_ahead_dropped)
    assert counters[False] == (0, 0, 0)
    assert counters[True][1] == 2 and counters[True][0] >= 2 and counters[True][2] == 0, counters
    assert np.array_equal(hist[False], hist[True])


def test_one_pipeline_render_with_the_box_filling_the_view(device):
    """A render below 2^19 rays runs as ONE pipeline over all rays: its buffers must hold every ray (a camera
    close to the box sees it in nearly every pixel; the first layout sized a pipeline for half the rays)."""
    from pixtrack_amd.synthetic import look_at_pose

    tb = Testbed(device=device)
    tb.load_snapshot(make_synthetic_nerf(11))
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    c = 0.5 * (lo + hi)
    eye = c + np.array([0.9, 0.5, 0.3]) / np.linalg.norm([0.9, 0.5, 0.3]) * 0.45
    R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
    tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
    W, H = 200, 150
    tb.fov = 80.0
    out = tb.render_device(W, H, 8, True, collect_stats=True)
    torch.cuda.synchronize()
    st = tb.read_stats()
    assert st["rays_hit"] > 0.75 * W * H * 8
    assert bool(torch.isfinite(out).all())


def _testbed(device, W, H):
    tb = Testbed(device=device)
    tb.load_snapshot(make_synthetic_nerf(11))
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = PREMIER_PROTEIN_AABB
    tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
    return tb


def test_render_frame_8bit_outputs_equal_the_separate_launches(device):
    """pxt_ngp_render_frame (round 4, between-stage fusion): the resolve kernel's own uint8 image and `!= 0` plane are
    bit for bit what rgba_to_u8 / depth_mask make of the float images of render_device / render_both_device - for
    Shade, Depth and both-in-one-march, with and without the float images alongside, odd sizes included."""
    from pixtrack_amd.ops import ops
    from pixtrack_amd.synthetic import look_at_pose
    from pixtrack_amd.visualization.run_vis_on_poses import rgba_to_u8

    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    c = 0.5 * (lo + hi)
    for (W, H), dist in (((160, 120), 1.7), ((203, 131), 1.2)):
        tb = _testbed(device, W, H)
        eye = c + np.array([0.9, 0.5, 0.3]) / np.linalg.norm([0.9, 0.5, 0.3]) * dist
        R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
        tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
        rgba, depth = tb.render_both_device(W, H, 4)
        want_u8 = rgba_to_u8(rgba, 0.0)
        want_mask = torch.empty(H, W, dtype=torch.uint8, device=device)
        ops.depth_mask(depth, 1, 5, want_mask, torch.empty(2 * H * W, dtype=torch.uint8, device=device))
        want_nz = ((depth[..., 0] * 255.0).to(torch.int64) & 255) != 0
        assert 0.02 < float(want_mask.float().mean()) < 0.98
        for want_float in (False, True):
            both = tb.render_frame_device(W, H, 4, mode=2, want_float=want_float)
            assert torch.equal(both["rgb_u8"], want_u8) and torch.equal(both["depth_nz"].bool(), want_nz)
            got_mask = torch.empty(H, W, dtype=torch.uint8, device=device)
            ops.depth_mask_plane(both["depth_nz"], 1, 5, got_mask)
            assert torch.equal(got_mask, want_mask)
            if want_float:
                assert torch.equal(both["rgba"], rgba) and torch.equal(both["depth"], depth)
            shade = tb.render_frame_device(W, H, 4, mode=0, want_float=want_float)
            assert torch.equal(shade["rgb_u8"], want_u8) and "depth_nz" not in shade
            dep = tb.render_frame_device(W, H, 4, mode=1, want_float=want_float)
            assert torch.equal(dep["depth_nz"].bool(), want_nz) and "rgb_u8" not in dep
            if want_float:
                assert torch.equal(shade["rgba"], rgba) and torch.equal(dep["rgba"], depth)
        # larger structuring elements take the pass-by-pass route: same mask as the float entry point
        m1 = torch.empty(H, W, dtype=torch.uint8, device=device)
        m2 = torch.empty(H, W, dtype=torch.uint8, device=device)
        ops.depth_mask(depth, 2, 8, m1, torch.empty(2 * H * W, dtype=torch.uint8, device=device))
        ops.depth_mask_plane(both["depth_nz"], 2, 8, m2)
        assert torch.equal(m1, m2)


def test_lm_epilogue_camera_equals_the_conversion_kernel(device):
    """pxt_lm_refine_cam: the camera the LM kernel's epilogue derives from its final pose (into the renderer's camera
    slot and a pinned record) has the bits of the one-thread conversion kernel of pxt_ngp_render_both_from_pose, and
    render_frame_device(from_slot=True) renders what that entry point renders."""
    from pixtrack_amd import _lib
    from pixtrack_amd.optimizer import LevelPack, PixTrackOptimizer, cstride_for
    from pixtrack_amd.synthetic import make_lm_scene
    from pixtrack_amd.visualization.run_vis_on_poses import rgba_to_u8

    W, H = 160, 120
    tb = _testbed(device, W, H)
    rng = np.random.default_rng(4)
    n2s = {"centroid": rng.normal(size=3) * 0.2, "avglen": 2.3, "R": np.eye(4), "totp": rng.normal(size=3) * 0.1}
    n2s["R"][:3, :3] = Rotation.random(random_state=3).as_matrix()
    conv = tb.pose_conversion(n2s)
    sc = make_lm_scene(seed=1007, width=160, height=120, n_points=600, sigma_px=2.0)
    packs = []
    for level in reversed(range(3)):
        fq = sc.feats_query[level]
        Cc = fq.shape[0] - 1
        fmap = torch.zeros(fq.shape[1], fq.shape[2], cstride_for(Cc))
        fmap[..., :Cc] = torch.nn.functional.normalize(fq[:-1], dim=0).permute(1, 2, 0)
        fmap[..., Cc] = fq[-1]
        fr = sc.feats_ref[level]
        fref = torch.zeros(fr.shape[0], cstride_for(Cc))
        fref[:, :Cc] = torch.nn.functional.normalize(fr[:, :-1], dim=1)
        fref[:, Cc] = fr[:, -1]
        packs.append(LevelPack(fmap.to(device).contiguous(), fref.to(device).contiguous(), Cc,
                               sc.camera.scale(sc.scales[level]), torch.full((6,), 1e-4)))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    opt = PixTrackOptimizer(dict(num_iters=150, pad=1))
    cam_rec = tb._next_cam_out()
    pend = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws,
                                           camera=(conv, [tb.camera_slot()], cam_rec))
    fused = tb.render_frame_device(W, H, 2, mode=2, from_slot=True)  # queued behind the launch, like the tracker's
    res = pend.result()
    assert not res.failed
    torch.cuda.synchronize()
    got = cam_rec.numpy().copy()
    assert got[12] == 1.0
    # the reference: the conversion kernel on the same pose record
    rgba, depth, cam_out = tb.render_both_from_pose_device(W, H, 2, pend.buf, conv)
    torch.cuda.synchronize()
    want = cam_out.numpy()
    assert want[12] == 1.0 and np.array_equal(got[:12].view(np.uint32), want[:12].view(np.uint32))
    assert torch.equal(fused["rgb_u8"], rgba_to_u8(rgba, 0.0))
    assert torch.equal(fused["depth_nz"].bool(), ((depth[..., 0] * 255.0).to(torch.int64) & 255) != 0)
    # a launch without camera outputs leaves the slot alone; no slot and no record is refused
    with pytest.raises(Exception):
        PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws, camera=(conv, [], None)).result()


@pytest.mark.parametrize("ahead", [True, False])
def test_tracking_with_the_pair_chain_equals_two_separate_renders(device, ahead, monkeypatch):
    """A frame's Depth + Shade pair as ONE chain (Testbed.render_frame_pair_device) against the same two renders made one
    after the other through the float-image entry points + rgba_to_u8 / the `!= 0` plane (what the tracker did before the
    renderer wrote the 8-bit planes itself): the same poses, bit for bit, over a tracked sequence."""
    from pixtrack_amd.visualization.run_vis_on_poses import rgba_to_u8

    n = 10
    assets = make_tracking_assets(seed=1002, width=320, height=240, n_frames=n)
    hist = {}
    for separate in (False, True):
        tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
        tr.fuse_identical_views = False
        tr.render_ahead = ahead and not separate  # (the separate renders take the host's camera)
        tb = tr.testbed
        if separate:
            def two_renders(depth_view, shade_view, spp=8, from_slot=False, workspace=None, tb=tb):
                assert not from_slot
                (dw, dh, dfov), (sw, sh, sfov) = depth_view, shade_view
                fov0, mode0 = tb.fov, tb.render_mode
                try:
                    tb.fov, tb.render_mode = dfov, tb.render_mode.Depth
                    depth = tb.render_device(dw, dh, spp)
                    tb.fov, tb.render_mode = sfov, tb.render_mode.Shade
                    rgba = tb.render_device(sw, sh, spp)
                finally:
                    tb.fov, tb.render_mode = fov0, mode0
                nz = (((depth[..., 0] * 255.0).to(torch.int64) & 255) != 0).to(torch.uint8)
                return nz, rgba_to_u8(rgba, 0.0)

            monkeypatch.setattr(tb, "render_frame_pair_device", two_renders)
        frames = render_query_frames(assets, tr.testbed)
        for i in range(n):
            tr.run_single_frame((f"{i:06d}.png", frames[i]))
        hist[separate] = np.stack(
            [np.concatenate([a.ravel() for a in tr.pose_history[f"{i:06d}.png"]["T_refined"].numpy()]) for i in range(n)])
        if tr.render_ahead:
            assert tr.renders_ahead_used >= n - 3, (tr.renders_ahead_used, tr.renders_ahead_rejected)
    assert np.array_equal(hist[False], hist[True])
