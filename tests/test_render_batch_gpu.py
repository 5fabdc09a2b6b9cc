"""pxt_ngp_render_frame_batch: K renders of K different renderer contexts - K objects tracked in lock-step, or a frame's Depth
(query camera) + Shade (reference camera) pair through a testbed's two contexts - as ONE staged chain of launches on one
stream.  Every image must be bit for bit the single render's: no ray's result depends on which rays share its launches."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from pixtrack_amd import _lib
from pixtrack_amd.ngp import Testbed
from pixtrack_amd.synthetic import PREMIER_PROTEIN_AABB, look_at_pose, make_synthetic_nerf

pytestmark = pytest.mark.gpu


def _testbed(device, seed, aabb, W, dist, direction):
    tb = Testbed(device=device)
    tb.load_snapshot(make_synthetic_nerf(seed))
    tb.background_color = [255, 255, 255, 0.0]
    tb.snap_to_pixel_centers = True
    tb.nerf.rendering_min_transmittance = 1e-7
    tb.render_aabb.min, tb.render_aabb.max = aabb
    tb.fov = math.degrees(2 * math.atan(W / (2 * 1.2 * W)))
    lo, hi = np.array(aabb)
    c = 0.5 * (lo + hi)
    d = np.asarray(direction, np.float64)
    eye = c + d / np.linalg.norm(d) * dist
    R, _ = look_at_pose(eye, c, up=np.array([0, 1.0, 0]))
    tb._cam_ngp = np.concatenate([R.T, eye[:, None]], 1)
    return tb


def _objects(device):
    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    c, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    slab = [list(c - half * np.array([1.0, 0.25, 1.0])), list(c + half * np.array([1.0, 0.25, 1.0]))]
    small = [list(c - half * 0.5), list(c + half * 0.5)]
    return [
        (_testbed(device, 11, PREMIER_PROTEIN_AABB, 160, 1.7, [0.9, 0.5, 0.3]), (160, 120)),
        (_testbed(device, 12, slab, 203, 1.3, [0.2, 0.9, 0.4]), (203, 131)),      # odd size, another NeRF, thin box
        (_testbed(device, 13, small, 96, 1.1, [-0.7, 0.1, 0.7]), (96, 64)),       # small render
        (_testbed(device, 11, PREMIER_PROTEIN_AABB, 160, 2.4, [0.1, 0.2, -1.0]), (160, 120)),
    ]


@pytest.mark.parametrize("mode", [2, 0, 1])
def test_batched_renders_equal_the_single_renders(device, mode):
    objs = _objects(device)
    want = [tb.render_frame_device(w, h, 4, mode=mode, pipelines=1) for tb, (w, h) in objs]
    ws = torch.empty(Testbed.batch_workspace_bytes(len(objs)), dtype=torch.uint8, device=device)
    for rep in range(2):  # (the second pass starts from counters the batch's own resolve kernel zeroed)
        got = Testbed.render_frame_batch_device([tb for tb, _ in objs], [s for _, s in objs], 4, mode=mode, workspace=ws)
        torch.cuda.synchronize()
        for k, (g, w_) in enumerate(zip(got, want)):
            assert set(g) == set(w_)
            for key in g:
                assert torch.equal(g[key], w_[key]), (mode, rep, k, key)
    if mode != 1:
        assert all(0.01 < float((w_["rgb_u8"].float().mean(-1) > 5).float().mean()) < 0.99 for w_ in want)  # (an object on black)
    # a single render after the batch (two pipelines again) still sees clean counters
    tb, (w, h) = objs[0]
    again = tb.render_frame_device(w, h, 4, mode=mode)
    for key in again:
        assert torch.equal(again[key], want[0][key])


def test_batched_render_full_size_and_stats(device):
    """640 x 480 x 8 spp (two-pipeline size for a single render): image and sample counts equal the single renders'."""
    lo, hi = np.array(PREMIER_PROTEIN_AABB)
    objs = [(_testbed(device, 21 + k, PREMIER_PROTEIN_AABB, 640, 1.6 + 0.2 * k, d), (640, 480))
            for k, d in enumerate(([0.9, 0.5, 0.3], [0.1, 0.3, 1.0], [-0.8, 0.2, 0.1]))]
    want, want_stats = [], []
    for tb, (w, h) in objs:
        tb.stats_accum = torch.zeros(4, dtype=torch.int64, device=device)
        want.append(tb.render_frame_device(w, h, 8, mode=2))
        want_stats.append(tb.stats_accum.clone())
        tb.stats_accum.zero_()
    got = Testbed.render_frame_batch_device([tb for tb, _ in objs], [s for _, s in objs], 8, mode=2)
    torch.cuda.synchronize()
    for k, (g, w_) in enumerate(zip(got, want)):
        assert torch.equal(g["rgb_u8"], w_["rgb_u8"]) and torch.equal(g["depth_nz"], w_["depth_nz"]), k
        # samples composited and rays that hit the box: equal.  (Counters 2 and 3 - rays / samples left to the straggler
        # kernel - follow the number of wavefront rounds, which differs between the batched chain and a single render.)
        assert torch.equal(objs[k][0].stats_accum[:2], want_stats[k][:2]), (objs[k][0].stats_accum, want_stats[k])
        assert int(want_stats[k][0]) > 100000


def test_a_batch_of_one_is_the_single_render(device):
    tb, (w, h) = _objects(device)[1]
    want = tb.render_frame_device(w, h, 4, mode=2)
    got = Testbed.render_frame_batch_device([tb], [(w, h)], 4, mode=2)[0]
    assert torch.equal(got["rgb_u8"], want["rgb_u8"]) and torch.equal(got["depth_nz"], want["depth_nz"])


@pytest.mark.parametrize("sizes", [((160, 120), (240, 180)), ((640, 480), (960, 720)), ((203, 131), (96, 64))])
def test_a_frames_depth_and_shade_pair_equals_the_two_single_renders(device, sizes):
    """Modes {1, 0} with different sizes and focal lengths at ONE pose - the mask's Depth render at the query camera and the
    reference image's Shade render at the reference camera (reference pixloc_tracker_r9.py:145-152, 207-214) - through
    render_frame_pair_device: one chain, the second render through the testbed's second context (shared tables)."""
    (dw, dh), (sw, sh) = sizes
    tb = _testbed(device, 11, PREMIER_PROTEIN_AABB, dw, 1.7, [0.9, 0.5, 0.3])
    fov_d = math.degrees(2 * math.atan(dw / (2 * 1.2 * dw)))
    fov_s = math.degrees(2 * math.atan(sw / (2 * 1.05 * sw)))
    spp = 8 if dw >= 640 else 4
    tb.fov = fov_d
    want_nz = tb.render_frame_device(dw, dh, spp, mode=1)["depth_nz"]
    tb.fov = fov_s
    want_u8 = tb.render_frame_device(sw, sh, spp, mode=0)["rgb_u8"]
    tb.fov = 33.0  # (the pair takes its fields of view from its arguments)
    for rep in range(2):
        nz, u8 = tb.render_frame_pair_device((dw, dh, fov_d), (sw, sh, fov_s), spp)
        torch.cuda.synchronize()
        assert torch.equal(nz, want_nz) and torch.equal(u8, want_u8), rep
    assert 0.01 < float(want_nz.float().mean()) < 0.99
    # the camera slots: a pair queued with from_slot reads each context's own slot
    cam = torch.from_numpy(np.asarray(tb._cam_ngp, np.float32).reshape(-1)).to(device)
    for side in (False, True):
        slot = tb.camera_slot(side=side)
        # (in the product the LM kernel's epilogue fills the slots; here a device-to-device copy does)
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(C.c_void_p(slot), C.c_void_p(cam.data_ptr()), 48, 3) == 0
    tb._cam_ngp = np.eye(4)[:3]  # the host-side camera must NOT be what the slot render uses
    nz, u8 = tb.render_frame_pair_device((dw, dh, fov_d), (sw, sh, fov_s), spp, from_slot=True)
    assert torch.equal(nz, want_nz) and torch.equal(u8, want_u8)


def test_batch_argument_errors(device):
    objs = _objects(device)[:2]
    tbs, sizes = [tb for tb, _ in objs], [s for _, s in objs]
    with pytest.raises(_lib.PxtError):  # the same context twice: one set of ray lists
        Testbed.render_frame_batch_device([tbs[0], tbs[0]], sizes, 4)
    three = _objects(device)[:3]
    with pytest.raises(_lib.PxtError):  # workspace too small (three renders: the records live in device memory)
        Testbed.render_frame_batch_device([tb for tb, _ in three], [s for _, s in three], 4,
                                          workspace=torch.empty(64, dtype=torch.uint8, device=device))
    with pytest.raises(_lib.PxtError):  # a mode per render
        Testbed.render_frame_batch_device(tbs, sizes, 4, mode=[2, 3])
    L = _lib.lib()
    assert L.pxt_ngp_batch_workspace_bytes(0) < 0 and L.pxt_ngp_batch_workspace_bytes(_lib.PXT_NGP_MAX_BATCH + 1) < 0
    assert L.pxt_ngp_render_frame_batch(None, None, 1, None, 0, None, None, None, None) == -1
    assert L.pxt_ngp_create_shared(None, None) == -1
