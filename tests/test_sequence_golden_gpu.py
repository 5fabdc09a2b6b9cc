"""FIVE consecutive frames of BASELINE configs[1] at full size (640x480, spp 8) through
PixLocPoseTrackerR9.run_single_frame - with the next frame's render queued behind the LM launch, as the
product runs - against the CPU oracle's restatement of the per-frame policy
(oracle/frame_oracle.track_sequence; vectors in tests/golden/sequence_640x480.npz, generator
scripts/make_sequence_golden.py; reference pixtrack/pose_trackers/pixloc_tracker_r9.py:216-275).

What only exists ACROSS frames and is pinned here at full size: the cost threshold frozen from frame 0,
the mask rendered from the previous ACCEPTED pose, a frame rejected by the cost gate leaving the pose untouched
and dropping the success flag, the frame after it running unmasked from the old pose, and the queued render being
consumed on accepted frames / discarded after the rejected one.
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from pixtrack_amd.pose_trackers.pixloc_tracker_r9 import PixLocPoseTrackerR9
from pixtrack_amd.synthetic import make_tracking_assets
from pixtrack_amd.utils.pose_utils import geodesic_distance_for_rotations

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden" / "sequence_640x480.npz"
ROT_TOL, TRANS_TOL = 1e-3, 1e-3


@pytest.mark.parametrize("render_ahead", [True, False])
def test_five_frame_sequence_matches_the_oracle_policy(device, render_ahead):
    g = np.load(GOLDEN)
    n, bad = int(g["n_frames"]), int(g["bad_frame"])
    assert (int(g["width"]), int(g["height"]), int(g["spp"]), n) == (640, 480, 8, 5)
    assets = make_tracking_assets(seed=int(g["seed"]), width=640, height=480, n_frames=n)
    assert np.array_equal(np.stack([p[0] for p in assets["gt_poses"]]), g["gt_R"])  # seeded generator reproduced
    tr = PixLocPoseTrackerR9("", "", "", "/tmp", debug=0, device=device, assets=assets)
    tr.render_ahead = render_ahead
    masks = {}
    for i in range(n):
        q = torch.from_numpy(g["queries"][i].astype(np.float32)).to(device)
        R_before = None if tr.pose is None else tr.pose.numpy()[0].copy()
        if i > 0:  # the frame starts where the oracle's did: the pose carried over is the last ACCEPTED one
            Rs, ts = tr.pose.numpy()
            assert geodesic_distance_for_rotations(Rs, g[f"f{i}_R_start"]) < ROT_TOL, i
            assert float(np.linalg.norm(ts - g[f"f{i}_t_start"])) < TRANS_TOL, i
        tr.run_single_frame((f"{i:06d}.png", q))
        ok = tr.success  # the gated decision (refine()'s return value; run_single_frame returns nothing, :21-37)
        ret = tr.pose_history[f"{i:06d}.png"]
        want_ok = bool(g[f"f{i}_success"])
        # (ret["success"] is the LM's own flag, as in the reference's pose_history; the gated decision is the return value)
        assert bool(ok) == want_ok, (i, ok, ret["cost"], float(g[f"f{i}_cost"]))
        assert bool(ret["success"]) == bool(g[f"f{i}_lm_success"]), i
        m = tr.localizer.refiner.query_mask
        assert (m is not None) == bool(g[f"f{i}_masked"]), i
        if m is not None:
            masks[i] = np.packbits((m != 0).cpu().numpy().astype(np.uint8))
        assert list(tr.localizer.refiner.conf.multiscale) == [int(v) for v in g[f"f{i}_multiscale"]], i
        if i == 0:
            assert tr.cost_threshold == pytest.approx(float(g["f0_cost_threshold"]), rel=0.03)
        if i == bad:
            # the LM itself ran to the end; the COST gate rejected the frame (:251-263)
            assert bool(g[f"f{i}_lm_success"]) and ret["cost"] > tr.cost_threshold
            assert np.array_equal(tr.pose.numpy()[0], R_before) and tr.success is False
        else:
            assert ret["cost"] == pytest.approx(float(g[f"f{i}_cost"]), rel=0.05), i
        if want_ok:
            Rr, tt = ret["T_refined"].numpy()
            rot = geodesic_distance_for_rotations(Rr, g[f"f{i}_R"])
            tra = float(np.linalg.norm(tt - g[f"f{i}_t"]))
            assert rot < ROT_TOL and tra < TRANS_TOL, (i, rot, tra)
    # masks are byte work: bit-exact wherever the oracle's own depth decision is not within 0.05 grey levels of
    # flipping (frames whose fixture counts fragile pixels are compared on the pre-morphology bits elsewhere)
    for i, bits in masks.items():
        want = g[f"f{i}_mask_bits"]
        if int(g[f"f{i}_depth_fragile_count"]) == 0:
            assert np.array_equal(bits, want), i
        else:
            diff = np.unpackbits(bits ^ want).sum()
            assert diff <= 25 * 121 * int(g[f"f{i}_depth_fragile_count"]), (i, int(diff))  # a flipped bit dilates to <= 21 x 21
    if render_ahead:
        # queued renders: consumed after the accepted steady frames, none after the rejected one
        assert tr.renders_ahead_used >= 1 and tr.renders_ahead_dropped == 0 and tr.renders_ahead_stale == 0
    else:
        assert tr.renders_ahead_used == 0
