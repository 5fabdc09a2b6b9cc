"""Parity of the fused HIP LM kernel (through the C ABI) against the CPU oracle.

Tolerance (BASELINE.json north_star): final pose within 1e-3 rad and 1e-3 (scene units,
"1 mm") of the oracle on identical inputs.  Iteration counts are NOT compared: fp32
reduction order differs, so a stop test may fire one iteration apart.
"""
import numpy as np
import pytest
import torch

from oracle import lm_oracle as O
from pixtrack_amd import _lib
from pixtrack_amd.geometry import Camera, Pose
from pixtrack_amd.optimizer import LevelPack, PixTrackOptimizer, cstride_for
from pixtrack_amd.synthetic import make_lm_scene

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-3, 1e-3


def pack_level(scene, level, device, cam_full):
    """What the refiner's fast path builds: normalised HWC query map + normalised refs."""
    fq = scene.feats_query[level]
    Cc = fq.shape[0] - 1
    cs = cstride_for(Cc)
    h, w = fq.shape[1:]
    fmap = torch.zeros(h, w, cs)
    fmap[..., :Cc] = O.l2_normalize(fq[:-1], dim=0).permute(1, 2, 0)
    fmap[..., Cc] = fq[-1]
    fr = scene.feats_ref[level]
    fref = torch.zeros(fr.shape[0], cs)
    fref[:, :Cc] = O.l2_normalize(fr[:, :-1], dim=1)
    fref[:, Cc] = fr[:, -1]
    return fmap.to(device).contiguous(), fref.to(device).contiguous(), Cc, cam_full.scale(scene.scales[level])


def lambdas(consts):
    return [O.damping_lambda(torch.as_tensor(c, dtype=torch.float32)) for c in consts]


CONSTS = [[-2.0] * 6, [-1.5, -2.5, -2.0, -1.8, -2.2, -2.0], [-2.0, -2.0, -1.0, -3.0, -2.0, -1.5]]


@pytest.mark.parametrize("seed,size,npts,k1", [(1001, (320, 240), 2048, 0.0), (1002, (256, 192), 777, -0.05),
                                               (1003, (640, 480), 2048, 0.0)])
def test_refine_matches_oracle(device, seed, size, npts, k1):
    sc = make_lm_scene(seed=seed, width=size[0], height=size[1], n_points=npts, sigma_px=2.0, k1=k1)
    lam = lambdas(CONSTS)
    conf = O.LMConf()
    log = O.LMLog()
    ref = O.refine_pose_using_features(
        sc.feats_query, sc.scales, sc.camera._data, torch.from_numpy(sc.R_init), torch.from_numpy(sc.t_init),
        sc.feats_ref, torch.from_numpy(sc.p3d), lam, conf, log=log)
    assert ref["success"]

    opt = PixTrackOptimizer(dict(num_iters=conf.num_iters, pad=conf.pad))
    packs = []
    for level in reversed(range(3)):  # coarse -> fine, optimizer[level]
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    res = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws).result()
    assert not res.failed
    R, t = res.T.R.double(), res.T.t.double()
    rot = O.rotation_angle_rad(R, ref["R"])
    trans = float((t - ref["t"]).norm())
    assert rot < ROT_TOL and trans < TRANS_TOL, (rot, trans, res.iters, log.num_iters)
    # the first logged cost of the first level is evaluated at the identical pose
    assert res.costs[0][0] == pytest.approx(log.costs[0][0], rel=2e-4)
    # and the optimiser actually converged towards the ground truth
    assert O.rotation_angle_rad(R, torch.from_numpy(sc.R_gt)) < 5e-3
    # deterministic: bit-identical on a second run
    res2 = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws).result()
    assert torch.equal(res.T.as12(), res2.T.as12())


@pytest.mark.parametrize("grid", [1, 8, 64, 200])
def test_grid_size_invariance(device, grid):
    """Any persistent grid size must give the same pose to fp32 reduction noise."""
    sc = make_lm_scene(seed=1004, width=320, height=240, n_points=1000, sigma_px=2.0)
    lam = lambdas(CONSTS)
    opt = PixTrackOptimizer(dict(num_iters=150, pad=1, n_workgroups=grid))
    packs = []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    res = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws).result()
    ref = O.refine_pose_using_features(
        sc.feats_query, sc.scales, sc.camera._data, torch.from_numpy(sc.R_init), torch.from_numpy(sc.t_init),
        sc.feats_ref, torch.from_numpy(sc.p3d), lam, O.LMConf())
    assert O.rotation_angle_rad(res.T.R.double(), ref["R"]) < ROT_TOL
    assert float((res.T.t.double() - ref["t"]).norm()) < TRANS_TOL


def test_single_level_run_pixloc_convention(device):
    """opt.run(p3d, F_ref, F_q, T, camera, W_ref_query=...) -> (T, failed): one level."""
    sc = make_lm_scene(seed=1005, width=320, height=240, n_points=1500, sigma_px=2.0)
    level = 1
    fq = sc.feats_query[level]
    F_q = O.l2_normalize(fq[:-1], dim=0)
    W_q = fq[-1:]
    fr = sc.feats_ref[level]
    F_ref, W_ref = O.l2_normalize(fr[:, :-1], dim=1), fr[:, -1:]
    lam = O.damping_lambda(torch.full((6,), -2.0))
    cam = O.camera_scale(sc.camera._data, sc.scales[level])
    log = O.LMLog()
    R, t, failed = O.lm_run(torch.from_numpy(sc.p3d).float(), F_ref, F_q, torch.from_numpy(sc.R_init).float(),
                            torch.from_numpy(sc.t_init).float(), cam, W_ref, W_q, lam, O.LMConf(), log=log)
    opt = PixTrackOptimizer(dict(num_iters=150, pad=1))
    opt.dampingnet.const = torch.full((6,), -2.0)
    seen = []
    opt.logging_fn = lambda **kw: seen.append((kw["i"], float((kw["valid"] * kw["cost"]).sum() / kw["valid"].sum())))
    T, fail = opt.run(sc.p3d, F_ref.to(device), F_q.to(device), sc.T_init.to(device).float(),
                      sc.camera.scale(sc.scales[level]).to(device), W_ref_query=(W_ref.to(device), W_q.to(device)))
    assert not bool(fail) and not failed
    assert O.rotation_angle_rad(T.R.cpu().double(), R.double()) < ROT_TOL
    assert float((T.t.cpu().double() - t.double()).norm()) < TRANS_TOL
    assert seen[0][0] == 0 and seen[0][1] == pytest.approx(log.costs[0][0], rel=2e-4)
    assert [i for i, _ in seen] == list(range(len(seen)))


def test_too_few_points_reports_failed(device):
    """< 10 valid points is an algorithmic failure (failed=True), not an exception."""
    sc = make_lm_scene(seed=1006, width=160, height=120, n_points=64, sigma_px=2.0)
    lam = lambdas(CONSTS)
    packs = []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    mask = torch.zeros(64, dtype=torch.uint8, device=device)
    mask[:5] = 1
    opt = PixTrackOptimizer(dict(num_iters=150, pad=1))
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    res = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws, mask=mask).result()
    assert res.failed
    assert res.iters[0] == 1 and res.iters[1] == 0
    # pose untouched by the masked (identity) step
    assert torch.allclose(res.T.as12(), sc.T_init.as12().float(), atol=1e-6)


def test_sample_sparse_matches_oracle(device):
    import ctypes as C

    sc = make_lm_scene(seed=1007, width=320, height=240, n_points=1200, sigma_px=2.0, k1=-0.02)
    # push some points out of view so the validity logic is exercised
    p3d = sc.p3d.copy()
    p3d[:50] += 10.0
    ref_scale, pad = 0.5, 1
    ref_cam = O.camera_scale(sc.camera._data, 2.0)  # "SfM camera" at 2x the map resolution
    obs_ref, valid_ref = O.interp_sparse_observations(
        sc.feats_query, sc.scales, ref_cam, ref_scale, torch.from_numpy(sc.R_gt).float(),
        torch.from_numpy(sc.t_gt).float(), torch.from_numpy(p3d).float(), pad)
    L = _lib.lib()
    n = p3d.shape[0]
    arr = (_lib.SampleLevel * 3)()
    outs, maps = [], []
    cam_full = Camera(ref_cam).scale(ref_scale)
    for l in range(3):
        fq = sc.feats_query[l]
        Cc = fq.shape[0] - 1
        cs = cstride_for(Cc)
        fmap = torch.zeros(fq.shape[1], fq.shape[2], cs)
        fmap[..., : Cc + 1] = fq.permute(1, 2, 0)
        fmap = fmap.to(device).contiguous()
        out = torch.full((n, cs), -7.0, device=device)
        maps.append(fmap)
        outs.append(out)
        arr[l].fmap, arr[l].out = fmap.data_ptr(), out.data_ptr()
        arr[l].h, arr[l].w, arr[l].C, arr[l].cstride = fq.shape[1], fq.shape[2], Cc, cs
        arr[l].cam[:] = cam_full.scale(sc.scales[l]).as10().tolist()
        arr[l].ndist = 2
    T = _lib.host_pose12(Pose.from_Rt(sc.R_gt, sc.t_gt))
    valid = torch.zeros(n, dtype=torch.uint8, device=device)
    pd = torch.from_numpy(p3d).float().to(device)
    _lib.check(L.pxt_sample_sparse(pd.data_ptr(), n, T, arr, 3, pad, 1, valid.data_ptr(),
                                   _lib.stream_ptr(device)), "pxt_sample_sparse")
    torch.cuda.synchronize()
    v = valid.cpu().bool()
    assert torch.equal(v, valid_ref)
    assert 0 < int(v.sum()) < n
    for l in range(3):
        Cc = sc.feats_query[l].shape[0] - 1
        got = outs[l].cpu()
        exp_f = O.l2_normalize(obs_ref[l][:, :Cc], dim=1)
        assert torch.allclose(got[v, :Cc], exp_f[v], atol=2e-5)
        assert torch.allclose(got[v, Cc], obs_ref[l][v, Cc], atol=2e-5)


def test_config5_sized_problem(device):
    """BASELINE configs[4] sizes: N = 10 000 points against 1024x576 maps (a 1920x1080 query after
    the extractor's resize): same parity bar, and every workgroup count stays deterministic."""
    sc = make_lm_scene(seed=1008, width=1024, height=576, n_points=10000, sigma_px=2.0)
    lam = lambdas(CONSTS)
    ref = O.refine_pose_using_features(
        sc.feats_query, sc.scales, sc.camera._data, torch.from_numpy(sc.R_init), torch.from_numpy(sc.t_init),
        sc.feats_ref, torch.from_numpy(sc.p3d), lam, O.LMConf())
    opt = PixTrackOptimizer(dict(num_iters=150, pad=1))
    packs = []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    res = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws).result()
    assert ref["success"] and not res.failed
    assert O.rotation_angle_rad(res.T.R.double(), ref["R"]) < ROT_TOL
    assert float((res.T.t.double() - ref["t"]).norm()) < TRANS_TOL


def test_hd_four_stage_plan_matches_oracle(device):
    """BASELINE configs[4]'s LM work on its own: the 4-stage plan {image scale 4: level [2]; image scale 1: levels
    [2, 1, 0]} (bench.py --config hd; builder-defined, the reference has 3 levels per image scale) at N = 10 000
    points against the pyramids of a 1920x1080 query after the extractor's resize (1024x576, and 256x144 at image
    scale 4), HIP kernel vs the oracle's loop over the same level lists, stage by stage."""
    s4 = make_lm_scene(seed=1009, width=256, height=144, n_points=10000, sigma_px=2.0)
    s1 = make_lm_scene(seed=1009, width=1024, height=576, n_points=10000, sigma_px=2.0)
    assert np.array_equal(s4.p3d, s1.p3d) and np.array_equal(s4.R_init, s1.R_init) and np.array_equal(s4.t_gt, s1.t_gt)
    lam = lambdas(CONSTS)
    conf = O.LMConf()
    p3d_o = torch.from_numpy(s1.p3d)
    log = O.LMLog()
    ra = O.refine_pose_using_features(s4.feats_query, s4.scales, s4.camera._data, torch.from_numpy(s4.R_init),
                                      torch.from_numpy(s4.t_init), s4.feats_ref, p3d_o, lam, conf, log=log, levels=[2])
    assert ra["success"]
    rb = O.refine_pose_using_features(s1.feats_query, s1.scales, s1.camera._data, ra["R"], ra["t"], s1.feats_ref, p3d_o,
                                      lam, conf, log=log, levels=[2, 1, 0])
    assert rb["success"] and len(log.num_iters) == 4

    opt = PixTrackOptimizer(dict(num_iters=conf.num_iters, pad=conf.pad))
    p3d = torch.from_numpy(s1.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)

    def packs_of(sc, levels):
        out = []
        for level in levels:
            fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
            out.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
        return out

    res_a = PixTrackOptimizer.refine_levels(p3d, packs_of(s4, [2]), s4.T_init, opt.native_conf(), ws).result()
    assert not res_a.failed and len([n for n in res_a.iters if n]) == 1
    assert O.rotation_angle_rad(res_a.T.R.double(), ra["R"]) < ROT_TOL
    assert float((res_a.T.t.double() - ra["t"]).norm()) < TRANS_TOL
    res_b = PixTrackOptimizer.refine_levels(p3d, packs_of(s1, [2, 1, 0]), res_a.T, opt.native_conf(), ws).result()
    assert not res_b.failed and len([n for n in res_b.iters if n]) == 3
    rot = O.rotation_angle_rad(res_b.T.R.double(), rb["R"])
    tra = float((res_b.T.t.double() - rb["t"]).norm())
    assert rot < ROT_TOL and tra < TRANS_TOL, (rot, tra, res_a.iters, res_b.iters, log.num_iters)
    assert O.rotation_angle_rad(res_b.T.R.double(), torch.from_numpy(s1.R_gt)) < 5e-3  # and towards the ground truth


def test_dirty_workspace_and_back_to_back_launches(device):
    """The inter-workgroup exchange polls tagged granules in the workspace: every polled word is zeroed by the
    launch's own memset, so a workspace full of garbage (or of the previous launch's tags) changes nothing, and
    two refinements enqueued back to back on one workspace give the bits of two separate ones."""
    sc = make_lm_scene(seed=1006, width=320, height=240, n_points=1200, sigma_px=2.0)
    lam = lambdas(CONSTS)
    opt = PixTrackOptimizer(dict(num_iters=150, pad=1))
    packs = []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    nbytes = int(_lib.lib().pxt_lm_workspace_bytes())
    clean = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    want = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), clean).result()
    dirty = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=device)
    got = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), dirty).result()
    assert torch.equal(got.T.as12(), want.T.as12()) and got.iters == want.iters
    first = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), dirty)
    second = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), dirty)
    r1, r2 = first.result(), second.result()
    assert torch.equal(r1.T.as12(), want.T.as12()) and torch.equal(r2.T.as12(), want.T.as12())


def test_forced_timeout_then_workspace_reuse(device):
    """a launch that times out must not poison the next launch on the same workspace.  A spin
    bound of ONE poll makes the inter-workgroup wait of a 128-workgroup grid give up (PXT_E_TIMEOUT -> PxtError);
    workgroups that lagged may still publish granules of an epoch the others never finished.  The kernel advances its
    tag base by a margin on that path and the host zeroes the workspace: the next launch must return bit for bit what
    a fresh workspace returns - also when the host's zeroing is bypassed (raw op call on the poisoned workspace)."""
    from pixtrack_amd.ops import ops

    sc = make_lm_scene(seed=1006, width=320, height=240, n_points=2048, sigma_px=2.0)
    lam = lambdas(CONSTS)
    packs = []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    nbytes = int(_lib.lib().pxt_lm_workspace_bytes())
    good = PixTrackOptimizer(dict(num_iters=150, pad=1))
    fresh = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, good.native_conf(),
                                            torch.zeros(nbytes, dtype=torch.uint8, device=device)).result()
    assert not fresh.failed
    bad = PixTrackOptimizer(dict(num_iters=150, pad=1, spin_limit=1))
    for zero_on_host in (True, False):
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        # a few healthy launches first, so that the tag base is not 0 and both granule areas hold old epochs
        for _ in range(3):
            PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, good.native_conf(), ws).result()
        timed_out = 0
        for _ in range(4):
            pend = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, bad.native_conf(), ws)
            if not zero_on_host:
                pend._keep = None  # (PendingLM.result() then has no workspace to zero: the kernel's margin alone)
            try:
                pend.result()
            except _lib.PxtError as e:
                assert "status" in str(e)
                timed_out += 1
            torch.cuda.synchronize()
        assert timed_out >= 1, "a one-poll spin bound did not time out on a 128-workgroup grid"
        again = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, good.native_conf(), ws).result()
        assert torch.equal(again.T.as12(), fresh.T.as12()) and again.iters == fresh.iters


@pytest.mark.parametrize("npts", [700, 2341, 3000, 5000])
def test_one_round_levels_equal_the_general_path(device, npts):
    """Round 4: a level whose points fit the grid's lane groups keeps them in registers (32-, 16- or 8-lane groups by
    point count: 700 / 2341 / 3000 / 5000 points at 128 workgroups take the 32-, 16-, 16- and 8-lane variants on the
    C = 128 levels); `lm_path = 2` forces the general several-rounds path.  Same sums in another association: the two
    must agree to fp32 reduction noise, far inside the 1e-3 parity bar, and both must be repeatable bit for bit."""
    sc = make_lm_scene(seed=1008, width=320, height=240, n_points=npts, sigma_px=2.0)
    lam = lambdas(CONSTS)
    packs = []
    for level in reversed(range(3)):
        fmap, fref, Cc, cam = pack_level(sc, level, device, sc.camera)
        packs.append(LevelPack(fmap, fref, Cc, cam, lam[level]))
    p3d = torch.from_numpy(sc.p3d).float().to(device)
    ws = torch.zeros(int(_lib.lib().pxt_lm_workspace_bytes()), dtype=torch.uint8, device=device)
    res = {}
    for path in (0, 2):
        opt = PixTrackOptimizer(dict(num_iters=150, pad=1, lm_path=path))
        a = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws).result()
        b = PixTrackOptimizer.refine_levels(p3d, packs, sc.T_init, opt.native_conf(), ws).result()
        assert not a.failed and torch.equal(a.T.as12(), b.T.as12())
        res[path] = a
    # (entry-wise: acos of a float32 rotation's trace resolves no angle below ~4e-4 rad)
    dR = float((res[0].T.R - res[2].T.R).abs().max())
    assert dR < 1e-5 and float((res[0].T.t - res[2].T.t).norm()) < 1e-5, (dR, res[0].iters, res[2].iters)
    assert res[0].iters == res[2].iters
    assert res[0].costs[0][0] == pytest.approx(res[2].costs[0][0], rel=1e-5)
    assert O.rotation_angle_rad(res[0].T.R.double(), torch.from_numpy(sc.R_gt)) < 5e-3
