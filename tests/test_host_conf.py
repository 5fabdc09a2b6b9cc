"""Host-side decisions that have no kernel in them: the optimizer's `pad` resolution (ADVICE r1 /
DESIGN section 4), the stop rule on plain numbers, the DebugTracker record (trackers.pkl schema),
the Interpolator's refusal to invent gradients, pixloc's resize factor."""
import pickle

import numpy as np
import pytest
import torch

from pixtrack_amd.feature_extractor import PixTrackFeatureExtractor
from pixtrack_amd.geometry import Pose
from pixtrack_amd.optimizer import Interpolator, PixTrackOptimizer
from pixtrack_amd.tracker import DebugTracker


def test_top_level_pad_is_promoted_to_the_interpolation_border():
    # pixloc_tracker_r9.py:48 passes optimizer = {num_iters: 150, pad: 1}
    opt = PixTrackOptimizer({"num_iters": 150, "pad": 1})
    assert opt.interpolator.pad == 1 and opt.native_conf().pad == 1 and opt.native_conf().num_iters == 150
    # pixloc's own default (interpolation.pad = 4) applies when the caller says nothing ...
    assert PixTrackOptimizer({}).interpolator.pad == 4
    # ... and an explicit interpolation.pad is honoured when no top-level key is given
    assert PixTrackOptimizer({"interpolation": {"pad": 2}}).native_conf().pad == 2
    # the top-level key wins over the nested default, as the back-compat promotion does upstream
    assert PixTrackOptimizer({"pad": 1, "interpolation": {"mode": "linear"}}).interpolator.pad == 1


def test_stop_rule_on_numbers_and_on_poses():
    opt = PixTrackOptimizer({"pad": 1})
    c = opt.conf
    assert opt.converged(c.dR_stop_criteria * 0.5, c.dt_stop_criteria * 0.5, 1.0)         # tiny step
    assert not opt.converged(c.dR_stop_criteria * 2.0, c.dt_stop_criteria * 0.5, 1.0)     # rotation still moving
    assert not opt.converged(c.dR_stop_criteria * 0.5, c.dt_stop_criteria * 2.0, 1.0)     # translation still moving
    assert opt.converged(10.0, 10.0, c.grad_stop_criteria * 0.5)                         # flat gradient
    # a batch stops only when every element does
    assert not opt.converged(torch.tensor([0.0, 1.0]), torch.tensor([0.0, 1.0]), torch.tensor([1.0, 1.0]))
    big = Pose.from_aa(torch.tensor([[0.0, 0.0, 0.02]]), torch.tensor([[0.0, 0.0, 0.0]]))   # 1.1 degrees
    tiny = Pose.from_aa(torch.tensor([[0.0, 0.0, 1e-5]]), torch.tensor([[1e-4, 0.0, 0.0]]))
    g = torch.ones(1, 6)
    assert not opt.early_stop(i=3, T_delta=big, grad=g)
    assert opt.early_stop(i=3, T_delta=tiny, grad=g)
    assert opt.early_stop(i=0, T_delta=big, grad=g * 1e-6)
    opt.training = True
    assert not opt.early_stop(i=3, T_delta=tiny, grad=g)


class _Refiner:
    def __init__(self):
        self.optimizer = [PixTrackOptimizer({"pad": 1}) for _ in range(3)]


def test_debug_tracker_bulk_record_equals_the_per_iteration_hook():
    rng = np.random.default_rng(0)
    T0 = Pose.from_Rt(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    poses, costs, steps = [], [], []
    T = T0
    for i in range(4):
        d = Pose.from_aa(torch.from_numpy(rng.normal(size=3) * 1e-2).float(), torch.from_numpy(rng.normal(size=3) * 1e-2).float())
        T = d @ T
        poses.append(T.as12().reshape(12).clone())
        costs.append(float(rng.uniform()))
        steps.append(float(d.magnitude()[1]))
    poses12 = torch.stack(poses)
    a, b = DebugTracker(_Refiner(), 1), DebugTracker(_Refiner(), 1)
    a.record_level(T0, costs, poses12, steps)
    Tp = T0
    for i in range(4):
        Ti = Pose(poses12[i].clone())
        b.log_optim_iter(i=i, T_init=T0, T=Ti, T_delta=Ti @ Tp.inv(), cost=torch.tensor([[costs[i]]]), valid=torch.ones(1, 1))
        Tp = Ti
    assert a.num_iters == b.num_iters == [4]
    assert np.allclose(np.ravel(a.costs[0]), np.ravel(b.costs[0]))
    assert len(a.T) == len(b.T) == 5
    assert np.allclose(np.ravel(a.dt), np.ravel(b.dt), atol=1e-6)
    # same element layout whichever path filled the record (trackers.pkl must not depend on it)
    assert {np.shape(x) for x in a.costs[0]} == {np.shape(x) for x in b.costs[0]} == {(1,)}
    assert {np.shape(x) for x in a.dt} == {np.shape(x) for x in b.dt} == {(1,)}
    # the refiner's optimizers report to the LAST tracker attached (pixloc BaseTracker behaviour)
    ref = _Refiner()
    t1 = DebugTracker(ref, 1)
    t2 = DebugTracker(ref, 0)
    assert ref.tracker is t2 and all(o.level_logging_fn == t2.record_level for o in ref.optimizer)
    t2.record_level(T0, costs, poses12, steps)
    assert t2.costs == [] and t1.costs == []  # debug 0 keeps nothing
    # trackers.pkl payload: no refiner inside, the reference's attribute names present
    blob = pickle.loads(pickle.dumps(a))
    assert not hasattr(blob, "refiner")
    for k in ("costs", "T", "dt", "num_iters", "dense", "p3d", "p3d_ids", "debug"):
        assert hasattr(blob, k)


def test_interpolator_never_returns_made_up_gradients():
    interp = Interpolator("linear", 1)
    with pytest.raises(NotImplementedError):
        interp(torch.zeros(4, 8, 8), torch.zeros(3, 2), return_gradients=True)


def test_resize_factor_is_the_unrounded_one_pixloc_returns():
    class _M:
        scales = [1, 4, 16]

    fe = PixTrackFeatureExtractor.__new__(PixTrackFeatureExtractor)
    fe.conf = type("C", (), {"resize": 1024, "resize_by": "max"})()
    h, w, sc = fe.target_size(1080, 1920, 1)
    assert (h, w) == (576, 1024) and sc == (1024 / 1920, 1024 / 1920)
    h, w, sc = fe.target_size(1081, 1920, 1)   # 576.53 -> 577 rows, the factor stays 1024 / 1920
    assert (h, w) == (577, 1024) and sc == (1024 / 1920, 1024 / 1920)
    assert fe.target_size(480, 640, 1) == (480, 640, (1.0, 1.0))


def test_vectorised_reference_ranking_equals_the_scalar_one():
    """update_reference_ids ranks [current] + covisible (> 50) references by geodesic distance; the one-pass
    form (geodesic_distances_to + argmin) must pick what the reference's dict + stable sort picks."""
    from pixtrack_amd.utils.pose_utils import geodesic_distance_for_rotations, geodesic_distances_to
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(3)
    rots = Rotation.random(40, random_state=5).as_matrix()
    for trial in range(50):
        Rq = Rotation.random(random_state=100 + trial).as_matrix()
        ids = list(rng.permutation(40)[: rng.integers(1, 12)])
        scalar = np.array([geodesic_distance_for_rotations(Rq, rots[i]) for i in ids])
        vector = geodesic_distances_to(Rq, rots[ids])
        assert np.allclose(scalar, vector, rtol=0, atol=1e-14)
        gd = {i: geodesic_distance_for_rotations(Rq, rots[i]) for i in ids}
        assert sorted(gd, key=lambda x: gd[x])[0] == ids[int(np.argmin(vector))]
