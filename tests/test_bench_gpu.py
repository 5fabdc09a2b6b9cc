"""bench.py's contract (one JSON line with the fields the driver reads) and a rehearsal of its
multi-rank control flow on a single GPU (gloo, ranks sharing cuda:0)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields(device):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["unit"] == "frames/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["tracked_ok"] == d["frames_total"] == 12
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.05 < r["frac"] < 1.0
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    # the extra passes that say what the headline leaves out (two renders, host frames + debug 1, K = 200)
    ex = d["extras"]
    for k in ("value_two_renders", "value_host_frames_debug1", "value_k200", "value_ycb_policy"):
        assert ex[k]["frames_per_s"] > 30.0 and ex[k]["tracked_ok"] == ex[k]["frames"], (k, ex[k])
    # the reference's own reference-image shapes (VERDICT r3 missing #1): 921x921 under the YCB policy, 960x720 and
    # 2016x1512 (-> 1024x768 in the extractor) under r9's
    for k, wh in (("value_ycb_refshape", [921, 921]), ("value_r9_phone", [960, 720]), ("value_r9_12mp", [2016, 1512])):
        assert ex[k]["reference_render_wh"] == wh and ex[k]["frames_per_s"] > 30.0, (k, ex[k])
        assert ex[k]["tracked_ok"] == ex[k]["frames"], (k, ex[k])
    assert ex["value_r9_12mp"]["reference_unet_input_wh"] in ([1024, 768], [640, 480])
    # the real-asset frame as a measured object: stage times + live UNet / LM rooflines of those passes
    for k in ("value_ycb_refshape", "value_r9_phone"):
        rb = ex[k]["roofline_refshape"]
        assert set(rb["stage_ms_per_frame"]) >= {"nerf_render", "unet", "lm"} and 0.05 < rb["unet"]["frac"] < 1.0, (k, rb)
        assert rb["lm"]["kernel_us"] > 10.0 and rb["unet"]["image_sizes_hw"][1] == [480, 640], (k, rb)
        # the reference pass ran on a window of the render (refiner.reference_window), not on all of it
        w, h = ex[k]["reference_unet_input_wh"]
        assert w * h < 0.7 * ex[k]["reference_render_wh"][0] * ex[k]["reference_render_wh"][1], (k, ex[k])
    # live per-stage rooflines of the headline run; what is read from committed profiles says so
    rs = d["roofline_stages"]
    assert 0.05 < rs["unet"]["frac"] < 1.0 and rs["unet"]["gflop_per_call"] == pytest.approx(2 * 241.4, rel=0.02), rs["unet"]
    assert 0.01 < rs["lm"]["frac"] < 1.0 and sum(rs["lm"]["iterations_per_level_mean"]) >= 3, rs["lm"]
    assert d["roofline"]["traffic_static"]["static"] is True and d["kernel_utilisation"]["static"] is True
    assert ex["value_two_renders"]["frames_per_s"] < d["value"] * 1.05
    assert ex["value_ycb_policy"]["renders_ahead_used"] >= ex["value_ycb_policy"]["frames"] - 2


def test_objects8_and_hd_workloads_run(device):
    """configs[3] (one config/*.sh object per rank) and configs[4] (1920x1080, 4-level stress pyramid,
    frame segments) at N = 1."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "objects8", "--steps", "8", "--warmup", "3",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    # configs[3] on ONE GPU: all eight config/*.sh objects in lock-step (MultiObjectTracker), 8 frames each
    assert "configs[3]" in d["config"]["workload"] and d["config"]["objects_per_rank"][0][0] == "bottle"
    assert len(d["config"]["objects_per_rank"][0]) == 8 and d["scaling"] == "strong"
    assert d["tracked_ok"] == d["frames_total"] == 64, (d["tracked_ok"], d["frames_total"])
    assert d["lockstep_frames"] == 8 * (3 + 8 + 4 - 1) and d["solo_frames_inside_lockstep"] == 8  # only the cold starts ran alone
    st = d["roofline_stages"]
    assert st["unet"]["images_per_call"] == 16 and 0.05 < st["unet"]["frac"] < 1.0 and st["lm"]["problems_per_launch"] == 8
    solo = d["solo_runs"]
    assert len(solo["frames_per_s_per_object"]) == 8 and solo["lockstep_speedup"] > 1.0, solo
    # defaults (batch-planned UNet layers, 32 LM workgroups per problem) against solo runs: fp32 summation order only -
    # the oracle bar (measured <= 6e-4 over 20 steps; the solo passes also encode reference WINDOWS for two objects)
    # (the thin slab's ill-conditioned LM problem - DESIGN.md section 6 - amplifies the same differences: 2.7e-3 over 20 steps)
    diffs = solo["max_abs_pose_difference_vs_lockstep"]
    assert max(x for i, x in enumerate(diffs) if i != 6) < 1e-3 and diffs[6] < 5e-3, solo
    # ... and the configuration that promises the solo runs' very bits keeps them, checked inside the bench run
    vb = d["value_bit_identical"]
    assert vb["bit_identical_to_solo_runs"] is True and vb["max_abs_pose_record_difference"] == 0.0 and vb["tracked_ok"] == 64, vb
    assert set(d["max_rot_err_vs_gt_rad"]) == set(d["config"]["objects_per_rank"][0])
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "hd", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["config"]["width"] == 1920 and d["scaling"] == "strong" and d["frames_total"] == 6
    assert d["config"]["lm_levels_per_frame"] == 4 and d["config"]["n_points_per_reference"] > 8000
    assert d["tracked_ok"] >= 5 and d["mean_rot_err_vs_gt_rad"] < 2e-2


def test_rccl_world_size_one_gather(device):
    """RCCL is loaded and runs a collective on the hardware (world size 1: init + all_gather + all_reduce)."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
        "from pixtrack_amd import parallel\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "torch.cuda.set_device(0)\n"
        "rec = torch.arange(28, dtype=torch.float64).reshape(2, 14)\n"
        "out = [torch.zeros(2, 14, dtype=torch.float64, device='cuda')]\n"
        "dist.all_gather(out, rec.cuda())\n"
        "assert torch.equal(out[0].cpu(), rec)\n"
        "t = torch.tensor([3.5], dtype=torch.float64, device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "assert float(t) == 3.5\n"
        "print('RCCL', torch.cuda.nccl.version()); dist.destroy_process_group()\n"
    )
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ROOT), env=env)
    assert out.returncode == 0 and "RCCL" in out.stdout, out.stderr[-3000:]


def test_two_ranks_self_launched_and_reported(device):
    """`python bench.py --gpus 2` with NO launcher spawns its own two ranks (gloo rehearsal: the ranks share
    cuda:0) and the line says what the collective layer saw."""
    env = dict(os.environ, PXT_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)  # rank 0 only prints
    assert d["n_gpus"] == 2 == len(d["ranks_seen"]) and d["ranks_seen"] == [0, 1]
    assert d["self_launched"] is True and d["dist_backend"] == "gloo"
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and len({r["pid"] for r in d["ranks"]}) == 2
    for r in d["ranks"]:
        assert r["frames"] == 10 and r["frames_per_s"] > 10.0 and r["device"].startswith("cuda:") and r["pci_bus"]
    assert d["frames_total"] == 20 and d["tracked_ok"] == 20
    assert abs(d["value"] - 20 / (10 * d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert d["value"] <= sum(r["frames_per_s"] for r in d["ranks"]) * 1.001  # max-over-ranks time
    assert d["cpu_baseline"]["value"] is None  # timed on rank 0 at N = 1 only


def test_launcher_shape_still_works_and_a_wrong_world_size_is_refused(device):
    """The driver's N > 1 command (torch.distributed.run) keeps working; a world size that differs from --gpus
    is an error, never a silently mislabelled number."""
    env = dict(os.environ, PXT_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
           "--config", "objects8", "--object-index", "6"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["ranks_seen"] == [0, 1] and d["self_launched"] is False and d["frames_total"] == 12 == d["tracked_ok"]
    assert "roncelli_blankk" in d["config"]["workload"]  # rank 0 = object 6 (the thin slab), rank 1 = object 7
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "2"],
                         capture_output=True, text=True, timeout=300, cwd=str(ROOT), env=env2)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_more_gpus_than_the_node_has_is_refused(device):
    """RCCL needs one GPU per rank: `--gpus 8` on a 1-GPU box exits non-zero with a clear message and no JSON."""
    import torch

    n = torch.cuda.device_count() + 7
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PXT_DIST_BACKEND")}
    for cfg in ("frames640", "objects8", "hd"):
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "2",
                              "--config", cfg], capture_output=True, text=True, timeout=300, cwd=str(ROOT), env=env)
        assert out.returncode != 0, cfg
        assert "GPU(s) visible" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("k", range(8))
def test_objects8_every_object_tracks_at_640x480(device, k):
    """BASELINE configs[3] at the metric's resolution: `bench.py --config objects8 --object-index k` for every
    config/*.sh object (k = 0 .. 7) at 640 x 480, spp 8; every frame must be tracked."""
    names = ["bottle", "cracker_box", "gimble", "motor_core", "pickle_rick", "premier_protein", "roncelli_blankk",
             "spirit_level"]
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "objects8", "--object-index", str(k),
                          "--steps", "12", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True,
                         timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert names[k] in d["config"]["workload"] and d["config"]["width"] == 640 and d["config"]["height"] == 480
    assert d["tracked_ok"] == d["frames_total"] == 12, (names[k], d["tracked_ok"])
    # (error against the SYNTHETIC ground truth: the bottle is nearly a solid of revolution, its rotation about the long
    # axis is weakly observable - 0.04 rad there, 1e-3 .. 1e-2 for the boxes; every frame passes the tracker's own gates)
    # (round 5: the cold-start frame of the objects8 workload carries sigma 24 instead of 12 - the bottle's steady frames
    # were refused by the cost gate otherwise - and the bottle's free rotation about its axis starts from a noisier pose)
    # (round 6: the CPU oracle tracks the bottle's scene with the same error - 0.04 rising to 0.137 rad at frame 15 and back to
    # 0.001 by frame 64, profiles/r06_oracle_drift_bottle*.log beside r06_hip_drift_bottle.log: 0.2 rad bounds the
    # algorithm's own excursion on this scene, it is not slack for the HIP path, which the oracle fixtures hold to 1e-3)
    assert d["mean_rot_err_vs_gt_rad"] < (0.2 if k == 0 else 0.1) and d["mean_trans_err_vs_gt"] < 0.05 and d["value"] > 100.0
    if k == 0:
        # the headline's cold-start noise (sigma 12) on the bottle: the cost gate (1.1 x the first frame's cost) refuses one
        # of these twelve steady frames - the run that made the objects8 workload take sigma 24.  Kept as its own assertion
        # so that a change of the gate's behaviour stays visible: 11 of 12 measured, 10 .. 12 accepted.
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "objects8", "--object-index", "0", "--steps", "12",
                              "--warmup", "3", "--no-cpu-baseline", "--first-frame-sigma", "12"], capture_output=True, text=True,
                             timeout=900, cwd=str(ROOT))
        assert out.returncode == 0, out.stderr[-3000:]
        d12 = _last_json(out.stdout)
        assert d12["frames_total"] == 12 and 10 <= d12["tracked_ok"] <= 12, d12["tracked_ok"]


def test_eight_ranks_rehearsal_on_one_gpu(device):
    """The driver's 8-GPU command shape with eight gloo ranks sharing cuda:0 (ports, NUMA binding, eight processes'
    memory, the pose gather of eight ranks): the first real 8-GPU node must not also be the first 8-rank run."""
    env = dict(os.environ, PXT_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", "29561", str(ROOT / "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 8 and d["ranks_seen"] == list(range(8)) and d["dist_backend"] == "gloo"
    assert len({r["pid"] for r in d["ranks"]}) == 8 and all(r["frames"] == 6 for r in d["ranks"])
    assert d["frames_total"] == 48 == d["tracked_ok"] and d["scaling"] == "weak"
    # eight independently seeded sequences: the gathered records are eight DIFFERENT tracks
    assert d["value"] <= sum(r["frames_per_s"] for r in d["ranks"]) * 1.001
    # the objects8 workload on the same shape: rank r tracks object r
    cmd = cmd[:-6] + ["--gpus", "8", "--steps", "4", "--warmup", "2", "--config", "objects8"]
    cmd[cmd.index("29561")] = "29563"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["ranks_seen"] == list(range(8)) and d["frames_total"] == 32 == d["tracked_ok"]
    # ... and on TWO ranks: four objects per rank in lock-step (rank 0: objects 0, 2, 4, 6; rank 1: 1, 3, 5, 7)
    cmd[cmd.index("--nproc-per-node") + 1] = "2"
    cmd[cmd.index("--gpus") + 1] = "2"
    cmd[cmd.index("29563")] = "29565"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["frames_total"] == 32 == d["tracked_ok"]
    assert d["config"]["objects_per_rank"] == [["bottle", "gimble", "pickle_rick", "roncelli_blankk"],
                                               ["cracker_box", "motor_core", "premier_protein", "spirit_level"]]
    assert all(r["frames"] == 16 for r in d["ranks"]) and d["scaling"] == "strong"
