"""The multi-GPU launch rules (toothgroupnetwork_amd/launch.py) and the sharded validation loop
(toothgroupnetwork_amd/eval_sharded.py, tools/forward_sharded.py = trainer.py:49-54 over ranks) on CPU / gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER = os.path.join(REPO, "tests", "forward_launcher.py")
ENV = {**os.environ, "PYTHONDONTWRITEBYTECODE": "1"}
for _k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    ENV.pop(_k, None)


def _json_line(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_ensure_ranks_spawns_only_when_needed():
    from toothgroupnetwork_amd import launch
    calls = []
    fake = lambda path, argv, env: calls.append((path, argv, env))   # noqa: E731
    launch.ensure_ranks(1, "bench.py", ["--gpus", "1"], env={}, run=fake)
    launch.ensure_ranks(4, "bench.py", ["--gpus", "4"], env={"WORLD_SIZE": "4", "RANK": "0"}, run=fake)
    assert calls == []                                   # one GPU, or already under torchrun: nothing to start
    launch.ensure_ranks(4, "bench.py", ["--gpus", "4", "--steps", "3"], backend="gloo", env={"PATH": "/bin"}, run=fake)
    (path, argv, env), = calls
    assert path == sys.executable and argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert argv[-5:] == ["bench.py", "--gpus", "4", "--steps", "3"]
    assert env["TGN_SELF_SPAWNED"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert env["TGN_RUN_DIR"]                            # the ranks' status files and the supervisor meet there


def test_require_world_is_an_error_not_a_warning(capsys):
    from toothgroupnetwork_amd import launch
    launch.require_world(1, 1)
    launch.require_world(0, 1)
    with pytest.raises(SystemExit) as e:
        launch.require_world(8, 1)
    assert e.value.code == 2 and "--gpus 8" in capsys.readouterr().err


def test_bench_refuses_a_rank_count_other_than_gpus():
    """bench.py under torchrun with 2 ranks but --gpus 4: exit status != 0 (round 4: a warning and an n_gpus=2 line).
    No GPU here, so the ranks stop at the first check that applies -- on a GPU box the same call stops at require_world
    (tests/test_gpu_launch.py)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                               # ONE line also on failure: the error record, never a measurement
    res = json.loads(lines[0])
    assert res["stage"] == "spawn" and "error" in res and "value" not in res


def _serial(root):
    """Trainer.test's loop as the reference runs it: one process, every scan in order, LossMeter averages."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from forward_launcher import CpuStep
    from toothgroupnetwork_amd import eval_sharded
    paths = eval_sharded.list_preprocessed(root)
    meter, step = eval_sharded.LossMeter(), CpuStep(None)
    for i, p in enumerate(paths):
        meter.aggr(step(i, eval_sharded.load_item(p)))
    return meter.get_avg_results(), len(paths)


@pytest.mark.parametrize("how", ["self_spawn", "torchrun"])
def test_forward_sharded_two_gloo_ranks_equal_the_serial_loop(tmp_path, how):
    from toothgroupnetwork_amd import eval_sharded
    root = str(tmp_path / "pre")
    eval_sharded.write_synthetic_preprocessed(root, 5, n_points=600)
    arr = np.load(eval_sharded.list_preprocessed(root)[0])
    assert arr.shape == (600, 7) and arr.dtype == np.float64 and set(np.unique(arr[:, 6])) <= set(range(17))
    want, n = _serial(root)
    args = [LAUNCHER, "--gpus", "2", "--backend", "gloo", "--input_data_dir_path", root]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29643"] + args
    else:
        cmd = [sys.executable] + args                    # no torchrun: the runner starts its two ranks itself
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0, out.stderr[-3000:]
    res = _json_line(out.stdout)
    assert res["n_gpus"] == 2 and res["scans"] == n == 5 and res["per_rank_steps"] == [3, 2]
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and res["ranks"][0]["pid"] != res["ranks"][1]["pid"]
    assert res["backend"] == "gloo" and res["self_spawned"] == (how == "self_spawn")
    assert set(res["avg"]) == set(want)
    for k, v in want.items():
        assert res["avg"][k] == pytest.approx(v, rel=1e-12, abs=1e-12)


def test_forward_sharded_more_ranks_than_scans(tmp_path):
    """one scan, two ranks: rank 1's shard is empty -- its vector is all zeros, the schema is still the step's fixed key tuple, and
    the averages are the one scan's"""
    from toothgroupnetwork_amd import eval_sharded
    root = str(tmp_path / "pre")
    eval_sharded.write_synthetic_preprocessed(root, 1, n_points=400)
    want, n = _serial(root)
    out = subprocess.run([sys.executable, LAUNCHER, "--gpus", "2", "--backend", "gloo", "--input_data_dir_path", root],
                         capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 0, out.stderr[-3000:]
    res = _json_line(out.stdout)
    assert n == 1 and res["scans"] == 1 and res["per_rank_steps"] == [1, 0]
    for k, v in want.items():
        assert res["avg"][k] == pytest.approx(v, rel=1e-12, abs=1e-12)


def test_forward_sharded_rank_mismatch_is_an_error(tmp_path):
    from toothgroupnetwork_amd import eval_sharded
    root = str(tmp_path / "pre")
    eval_sharded.write_synthetic_preprocessed(root, 2, n_points=300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29645", LAUNCHER, "--gpus", "3", "--backend", "gloo", "--input_data_dir_path", root]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode != 0 and "refusing to report" in out.stderr


def test_loss_meter_and_print_dict_follow_the_reference():
    """loss_meter.py:2-23 / :50-62 semantics: weighted values, a total, sums over steps, averages by step count"""
    from toothgroupnetwork_amd import eval_sharded as E
    d = E.print_dict({"a": (2.0, 0.5), "b": (3.0, 1)}, "val")
    assert d == {"a_val": 1.0, "b_val": 3.0, "total_val": 4.0}
    m = E.LossMeter()
    m.aggr(d)
    m.aggr({"a_val": 3.0, "b_val": 1.0, "total_val": 4.0})
    assert m.step_num == 2 and m.get_avg_results() == {"a_val": 2.0, "b_val": 2.0, "total_val": 4.0}
    m.init()
    assert m.step_num == 0 and m.loss_meter_dict == {}


@pytest.mark.parametrize("net,prefix", [("PointNetPPSeg", "first_sem_model."), ("PointTransformerSeg", "first_ins_cent_model.")])
def test_forward_sharded_loads_a_reference_style_checkpoint(tmp_path, net, prefix):
    """the reference saves `self.module.state_dict()` of its wrapper module (base_model.py:33-37): the network's keys carry the
    wrapper's attribute name (pointnet_pp.py:78, point_transformer.py:9) and the loss's buffers ride along as `criterion.*`;
    tools/forward_sharded.py --checkpoint must load such a file into the bare network mirror, and a bare state_dict still loads"""
    import importlib.util

    import torch

    from toothgroupnetwork_amd import eval_sharded, nets
    torch.manual_seed(3)
    src = getattr(nets, net)()
    wrapped = {prefix + k: v.clone() for k, v in src.state_dict().items()}
    wrapped["criterion.class_weight"] = torch.ones(17)
    stripped = eval_sharded.reference_state_dict(wrapped)
    assert set(stripped) == set(src.state_dict()) and eval_sharded.reference_state_dict(src.state_dict()) is not None
    path = str(tmp_path / "ckpt.h5")
    torch.save(wrapped, path)
    spec = importlib.util.spec_from_file_location("forward_sharded", os.path.join(REPO, "tools", "forward_sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    name = "pointnetpp" if net == "PointNetPPSeg" else "pointtransformer"
    step = mod.build_step(name, torch.device("cpu"), checkpoint=path, seed=99)       # (seed 99: other weights unless the file is loaded)
    for k, v in src.state_dict().items():
        assert torch.equal(step.module.state_dict()[k], v), k
    bare = str(tmp_path / "bare.h5")
    torch.save(src.state_dict(), bare)
    step = mod.build_step(name, torch.device("cpu"), checkpoint=bare, seed=98)
    assert all(torch.equal(step.module.state_dict()[k], v) for k, v in src.state_dict().items())
