"""The N > 1 launch path on a GPU box: `python bench.py --gpus 2` (no torchrun) must itself start two ranks, say who they
were, and anything other than two ranks must be an error.  One GPU is enough: the gloo backend lets both ranks share it
(RCCL refuses two ranks on one device) -- this exercises ensure_ranks / require_world / describe_ranks, the barrier and
max-over-ranks timing and the one all_gather on device-resident work; RCCL itself needs the driver's multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {**os.environ, "PYTHONDONTWRITEBYTECODE": "1"}
for _k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    ENV.pop(_k, None)

pytestmark = pytest.mark.gpu


def _json_line(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_bench_gpus_2_spawns_two_ranks_itself(dev):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch", "16", "--cpu-meshes", "0", "--secondary", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=ENV)
    assert out.returncode == 0, out.stderr[-3000:]
    res = _json_line(out.stdout)
    assert res["n_gpus"] == 2 and res["self_spawned"] is True and res["backend"] == "gloo"
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and len({r["pid"] for r in res["ranks"]}) == 2
    assert all(r["device_index"] is not None and r["name"] for r in res["ranks"])
    assert len(res["per_rank_seconds"]) == 2 and res["value"] > 0
    assert res["config"]["meshes_per_step_per_gpu"] == 16


def test_bench_under_torchrun_the_drivers_form(dev):
    """the command line the driver uses for N > 1 (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...), with gloo so that both ranks may share this box's one GPU"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
           "--batch", "16", "--cpu-meshes", "0", "--secondary", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=ENV)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE JSON line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["self_spawned"] is False and res["scaling"] == "weak" and res["steps"] == 3
    assert res["value"] == pytest.approx(16 * 3 * 2 / max(res["per_rank_seconds"]), rel=1e-3)    # whole-job rate over the slowest rank


def test_bench_rank_mismatch_exits_nonzero(dev):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29651", os.path.join(REPO, "bench.py"), "--gpus", "4", "--backend", "gloo", "--steps", "1", "--warmup", "1",
           "--batch", "8", "--cpu-meshes", "0", "--secondary", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=ENV)
    assert out.returncode != 0 and "refusing to report" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # the error record -- never a 2-rank measurement under a 4-GPU label
    res = json.loads(lines[0])
    assert res["stage"] == "spawn" and "--gpus 4 but 2 rank(s)" in res["error"] and "value" not in res


def test_bench_more_gpus_than_the_node_has_is_an_error(dev):
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=ENV)
    assert out.returncode == 2 and f"--gpus {n}" in out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # over-subscription: the error line, stage "spawn", nothing started
    res = json.loads(lines[0])
    assert res["stage"] == "spawn" and f"--gpus {n}" in res["error"] and res["n_gpus"] == n and res["ranks"] == []


@pytest.mark.parametrize("model", ["pointnetpp", "pointtransformer"])
def test_forward_sharded_two_ranks_equal_one_rank(dev, tmp_path, model):
    """tools/forward_sharded.py (trainer.py:49-54 sharded): the LossMeter averages of two ranks (self-spawned, gloo, sharing
    this GPU) equal the one-process loop's -- every scan's loss is computed by the same kernels either way."""
    from toothgroupnetwork_amd import eval_sharded
    root = str(tmp_path / "pre")
    eval_sharded.write_synthetic_preprocessed(root, 3, n_points=8000)
    base = [sys.executable, os.path.join(REPO, "tools", "forward_sharded.py"), "--input_data_dir_path", root, "--model", model]
    one = subprocess.run(base + ["--gpus", "1"], capture_output=True, text=True, timeout=900, env=ENV)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run(base + ["--gpus", "2", "--backend", "gloo"], capture_output=True, text=True, timeout=900, env=ENV)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = _json_line(one.stdout), _json_line(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and a["scans"] == b["scans"] == 3 and b["per_rank_steps"] == [2, 1]
    assert b["self_spawned"] is True and len(b["ranks"]) == 2
    assert set(a["avg"]) == {"tooth_class_loss_1_val", "total_val"}
    for k in a["avg"]:
        assert a["avg"][k] > 0 and b["avg"][k] == pytest.approx(a["avg"][k], rel=1e-6)


def test_rccl_initialises_and_runs_the_metric_gather_with_one_rank(dev):
    """The boxes this suite runs on have one GPU, so the N > 1 tests above use gloo.  RCCL itself can still be brought up with ONE
    rank: backend "nccl" (= RCCL on ROCm) initialised the way sharding.init_from_env does it (device_id), the collective of the
    sharded runners (all_gather_into_tensor of an fp64 vector on the device), a barrier, and the rank record bench.py prints."""
    code = r'''
import os, sys, json
sys.path.insert(0, os.environ["TGN_REPO"])
import torch, torch.distributed as dist
from toothgroupnetwork_amd import launch
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
vec = torch.tensor([3.0, 0.25, 7.0], dtype=torch.float64, device=dev)
out = torch.empty(3, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(out, vec)
dist.barrier()
torch.cuda.synchronize()
objs = [None]
dist.all_gather_object(objs, {"rank": dist.get_rank()})
print(json.dumps({"out": out.cpu().tolist(), "backend": str(dist.get_backend()), "rccl": launch.rccl_version(), "objs": objs}))
dist.destroy_process_group()
'''
    env = {**ENV, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29657", "TGN_REPO": REPO, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = _json_line(out.stdout)
    assert res["out"] == [3.0, 0.25, 7.0] and res["backend"] == "nccl" and res["rccl"] and res["objs"] == [{"rank": 0}]
