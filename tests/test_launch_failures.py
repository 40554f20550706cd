"""A multi-rank run must be diagnosable from its ONE JSON line whatever fails (toothgroupnetwork_amd/
launch.py, sharding.bring_up_group): every failure below is injected into tests/fault_launcher.py -- the bring-up sequence of
bench.py on CPU ranks -- and the record is checked: exactly one line starting with "{", `error` + `stage` + per-rank records on
failure, a non-zero exit status; and the RCCL -> gloo fallback keeps the run alive and says so in `backend`."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER = os.path.join(REPO, "tests", "fault_launcher.py")
ENV = {**os.environ, "PYTHONDONTWRITEBYTECODE": "1", "TGN_WATCHER_GRACE_S": "0.5"}
for _k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TGN_RUN_DIR"):
    ENV.pop(_k, None)


def _run(fault, how="self_spawn", port=None, extra_env=None, gpus=2):
    args = [LAUNCHER, "--gpus", str(gpus), "--fault", fault]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**ENV, **(extra_env or {})})
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def test_no_fault_one_result_line():
    out, lines = _run("none")
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert "error" not in res and res["n_gpus"] == 2 and res["backend"] == "gloo" and res["value"] == 2.0
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and all("numa_pin" in r for r in res["ranks"])


@pytest.mark.parametrize("fault,how", [("rccl_all", "self_spawn"), ("rccl_all", "torchrun"), ("rccl_rank1", "self_spawn")])
def test_rccl_init_failure_falls_back_to_gloo_and_says_so(fault, how):
    """init_process_group("nccl") raising -- on every rank, or on one only -- must not lose the run: the <= 1 KB gather goes over
    gloo, the value is the same, `backend` records what happened"""
    out, lines = _run(fault, how, port=29671)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert "error" not in res and res["n_gpus"] == 2 and res["value"] == 2.0
    assert res["backend"].startswith("gloo (rccl init failed:") and "injected" in res["backend"] or "rank(s) [1]" in res["backend"]
    assert res["rccl_version"] is None


@pytest.mark.parametrize("how", ["self_spawn", "torchrun"])
def test_a_rank_that_raises_gives_the_error_line_with_its_stage(how):
    out, lines = _run("raise_rank1", how, port=29673)
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    res = json.loads(lines[0])
    assert res["stage"] == "calibrate" and res["failed_rank"] == 1 and "injected: calibration failed on rank 1" in res["error"]
    assert res["n_gpus"] == 2 and "value" not in res
    by_rank = {r["rank"]: r for r in res["ranks"]}
    assert by_rank[1]["stage"] == "calibrate" and "traceback" in by_rank[1] and by_rank[0]["pid"] != by_rank[1]["pid"]


def test_a_rank_that_exits_early_gives_the_error_line():
    """rank 1 vanishes without a word before the gather: torchrun ends rank 0, whose watcher prints the line"""
    out, lines = _run("exit_rank1", "torchrun", port=29675)
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    res = json.loads(lines[0])
    assert "error" in res and res["stage"] in ("timed", "gather", "calibrate") and res["reported_by"] in ("watcher", "rank 0")
    assert {r["rank"] for r in res["ranks"]} == {0, 1}


def test_rank0_killed_gives_the_error_line_from_its_watcher():
    out, lines = _run("kill_rank0", "self_spawn")
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout + out.stderr[-2000:]
    res = json.loads(lines[0])
    assert res["reported_by"] in ("watcher", "supervisor") and res["stage"] == "timed" and res["failed_rank"] is None


def test_single_process_failure_gives_the_error_line():
    out, lines = _run("raise_rank0", gpus=1)
    assert out.returncode != 0
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["stage"] == "timed" and res["failed_rank"] == 0 and res["reported_by"] == "rank 0" and res["n_gpus"] == 1


def test_rank_mismatch_gives_the_error_line_and_status_2():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29677", LAUNCHER, "--gpus", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=ENV)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode != 0 and "refusing to report" in out.stderr
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["stage"] == "spawn" and "--gpus 3 but 2 rank(s)" in res["error"]


def test_empty_numa_intersection_leaves_the_mask_alone(tmp_path):
    """pin_to_gpu_numa: a local_cpulist that shares no CPU with the mask the process was given (a container pinned to the other
    socket) must leave the process unpinned and say so in the rank record"""
    from toothgroupnetwork_amd import sharding
    before = os.sched_getaffinity(0)
    bdf = "0000:f5:00.0"
    os.makedirs(tmp_path / bdf)
    outside = max(before) + 1000
    (tmp_path / bdf / "local_cpulist").write_text(f"{outside}-{outside + 7}\n")
    assert sharding.pin_to_gpu_numa(0, sysfs=str(tmp_path), bdf=bdf) is None
    assert os.sched_getaffinity(0) == before
    rec = sharding.pin_record()
    assert rec["pinned"] is False and rec["cpus"] is None and "does not intersect" in rec["reason"] and rec["n"] == len(before)
    # an intersecting list pins to the intersection and records the cpulist
    keep = sorted(before)[: max(1, len(before) // 2)]
    (tmp_path / bdf / "local_cpulist").write_text(sharding.cpulist_string(keep + [outside]) + "\n")
    try:
        assert sharding.pin_to_gpu_numa(0, sysfs=str(tmp_path), bdf=bdf) == keep
        rec = sharding.pin_record()
        assert rec["pinned"] is True and sharding.parse_cpulist(rec["cpus"]) == keep and rec["n"] == len(keep)
    finally:
        os.sched_setaffinity(0, before)
        sharding._PIN.clear()
    # no sysfs entry at all: unpinned, with the reason
    assert sharding.pin_to_gpu_numa(0, sysfs=str(tmp_path), bdf="0000:00:00.0") is None
    assert "FileNotFoundError" in sharding.pin_record()["reason"]
    sharding._PIN.clear()


def test_cpulist_string_round_trips():
    from toothgroupnetwork_amd import sharding
    for cpus in ([0], [0, 1, 2, 3], [0, 2, 4], [3, 4, 5, 9, 10, 64], list(range(128))):
        assert sharding.parse_cpulist(sharding.cpulist_string(cpus)) == cpus


def test_bench_single_gpu_legs_are_impossible_at_n_gt_1():
    """the CPU baseline, the secondary configurations and the prefix-identity re-run belong to N = 1: at N > 1 they would run on
    rank 0 while the other ranks wait in a collective"""
    sys.path.insert(0, REPO)
    import argparse

    import bench
    a = argparse.Namespace(secondary=1, cpu_meshes=-1, no_alt=False)
    bench.single_gpu_legs_only(a, world=8)
    assert (a.secondary, a.cpu_meshes, a.no_alt) == (0, 0, True)
    a = argparse.Namespace(secondary=1, cpu_meshes=-1, no_alt=False)
    bench.single_gpu_legs_only(a, world=1)
    assert (a.secondary, a.cpu_meshes, a.no_alt) == (1, -1, False)
