"""Test-side mini runner for the failure records of a multi-rank run (toothgroupnetwork_amd/launch.py): brings the ranks up
the way bench.py does -- launch.begin, sharding.bring_up_group, the one gather, launch.emit under launch.guard -- on CPU
ranks, with one fault injected:

  rccl_all     RCCL cannot come up on any rank (its init raises)            -> the gather runs over gloo, backend says why
  rccl_rank1   RCCL comes up on rank 0 only                                 -> all ranks agree on gloo
  raise_rank1  rank 1 raises while calibrating                              -> error line, stage "calibrate", failed_rank 1
  exit_rank1   rank 1 disappears (os._exit) before the gather               -> error line from the watcher
  kill_rank0   rank 0 is SIGKILLed inside the timed region                  -> error line from the watcher, stage "timed"
  raise_rank0  rank 0 raises in the timed region                            -> error line from rank 0 itself
  none         no fault
"""
import argparse
import os
import signal
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from toothgroupnetwork_amd import launch, sharding  # noqa: E402

METRIC = "fault-injection runner (tests)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--fault", default="none")
    args = ap.parse_args()
    launch.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], backend="gloo", metric=METRIC)
    launch.begin(METRIC)
    rank, _, world = sharding.env_rank_world()
    launch.require_world(args.gpus, world)
    device = torch.device("cpu")
    backend = "gloo"
    if args.fault in ("rccl_all", "rccl_rank1"):
        backend = "nccl"

        def fake_rccl(rank_, world_, device_, timeout_):
            if args.fault == "rccl_all" or rank_ == 1:
                raise RuntimeError("injected: ncclCommInitRank failed (hipIpcGetMemHandle: invalid argument)")
            # (rank 0 of rccl_rank1: "RCCL" came up here -- the agreement step must still move every rank to gloo)

        sharding._init_rccl = fake_rccl
    if world > 1:
        sharding.bring_up_group(rank, world, device, backend, timeout_s=30.0)
    launch.stage("calibrate")
    if args.fault == "raise_rank1" and rank == 1:
        raise ValueError("injected: calibration failed on rank 1")
    if args.fault == "exit_rank1" and rank == 1:
        os._exit(7)
    launch.stage("timed")
    sharding.barrier()
    if args.fault == "kill_rank0" and rank == 0:
        os.kill(os.getpid(), signal.SIGKILL)
    if args.fault == "raise_rank0" and rank == 0:
        raise RuntimeError("injected: the timed region failed on rank 0")
    time.sleep(0.05)
    launch.stage("gather")
    mat = sharding.gather_metrics([float(rank), 1.0], device=device)
    who = launch.describe_ranks(device)
    if rank == 0:
        launch.emit({"metric": METRIC, "value": float(mat[:, 1].sum()), "n_gpus": world, **who})
    launch.shutdown()


if __name__ == "__main__":
    launch.guard(main)
