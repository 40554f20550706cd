#!/usr/bin/env python3
"""The reference's validation loop (trainer.py:49-54: one forward + loss per scan, LossMeter averages) over a directory
of preprocessed scans, sharded over the GPUs of one node -- one process per GPU, no exchange while computing, ONE
all_gather of the loss sums at the end (RCCL over xGMI).

    python tools/forward_sharded.py --gpus 8 --input_data_dir_path data_preprocessed_path --model pointnetpp
    python tools/forward_sharded.py --gpus 8 --synthetic 64 --model pointtransformer [--checkpoint ckpt.h5]

`--gpus N` starts the N ranks itself (toothgroupnetwork_amd.launch.ensure_ranks) when it was not launched by torchrun; the
torchrun form works too:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        tools/forward_sharded.py --gpus 8 --input_data_dir_path DIR

A rank count that differs from --gpus is an error, never a warning.  Rank 0 prints one JSON line: the LossMeter averages
over all scans (the numbers `Trainer.test` logs), scans/s of the whole job, the ranks that took part."""
import argparse
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from toothgroupnetwork_amd import eval_sharded, launch, sharding  # noqa: E402

METRIC = "validation scans/sec (per-scan forward + loss, trainer.py:49-54)"


def build_step(name, device, checkpoint=None, seed=0):
    import torch

    from toothgroupnetwork_amd import nets
    torch.manual_seed(seed)                       # every rank builds the same random-init weights when no checkpoint is given
    if name == "pointnetpp":
        module, step_cls = nets.PointNetPPSeg(), eval_sharded.PointNetPPStep
    elif name == "pointtransformer":
        module, step_cls = nets.PointTransformerSeg(), eval_sharded.PointTransformerStep
    else:
        raise SystemExit(f"unknown --model {name!r}")
    if checkpoint:
        module.load_state_dict(eval_sharded.reference_state_dict(torch.load(checkpoint, map_location="cpu")))
    return step_cls(module, device)


def main(argv=None, step_factory=None, script=None):
    """step_factory(device) -> step object: tests of the control flow pass their own (CPU) step; None = the GPU networks.
    script: the file the self-spawned ranks run (a test-side launcher passes itself)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--input_data_dir_path", default=None, help="directory of *_sampled_points.npy (generator.py:13)")
    ap.add_argument("--synthetic", type=int, default=0, help="write this many synthetic preprocessed scans first (no dataset here)")
    ap.add_argument("--model", default="pointnetpp", choices=["pointnetpp", "pointtransformer"])
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--backend", default=None, help="default nccl (= RCCL) on GPUs, gloo on CPU")
    ap.add_argument("--points", type=int, default=24000)
    args = ap.parse_args(argv)
    launch.ensure_ranks(args.gpus, script=script or os.path.abspath(__file__), argv=sys.argv[1:] if argv is None else list(argv),
                        backend=args.backend, metric=METRIC)
    launch.begin(METRIC)
    launch.require_world(args.gpus, sharding.env_rank_world()[2])
    rank, local_rank, world, device = sharding.init_from_env(backend=args.backend)
    import torch
    # host threads of this rank: its share of the cores it can really use (torchrun sets OMP_NUM_THREADS=1 for its children; a single
    # process would otherwise run every small CPU op on one thread per hardware thread of the box)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), sharding.cpus_for_this_rank(int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    if step_factory is None and device.type != "cuda":
        launch.fail("forward_sharded.py needs a ROCm GPU: the operators have no CPU implementation", stage_name="setup")
    root = args.input_data_dir_path
    if args.synthetic:
        root = root or os.environ.get("TGN_SYNTH_EVAL_DIR") or os.path.join(tempfile.gettempdir(), f"tgn_eval_{args.synthetic}_{args.points}")
        eval_sharded.write_synthetic_preprocessed(root, args.synthetic, rank, world, n_points=args.points)
        sharding.barrier()
    if not root:
        launch.fail("--input_data_dir_path or --synthetic N", stage_name="setup", code=2)
    paths = eval_sharded.list_preprocessed(root)
    step = step_factory(device) if step_factory is not None else build_step(args.model, device, args.checkpoint)
    launch.stage("calibrate")
    if step_factory is None and paths:            # first use of the GPU by this rank (context, code objects, memoised folds): no scan's time
        step(-1, eval_sharded.load_item(paths[rank % len(paths)]))
    launch.stage("timed")
    res = eval_sharded.eval_sharded(paths, step, rank, world, device=device)      # (stage "gather" inside, around the one collective)
    launch.stage("report")
    ranks = launch.describe_ranks(device)
    if rank == 0:
        launch.emit({"metric": METRIC, "value": res["scans_per_s"],
                     "unit": "scans/s", "n_gpus": world, "model": args.model, "scans": res["steps"], **res, **ranks})
    launch.shutdown()


if __name__ == "__main__":
    launch.guard(main)
