#!/usr/bin/env python3
"""BASELINE.json config 5: preprocess_data.py over a directory of scans, sharded over the GPUs of one node.

    python tools/preprocess_sharded.py --source_obj_data_path OBJ --source_json_data_path JSON --save_data_path OUT
    python tools/preprocess_sharded.py --gpus 8 --synthetic 512 --save_data_path /tmp/out      (starts its 8 ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        tools/preprocess_sharded.py --synthetic 512 --save_data_path /tmp/out

Same three path arguments as the reference script (preprocess_data.py:8-11).  One process per GPU (RANK / LOCAL_RANK /
WORLD_SIZE from the environment), rank r takes every world-th scan, nothing is exchanged while working, ONE
all_gather of an 8-number fp64 vector at the end (RCCL over xGMI; gloo on CPU).  --synthetic N writes N synthetic raw
scans (about 100 000 vertices each, seeded) into a temporary directory first, for benchmarking without the dataset.
Rank 0 prints one JSON line: scans/s over the whole job, time split into host load (OBJ parse + normals) and FPS."""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toothgroupnetwork_amd import launch, preprocess, sharding, synth  # noqa: E402

METRIC = "preprocessed scans/sec (OBJ parse + normals + FPS N_raw->24000 + npy)"


def _write_one(job):
    root, i = job
    name, jaw = f"SYN{i:05d}_{'upper' if i % 2 == 0 else 'lower'}", ("upper" if i % 2 == 0 else "lower")
    os.makedirs(os.path.join(root, "obj", name), exist_ok=True)
    os.makedirs(os.path.join(root, "json", name), exist_ok=True)
    nu, nv = 330 + (i % 7) * 10, 300
    if os.path.exists(os.path.join(root, "json", name, name + ".json")):   # (written last: the scan is complete)
        return
    with open(os.path.join(root, "obj", name, name + ".obj"), "w") as f:
        f.write(synth.obj_text(nu, nv, 1000 + i, "plain", with_tail=False))
    with open(os.path.join(root, "json", name, name + ".json"), "w") as f:
        json.dump({"jaw": jaw, "labels": synth.fdi_labels(nu * nv, jaw, i)}, f)


def write_synthetic(root, n, rank, world):
    """(forks worker processes: call it BEFORE the process touches the GPU)"""
    import multiprocessing as mp
    jobs = [(root, i) for i in range(rank, n, world)]
    procs = max(1, min(len(jobs), (os.cpu_count() or 1) // max(world, 1), 32))
    if procs == 1:
        for job in jobs:
            _write_one(job)
        return
    def some(k):
        for job in jobs[k::procs]:
            _write_one(job)

    ps = [mp.get_context("fork").Process(target=some, args=(k,)) for k in range(procs)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
        if p.exitcode:
            raise RuntimeError("writing the synthetic scans failed")


def main(argv=None, fps_batch=None, script=None):
    """fps_batch: the sampler handed to preprocess.preprocess_sharded (None = the GPU kernel, the only one this runner
    knows; tests of the control flow pass their own callable through tests/sharded_launcher.py)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--source_obj_data_path", default=None)
    ap.add_argument("--source_json_data_path", default=None)
    ap.add_argument("--save_data_path", default="data_preprocessed_path")
    ap.add_argument("--synthetic", type=int, default=0, help="generate this many synthetic raw scans instead of reading a dataset")
    ap.add_argument("--batch", type=int, default=32, help="most scans per FPS launch (a launch costs ~50 ms whatever its size -- the sampling chain of one raw scan -- "
                    "so the loop takes what the loaders have ready, between a quarter of this and this)")
    ap.add_argument("--backend", default=None)
    ap.add_argument("--gpus", type=int, default=0, help="ranks of this run: without torchrun the script starts them itself; a rank count "
                    "other than this is an error (0: whatever the environment says -- the torchrun form of rounds 1-4)")
    args = ap.parse_args(argv)
    if args.gpus and script is not False:
        launch.ensure_ranks(args.gpus, script or os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), backend=args.backend,
                            metric=METRIC)
    launch.begin(METRIC)
    if args.gpus:
        launch.require_world(args.gpus, sharding.env_rank_world()[2])
    if args.synthetic:                                          # (before the GPU / process group exist: it forks)
        root = os.environ.get("TGN_SYNTH_DIR") or os.path.join(tempfile.gettempdir(), f"tgn_synth_{args.synthetic}")
        env_rank, _, env_world = sharding.env_rank_world()
        write_synthetic(root, args.synthetic, env_rank, env_world)
    rank, local_rank, world, device = sharding.init_from_env(backend=args.backend)
    if args.synthetic:
        sharding.barrier()
        args.source_obj_data_path, args.source_json_data_path = os.path.join(root, "obj"), os.path.join(root, "json")
    pairs = preprocess.list_scans(args.source_obj_data_path, args.source_json_data_path)
    warm = 0.0
    launch.stage("calibrate")
    if fps_batch is None and device.type == "cuda":
        # first use of the GPU by this process (context, code object, allocator) -- ~0.3 s that belong to no scan
        import time
        from toothgroupnetwork_amd import resample
        t0 = time.perf_counter()
        resample.fps_batch([synth.arch_cloud(30000, seed=1, with_normals=False)], preprocess.N_SAMPLED)
        warm = time.perf_counter() - t0
    launch.stage("timed")
    res = preprocess.preprocess_sharded(pairs, args.save_data_path, rank, world, batch=args.batch, fps_batch=fps_batch,
                                        device=device if device.type == "cuda" else None)
    launch.stage("report")
    who = launch.describe_ranks(device)
    if rank == 0:
        launch.emit({"metric": METRIC, "value": res["meshes_per_s"],
                     "unit": "scans/s", "n_gpus": world, "gpu_warmup_s_excluded": round(warm, 3), **res, **who})
    launch.shutdown()


if __name__ == "__main__":
    launch.guard(main)
