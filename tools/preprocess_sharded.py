#!/usr/bin/env python3
"""BASELINE.json config 5: preprocess_data.py over a directory of scans, sharded over the GPUs of one node.

    python tools/preprocess_sharded.py --source_obj_data_path OBJ --source_json_data_path JSON --save_data_path OUT
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        tools/preprocess_sharded.py --synthetic 512 --save_data_path /tmp/out

Same three path arguments as the reference script (preprocess_data.py:8-11).  One process per GPU (RANK / LOCAL_RANK /
WORLD_SIZE from the environment), rank r takes every world-th scan, nothing is exchanged while working, ONE
all_gather of a 7-number fp64 vector at the end (RCCL over xGMI; gloo on CPU).  --synthetic N writes N synthetic raw
scans (about 100 000 vertices each, seeded) into a temporary directory first, for benchmarking without the dataset.
Rank 0 prints one JSON line: scans/s over the whole job, time split into host load (OBJ parse + normals) and FPS."""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toothgroupnetwork_amd import preprocess, sharding, synth  # noqa: E402


def write_synthetic(root, n, rank, world):
    for i in range(rank, n, world):
        name, jaw = f"SYN{i:05d}_{'upper' if i % 2 == 0 else 'lower'}", ("upper" if i % 2 == 0 else "lower")
        os.makedirs(os.path.join(root, "obj", name), exist_ok=True)
        os.makedirs(os.path.join(root, "json", name), exist_ok=True)
        nu, nv = 330 + (i % 7) * 10, 300
        with open(os.path.join(root, "obj", name, name + ".obj"), "w") as f:
            f.write(synth.obj_text(nu, nv, 1000 + i, "plain", with_tail=False))
        with open(os.path.join(root, "json", name, name + ".json"), "w") as f:
            json.dump({"jaw": jaw, "labels": synth.fdi_labels(nu * nv, jaw, i)}, f)


def main(argv=None, fps_batch=None):
    """fps_batch: the sampler handed to preprocess.preprocess_sharded (None = the GPU kernel, the only one this runner
    knows; tests of the control flow pass their own callable through tests/sharded_launcher.py)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--source_obj_data_path", default=None)
    ap.add_argument("--source_json_data_path", default=None)
    ap.add_argument("--save_data_path", default="data_preprocessed_path")
    ap.add_argument("--synthetic", type=int, default=0, help="generate this many synthetic raw scans instead of reading a dataset")
    ap.add_argument("--batch", type=int, default=32, help="scans per FPS launch (a launch costs ~50 ms whatever its size: the sampling chain of one raw scan)")
    ap.add_argument("--backend", default=None)
    args = ap.parse_args(argv)
    rank, local_rank, world, device = sharding.init_from_env(backend=args.backend)
    tmp = None
    if args.synthetic:
        root = os.environ.get("TGN_SYNTH_DIR") or os.path.join(tempfile.gettempdir(), f"tgn_synth_{args.synthetic}")
        write_synthetic(root, args.synthetic, rank, world)
        sharding.barrier()
        args.source_obj_data_path, args.source_json_data_path = os.path.join(root, "obj"), os.path.join(root, "json")
    pairs = preprocess.list_scans(args.source_obj_data_path, args.source_json_data_path)
    res = preprocess.preprocess_sharded(pairs, args.save_data_path, rank, world, batch=args.batch, fps_batch=fps_batch,
                                        device=device if device.type == "cuda" else None)
    if rank == 0:
        print(json.dumps({"metric": "preprocessed scans/sec (OBJ parse + normals + FPS N_raw->24000 + npy)", "value": res["meshes_per_s"],
                          "unit": "scans/s", "n_gpus": world, **res}))
    import torch.distributed as dist
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
