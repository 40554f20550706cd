"""Starting the ranks of a one-node multi-GPU run, and recording who took part.

The reference is a single process (runner.py:28-37, preprocess_data.py:35); here a multi-GPU run is one process per GPU
under ``torch.distributed`` (backend "nccl" = RCCL over xGMI).  Two rules make a silently single-GPU measurement impossible:

* ``ensure_ranks(n, script, argv)`` -- a script asked for n > 1 GPUs that was NOT started by torchrun (no WORLD_SIZE in its
  environment) replaces itself by ``python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
  --master-port <free port> script argv``: ``python bench.py --gpus 8`` and the torchrun form are the same run.
* ``require_world(n, world)`` -- the ranks a script ends up with must equal what it was asked for; anything else exits with
  status 2 (an error, not a warning).

``describe_ranks`` gathers, once, outside any timed region, each rank's device (index, PCI bus id, uuid, name), the
backend string and the RCCL version, so that "did RCCL see N distinct GPUs" is answerable from the JSON line of a run.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_command(n, script, argv, port=None, python=None):
    """the torchrun command line `ensure_ranks` replaces the process with (a list; tests inspect it)"""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def launched_by_torchrun(env=None):
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def ensure_ranks(n, script, argv, backend=None, env=None, execv=None):
    """n <= 1 or already under torchrun: returns.  Otherwise replaces this process by n ranks of `script argv` (never
    returns).  With the RCCL backend the node must have n GPUs -- asking for more exits with status 2 before anything is
    started (gloo lets several ranks share a GPU: tests of the N > 1 branch on a one-GPU box)."""
    env = os.environ if env is None else env
    if n <= 1 or launched_by_torchrun(env):
        return
    if backend in (None, "nccl") and torch.cuda.is_available() and torch.cuda.device_count() < n:
        raise SystemExit(f"error: --gpus {n} but this node has {torch.cuda.device_count()} GPU(s) visible "
                         f"(HIP_VISIBLE_DEVICES={env.get('HIP_VISIBLE_DEVICES', '<unset>')})")
    cmd = spawn_command(n, script, list(argv))
    sys.stdout.flush()
    sys.stderr.flush()
    (execv or os.execve)(cmd[0], cmd, {**env, "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                                       "TGN_SELF_SPAWNED": "1"})


def require_world(n, world):
    if world != max(int(n), 1):
        print(f"error: --gpus {n} but {world} rank(s) are running (WORLD_SIZE={os.environ.get('WORLD_SIZE', '<unset>')}): "
              f"refusing to report a {world}-rank measurement as a {n}-GPU one", file=sys.stderr)
        raise SystemExit(2)


def _this_rank(device):
    info = {"rank": dist.get_rank() if dist.is_initialized() else 0, "pid": os.getpid(), "host": socket.gethostname(),
            "device_index": None, "pci_bus_id": None, "uuid": None, "name": None}
    if device is not None and device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        info["device_index"] = device.index if device.index is not None else torch.cuda.current_device()
        try:
            info["pci_bus_id"] = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except AttributeError:
            pass
        info["uuid"] = str(getattr(pr, "uuid", "")) or None
        info["name"] = pr.name
    return info


def rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        return None


def describe_ranks(device):
    """{"ranks": [one record per rank, in rank order], "backend", "rccl_version", "distinct_devices", "self_spawned"} --
    the same on every rank.  One all_gather_object at set-up / tear-down time, never inside a timed region."""
    mine = _this_rank(device)
    live = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if live:
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, mine)
        backend = str(dist.get_backend())
    else:
        ranks, backend = [mine], "none (single process)"
    devs = {(r["host"], r["pci_bus_id"] or r["uuid"] or r["device_index"]) for r in ranks}
    return {"ranks": ranks, "backend": backend, "rccl_version": rccl_version() if backend == "nccl" or not live else None,
            "distinct_devices": len(devs), "self_spawned": os.environ.get("TGN_SELF_SPAWNED") == "1"}


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
