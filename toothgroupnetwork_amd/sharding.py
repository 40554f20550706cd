"""Multi-GPU execution of the hot path: independent meshes shard across ranks, nothing is exchanged
while computing, and ONE small collective gathers per-rank counters at the end.

The reference has no distributed code at all (single process, batch 1: runner.py:28-37,
preprocess_data.py:35 is a serial loop over scans).  On an 8-GPU MI355X node the natural unit of
parallelism is the mesh: one process per GPU (``torch.distributed``, backend "nccl" == RCCL over xGMI),
rank r works on ``shard_indices(n_items, r, world)``, and the per-rank metric vectors (mesh count,
seconds, checksums, loss sums -- LossMeter fields, loss_meter.py:9-20) are combined with a single
``all_gather_into_tensor`` of a <= 1 KB fp64 vector, which is latency-bound on any fabric.
On CPU (tests) the same code runs over the gloo backend.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous split: ranks < n_items % world get one extra item. Returns (start, stop)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(int(n_items), world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_indices(n_items, rank, world, mode="contiguous"):
    """Indices of the items rank `rank` owns. 'round_robin' balances sorted-by-size lists."""
    if mode == "contiguous":
        s, e = shard_range(n_items, rank, world)
        return list(range(s, e))
    if mode == "round_robin":
        if world <= 0 or not (0 <= rank < world):
            raise ValueError(f"bad rank/world {rank}/{world}")
        return list(range(rank, int(n_items), world))
    raise ValueError(f"unknown shard mode {mode!r}")


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist syntax) -> sorted list of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def effective_cpus(cgroup_root="/sys/fs/cgroup"):
    """Host cores this process can really use: the affinity mask, cut down by the container's CPU bandwidth quota (cgroup v2
    `cpu.max` = "quota period", v1 `cpu.cfs_quota_us` / `cpu.cfs_period_us`).  os.cpu_count() reports the machine -- 256 on the
    MI355X boxes whose sandbox grants 16 cores' worth of time -- and sizing thread pools by it oversubscribes the quota: the
    loaders of the preprocess runner then throttle each other (measured: 1 240 scans/s from 16 loader threads, 840 from 128;
    profiles/r04_preprocess_host_scaling.txt)."""
    n = _affinity_count()
    quota = cpu_quota(cgroup_root)
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(n, 1)


def _affinity_count():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_quota(cgroup_root="/sys/fs/cgroup"):
    """cores' worth of CPU time the container may use (cgroup v2 `cpu.max`, v1 cfs quota / period), or None if unlimited"""
    quota = None
    try:
        q, period = open(os.path.join(cgroup_root, "cpu.max")).read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")).read())
            period = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    return quota


_PIN = {}   # affinity-mask sizes before / after pin_to_gpu_numa narrowed this process's mask


def cpus_for_this_rank(local_world, cgroup_root="/sys/fs/cgroup"):
    """Host cores one of `local_world` ranks of a node should size its thread pools by.  The CPU quota is shared by all ranks of
    the node; the affinity mask, once pin_to_gpu_numa has narrowed it to the GPU's NUMA node, only by the ranks whose GPUs hang
    off that node (local_world x mask-after / mask-before of them, GPUs being spread evenly over the nodes) -- dividing the
    narrowed mask by ALL ranks counted the split twice (2 x 64 cores, 8 ranks: 8 loaders instead of 16)."""
    local_world = max(int(local_world), 1)
    aff = _affinity_count()
    before = max(_PIN.get("before", aff), aff)
    sharing = max(1, min(local_world, int(round(local_world * aff / before))))
    n = aff // sharing
    quota = cpu_quota(cgroup_root)
    if quota is not None:
        n = min(n, int(quota + 0.5) // local_world)
    return max(n, 1)


def pin_to_gpu_numa(device_index, sysfs="/sys/bus/pci/devices"):
    """Restrict this process's host threads to the CPUs of the NUMA node its GPU hangs off (the PCI device's
    local_cpulist): one process per GPU on an 8-GPU node otherwise lets the OBJ parser / OpenMP threads of eight ranks
    wander over both sockets (the preprocess runner is half host time).  Best effort: returns the CPU list, or None when
    sysfs has no answer; TGN_NUMA_PIN=0 disables it."""
    if os.environ.get("TGN_NUMA_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        cpus = parse_cpulist(open(os.path.join(sysfs, bdf, "local_cpulist")).read())
        current = os.sched_getaffinity(0)
        allowed = sorted(set(cpus) & set(current))
        if not allowed:
            return None
        _PIN.setdefault("before", len(current))
        os.sched_setaffinity(0, allowed)
        _PIN["after"] = len(allowed)
        return allowed
    except Exception:
        return None


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, local_rank, world, device).  world == 1 needs no process group.  With several ranks on one node every
    rank's host threads are pinned to its GPU's NUMA node (pin_to_gpu_numa)."""
    rank, local_rank, world = env_rank_world()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        device = torch.device("cuda", torch.cuda.current_device())
        if world > 1:
            pin_to_gpu_numa(device.index)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if use_cuda else "gloo"  # "nccl" is RCCL on ROCm
        kwargs = {}
        if use_cuda and backend == "nccl":
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world, device


def gather_metrics(vec, device=None):
    """All ranks contribute a 1-D fp64 vector of equal length; every rank receives the (world, k) matrix.
    This is the only collective of a sharded run."""
    vec = torch.as_tensor(vec, dtype=torch.float64)
    live = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if live and dist.get_backend() == "gloo":
        device = torch.device("cpu")            # gloo ranks (CPU tests, several ranks sharing one GPU): host tensors
    if device is not None:
        vec = vec.to(device)
    if not live:
        return vec.reshape(1, -1)
    world = dist.get_world_size()
    out = torch.empty(world * vec.numel(), dtype=torch.float64, device=vec.device)
    dist.all_gather_into_tensor(out, vec.contiguous())
    return out.reshape(world, -1)


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device=None):
    """Max of a python float over ranks (bench timing)."""
    m = gather_metrics([float(value)], device=device)
    return float(m.max().item())


def run_sharded(items, fn, rank, world, mode="contiguous", device=None):
    """Apply fn(item) -> dict of floats to this rank's shard; gather {key: sum over all ranks} plus counts.
    Keys must be identical on every rank (fixed metric schema)."""
    mine = shard_indices(len(items), rank, world, mode)
    sums = {}
    for i in mine:
        res = fn(items[i]) or {}
        for k, v in res.items():
            sums[k] = sums.get(k, 0.0) + float(v)
    keys = sorted(sums.keys())
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, keys)
        keys = sorted(set().union(*[set(k) for k in gathered]))
    vec = [float(len(mine))] + [sums.get(k, 0.0) for k in keys]
    mat = gather_metrics(vec, device=device)
    total = mat.sum(0).cpu().tolist()
    return {"count": int(round(total[0])), **{k: total[1 + j] for j, k in enumerate(keys)},
            "per_rank_count": [int(round(c)) for c in mat[:, 0].cpu().tolist()]}
