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


def cpulist_string(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8' (the sysfs syntax parse_cpulist reads)"""
    cpus = sorted(set(int(c) for c in cpus))
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)


def pin_record():
    """what pin_to_gpu_numa did for this process: {"pinned": bool, "cpus": cpulist or None, "n": count, "reason": why not}
    (bench.py / the sharded runners print it per rank)"""
    return dict(_PIN.get("record", {"pinned": False, "cpus": None, "n": _affinity_count(), "reason": "not attempted (single rank or no GPU)"}))


def pin_to_gpu_numa(device_index, sysfs="/sys/bus/pci/devices", bdf=None):
    """Restrict this process's host threads to the CPUs of the NUMA node its GPU hangs off (the PCI device's
    local_cpulist): one process per GPU on an 8-GPU node otherwise lets the OBJ parser / OpenMP threads of eight ranks
    wander over both sockets (the preprocess runner is half host time).  Best effort and never fatal: returns the CPU
    list, or None -- the affinity mask is then left as it was -- when sysfs has no answer or the node's CPUs do not
    intersect the mask the process was given (a container pinned elsewhere); what happened is kept for pin_record().
    TGN_NUMA_PIN=0 disables it."""
    def give_up(reason):
        _PIN["record"] = {"pinned": False, "cpus": None, "n": _affinity_count(), "reason": reason}
        return None

    if os.environ.get("TGN_NUMA_PIN", "1") == "0":
        return give_up("disabled (TGN_NUMA_PIN=0)")
    if not hasattr(os, "sched_setaffinity"):
        return give_up("no sched_setaffinity on this platform")
    try:
        if bdf is None:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        text = open(os.path.join(sysfs, bdf, "local_cpulist")).read()
        cpus = parse_cpulist(text)
        current = os.sched_getaffinity(0)
        allowed = sorted(set(cpus) & set(current))
        if not allowed:
            return give_up(f"local_cpulist of {bdf} ({text.strip() or 'empty'}) does not intersect the affinity mask "
                           f"({cpulist_string(current)}): left unpinned")
        _PIN.setdefault("before", len(current))
        os.sched_setaffinity(0, allowed)
        _PIN["after"] = len(allowed)
        _PIN["record"] = {"pinned": True, "cpus": cpulist_string(allowed), "n": len(allowed), "reason": None, "pci_bus_id": bdf}
        return allowed
    except Exception as e:  # noqa: BLE001
        return give_up(f"{type(e).__name__}: {str(e)[:160]}")


# the process group the ONE collective of a sharded run goes over, and how it came about
_GROUP = {"backend": None, "note": None}


def backend_description():
    """"nccl", "gloo", or "gloo (rccl init failed: ...)" when the RCCL group could not be brought up and the <= 1 KB gather
    runs over gloo instead (the data path has no collective: the measured throughput is the same run either way)"""
    if not (dist.is_available() and dist.is_initialized()):
        return "none (single process)"
    b = _GROUP["backend"] or str(dist.get_backend())
    return f"{b} ({_GROUP['note']})" if _GROUP["note"] else b


def _agree(run_dir, name, rank, world, ok, timeout_s):
    """file-based agreement of the node's ranks (no collective -- this runs when the collective layer is what failed):
    every rank writes `<name>_<rank>` = "1" / "0" into the shared run directory and waits for all of them; returns the
    list of flags, None for ranks that did not answer in time"""
    import time
    try:
        with open(os.path.join(run_dir, f"{name}_{rank}.tmp"), "w") as f:
            f.write("1" if ok else "0")
        os.replace(os.path.join(run_dir, f"{name}_{rank}.tmp"), os.path.join(run_dir, f"{name}_{rank}"))
    except OSError:
        return [None] * world
    deadline = time.time() + timeout_s
    while True:
        flags = []
        for r in range(world):
            try:
                flags.append(open(os.path.join(run_dir, f"{name}_{r}")).read().strip() == "1")
            except OSError:
                flags.append(None)
        if all(f is not None for f in flags) or time.time() > deadline:
            return flags
        time.sleep(0.05)


def _init_rccl(rank, world, device, timeout):
    """backend "nccl" with an eager communicator and one probe collective; raises whatever RCCL raises"""
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, timeout=timeout, device_id=device)
    probe = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(probe)
    torch.cuda.synchronize(device)
    if int(probe.item()) != world:
        raise RuntimeError(f"probe all_reduce over RCCL returned {probe.item()} instead of {world}")


def _init_gloo(rank, world, timeout, run_dir=None, fallback=False):
    """backend "gloo".  As the fallback of a failed RCCL init the rendezvous goes through a FileStore in the run directory:
    the TCP store may be in whatever state the failed attempt left it in, and the ranks need not have failed at the same point"""
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # (the container hostname may not resolve)
    if fallback and run_dir:
        store = dist.FileStore(os.path.join(run_dir, "gloo_fallback_store"), world)
        dist.init_process_group(backend="gloo", store=store, rank=rank, world_size=world, timeout=timeout, group_name="tgn_gloo_fallback")
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=timeout)


def bring_up_group(rank, world, device, backend, timeout_s):
    """the process group of a sharded run (see init_from_env): RCCL with a time limit and a probe, gloo for everybody if RCCL
    does not come up on every rank"""
    import datetime

    from . import launch
    timeout = datetime.timedelta(seconds=timeout_s)
    _GROUP["backend"], _GROUP["note"] = backend, None
    if backend == "nccl":
        launch.stage("rccl_init", dist_timeout_s=timeout_s)
        why = None
        try:
            _init_rccl(rank, world, device, timeout)
        except Exception as e:  # noqa: BLE001
            why = f"{type(e).__name__}: {str(e).strip().splitlines()[-1] if str(e).strip() else ''}"[:240]
        run_dir = launch.run_dir()
        flags = _agree(run_dir, "rccl_ok", rank, world, why is None, timeout_s)
        if not all(f is True for f in flags):
            bad = [r for r, f in enumerate(flags) if f is not True]
            if why is None:
                why = f"rank(s) {bad} could not bring RCCL up"
            launch.note(rccl_error=why, rccl_failed_ranks=bad)
            if dist.is_initialized():
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
            launch.stage("gloo_init")
            _init_gloo(rank, world, timeout, run_dir=run_dir, fallback=True)
            _GROUP["backend"], _GROUP["note"] = "gloo", f"rccl init failed: {why}"
    else:
        launch.stage("gloo_init" if backend == "gloo" else "rccl_init", dist_timeout_s=timeout_s)
        if backend == "gloo":
            _init_gloo(rank, world, timeout)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    launch.stage("setup", backend=backend_description())


def init_from_env(backend=None, timeout_s=None):
    """Initialise torch.distributed from the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, local_rank, world, device).  world == 1 needs no process group.  With several ranks on one node every
    rank's host threads are pinned to its GPU's NUMA node (pin_to_gpu_numa; best effort, recorded).

    Backend "nccl" (RCCL; the default on GPUs) is brought up with a time limit (TGN_DIST_TIMEOUT_S, default 180 s) and probed
    with one all_reduce.  If that raises on any rank, ALL ranks -- they agree through files in the run directory, not
    through the layer that just failed -- drop it and bring up gloo instead; `backend_description()` then reads
    "gloo (rccl init failed: ...)".  The sharded runners exchange nothing while computing and gather <= 1 KB at the end,
    so the run and its throughput are the same; only the transport of that one vector differs.  A rank that cannot
    join either group fails with stage "rccl_init" / "gloo_init" in its status record (launch.py)."""
    from . import launch
    rank, local_rank, world = env_rank_world()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        device = torch.device("cuda", torch.cuda.current_device())
        if world > 1:
            launch.stage("numa_pin")
            pin_to_gpu_numa(device.index)
            launch.note(numa_pin=pin_record())
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if use_cuda else "gloo"  # "nccl" is RCCL on ROCm
        if timeout_s is None:
            timeout_s = float(os.environ.get("TGN_DIST_TIMEOUT_S", "180"))
        bring_up_group(rank, world, device, backend, timeout_s)
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
