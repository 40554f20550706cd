"""Starting the ranks of a one-node multi-GPU run, recording who took part -- and what went wrong when something did.

The reference is a single process (runner.py:28-37, preprocess_data.py:35); here a multi-GPU run is one process per GPU
under ``torch.distributed`` (backend "nccl" = RCCL over xGMI).  Rules:

* ``ensure_ranks(n, script, argv)`` -- a script asked for n > 1 GPUs that was NOT started by torchrun (no WORLD_SIZE in its
  environment) runs ``python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port
  <free port> script argv`` as a supervised child and exits with its status: ``python bench.py --gpus 8`` and the torchrun
  form are the same run.
* ``require_world(n, world)`` -- the ranks a script ends up with must equal what it was asked for; anything else exits with
  status 2 (an error, not a warning).
* **One JSON line, always.**  A runner calls ``begin(metric)`` first and prints its result through ``emit``.  Every rank
  keeps a small status file in a run directory shared by the ranks of the node (``TGN_RUN_DIR``, else
  /tmp/tgn_run_<MASTER_PORT>_<parent pid>): the stage it is in ("spawn", "numa_pin", "rccl_init", "calibrate", "timed",
  "gather", ...), its device, the CPUs it was pinned to, and -- if it fails -- the error.  Rank 0 also starts a *watcher*
  (a tiny python child holding the read end of a pipe): when rank 0 goes away without having printed its line -- an
  exception, a SIGTERM from torchrun after another rank died, a SIGKILL, an abort inside RCCL -- the watcher prints
  ``{"error": ..., "stage": ..., "failed_rank": ..., "ranks": [...]}`` built from the status files.  ``guard(main)`` turns an
  exception on rank 0 into the same line directly.  The exit status is non-zero in every such case.

``describe_ranks`` gathers, once, outside any timed region, each rank's device (index, PCI bus id, uuid, name), the
backend string and the RCCL version, so that "did RCCL see N distinct GPUs" is answerable from the JSON line of a run.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time
import traceback

import torch
import torch.distributed as dist

STAGES = ("spawn", "numa_pin", "rccl_init", "gloo_init", "setup", "calibrate", "timed", "gather", "report", "done")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_command(n, script, argv, port=None, python=None):
    """the torchrun command line `ensure_ranks` runs (a list; tests inspect it)"""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def launched_by_torchrun(env=None):
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


# ---------------------------------------------------------------------------------------------------------------------
# run record: one status file per rank, one JSON line per run
# ---------------------------------------------------------------------------------------------------------------------
class _Run:
    def __init__(self):
        self.dir = None
        self.rank, self.world = 0, 1
        self.rec = {}
        self.metric = None
        self.printed = False
        self.watcher = None


_RUN = _Run()


def run_dir(env=None, create=True):
    """the directory the ranks of this run share: TGN_RUN_DIR, else one named after the rendezvous port and the common
    parent (the torchrun agent), else -- single process -- after this pid"""
    env = os.environ if env is None else env
    d = env.get("TGN_RUN_DIR")
    if not d:
        key = f"{env.get('MASTER_PORT', 'single')}_{os.getppid() if launched_by_torchrun(env) else os.getpid()}"
        d = os.path.join(tempfile.gettempdir(), f"tgn_run_{key}")
    if create:
        os.makedirs(d, exist_ok=True)
    return d


def _write_json(path, obj):
    tmp = f"{path}.tmp{os.getpid()}"
    try:
        with open(tmp, "w") as f:
            json.dump(obj, f)
        os.replace(tmp, path)
    except OSError:
        pass


def _flush_record():
    if _RUN.dir is not None:
        _write_json(os.path.join(_RUN.dir, f"rank{_RUN.rank}.json"), _RUN.rec)


_WATCHER = r'''
import glob, json, os, signal, sys, time
for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(s, signal.SIG_IGN)
d, world, metric = sys.argv[1], int(sys.argv[2]), sys.argv[3]
try:
    while os.read(0, 4096):
        pass
except OSError:
    pass
if os.path.exists(os.path.join(d, "printed")):
    sys.exit(0)
time.sleep(float(os.environ.get("TGN_WATCHER_GRACE_S", "1.0")))     # the other ranks' last words
if os.path.exists(os.path.join(d, "printed")):
    sys.exit(0)
ranks = []
for p in sorted(glob.glob(os.path.join(d, "rank*.json"))):
    try:
        ranks.append(json.load(open(p)))
    except Exception:
        pass
ranks.sort(key=lambda r: r.get("rank", 0))
bad = [r for r in ranks if r.get("error")]
first = min(bad, key=lambda r: r.get("failed_at", 0.0)) if bad else None
r0 = next((r for r in ranks if r.get("rank") == 0), {})
line = {"error": (first["error"] if first else "rank 0 ended without printing its result (killed or aborted); no rank recorded an error"),
        "stage": (first or r0).get("stage", "spawn"), "failed_rank": first["rank"] if first else None,
        "metric": metric, "n_gpus": world, "ranks_reporting": len(ranks), "ranks": ranks, "reported_by": "watcher"}
try:
    open(os.path.join(d, "printed"), "w").write("watcher")
except OSError:
    pass
sys.stdout.write(json.dumps(line) + "\n")
sys.stdout.flush()
'''


def begin(metric, env=None, watcher=True):
    """Call first in a runner's main(): opens this rank's status record and, on rank 0, the watcher that guarantees the
    run's one JSON line.  Idempotent."""
    env = os.environ if env is None else env
    if _RUN.dir is not None:
        return _RUN.dir
    _RUN.rank, _RUN.world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
    _RUN.metric = metric
    _RUN.dir = run_dir(env)
    _RUN.rec = {"rank": _RUN.rank, "pid": os.getpid(), "host": socket.gethostname(), "stage": "spawn", "stages": ["spawn"],
                "started": time.time()}
    # a directory reused by a later run of the same port / parent: drop what the earlier run left under this rank's names
    for name in ([f"rccl_ok_{_RUN.rank}"] + (["printed", "gloo_fallback_store"] if _RUN.rank == 0 else [])):
        try:
            os.unlink(os.path.join(_RUN.dir, name))
        except OSError:
            pass
    _flush_record()
    if _RUN.rank == 0 and watcher and env.get("TGN_WATCHER", "1") != "0":
        try:
            _RUN.watcher = subprocess.Popen([sys.executable, "-c", _WATCHER, _RUN.dir, str(_RUN.world), str(metric)],
                                            stdin=subprocess.PIPE, close_fds=True)
        except OSError:
            _RUN.watcher = None
    return _RUN.dir


def stage(name, **notes):
    """this rank is now in stage `name` (recorded in its status file); notes are merged into the record"""
    if _RUN.dir is None:
        return
    _RUN.rec["stage"] = name
    _RUN.rec.setdefault("stages", []).append(name)
    _RUN.rec.update(notes)
    _flush_record()


def note(**notes):
    if _RUN.dir is None:
        return
    _RUN.rec.update(notes)
    _flush_record()


def current_stage():
    return _RUN.rec.get("stage", "spawn")


def rank_records():
    """the status records of every rank that has written one (best effort, rank order)"""
    out = []
    if _RUN.dir is None:
        return out
    for r in range(max(_RUN.world, 1)):
        try:
            out.append(json.load(open(os.path.join(_RUN.dir, f"rank{r}.json"))))
        except (OSError, ValueError):
            pass
    return out


def _close_watcher():
    w, _RUN.watcher = _RUN.watcher, None
    if w is not None:
        try:
            w.stdin.close()
            w.wait(timeout=10)
        except Exception:
            pass


def emit(obj):
    """print the run's ONE JSON line (rank 0 only; a second call is ignored) and tell the watcher it is out"""
    if _RUN.rank != 0 or _RUN.printed:
        return False
    _RUN.printed = True
    if _RUN.dir is not None:
        if os.path.exists(os.path.join(_RUN.dir, "printed")):
            return False                                    # the watcher of an earlier incarnation got there first
        try:
            open(os.path.join(_RUN.dir, "printed"), "w").write("rank0")
        except OSError:
            pass
    sys.stdout.write(json.dumps(obj) + "\n")
    sys.stdout.flush()
    _close_watcher()
    return True


def fail(message, stage_name=None, code=1, exc=None):
    """record a failure of this rank (status file), print the error line if this is rank 0, and exit with `code`"""
    st = stage_name or current_stage()
    _RUN.rec.update({"stage": st, "error": str(message)[:600], "failed_at": time.time()})
    if exc is not None:
        _RUN.rec["traceback"] = "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-1500:]
    _flush_record()
    print(f"error [{st}] rank {_RUN.rank}: {message}", file=sys.stderr)
    if _RUN.rank == 0:
        time.sleep(0.2)                                     # a failure on every rank at once: let the others write theirs
        recs = rank_records() or [_RUN.rec]
        bad = [r for r in recs if r.get("error")]
        first = min(bad, key=lambda r: r.get("failed_at", 0.0)) if bad else _RUN.rec
        emit({"error": first.get("error", str(message)), "stage": first.get("stage", st), "failed_rank": first.get("rank", 0),
              "metric": _RUN.metric, "n_gpus": _RUN.world, "ranks_reporting": len(recs), "ranks": recs, "reported_by": "rank 0"})
    raise SystemExit(code if code else 1)


def guard(main, *args, **kwargs):
    """run main(*args): an exception (or a non-zero SystemExit) becomes this rank's recorded failure -- and, on rank 0, the
    run's error line -- instead of a bare traceback; the exit status stays non-zero"""
    try:
        return main(*args, **kwargs)
    except SystemExit as e:
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if code == 0 or _RUN.rec.get("error"):
            raise
        if _RUN.dir is None:
            raise
        msg = e.code if isinstance(e.code, str) else f"exit status {code}"
        try:
            fail(msg, code=code)
        except SystemExit:
            pass
        raise SystemExit(code)
    except BaseException as e:  # noqa: BLE001
        if _RUN.dir is None:
            raise
        traceback.print_exc()
        try:
            fail(f"{type(e).__name__}: {e}", exc=e)
        except SystemExit:
            pass
        raise SystemExit(1)


def _error_line_without_run(message, stage_name, metric, n):
    sys.stdout.write(json.dumps({"error": message, "stage": stage_name, "failed_rank": None, "metric": metric, "n_gpus": n,
                                 "ranks_reporting": 0, "ranks": [], "reported_by": "launcher"}) + "\n")
    sys.stdout.flush()


def ensure_ranks(n, script, argv, backend=None, env=None, run=None, metric=None):
    """n <= 1 or already under torchrun: returns.  Otherwise runs n ranks of `script argv` under torchrun as a supervised
    child and exits with its status (never returns).  With the RCCL backend the node must have n GPUs -- asking for more
    prints the error line (stage "spawn") and exits with status 2 before anything is started (gloo lets several ranks
    share a GPU: tests of the N > 1 branch on a one-GPU box).  If the children end without a result line -- torchrun could
    not start them, every rank died -- the supervisor prints the error line itself."""
    env = os.environ if env is None else env
    if n <= 1 or launched_by_torchrun(env):
        return
    if backend in (None, "nccl") and torch.cuda.is_available() and torch.cuda.device_count() < n:
        msg = (f"--gpus {n} but this node has {torch.cuda.device_count()} GPU(s) visible "
               f"(HIP_VISIBLE_DEVICES={env.get('HIP_VISIBLE_DEVICES', '<unset>')})")
        print("error: " + msg, file=sys.stderr)
        _error_line_without_run(msg, "spawn", metric, n)
        raise SystemExit(2)
    port = free_port()
    cmd = spawn_command(n, script, list(argv), port=port)
    d = env.get("TGN_RUN_DIR") or os.path.join(tempfile.gettempdir(), f"tgn_run_{port}_{os.getpid()}")
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; without it RCCL and CUDA-tensor sharing
    # across processes fail with `hipIpcGetMemHandle: invalid argument` (kept if the caller has set it)
    child_env = {**env, "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), "TGN_SELF_SPAWNED": "1",
                 "TGN_RUN_DIR": d}
    sys.stdout.flush()
    sys.stderr.flush()
    if run is not None:                                      # tests: inspect the command line instead of running it
        return run(cmd[0], cmd, child_env)
    os.makedirs(d, exist_ok=True)
    proc = subprocess.Popen(cmd, env=child_env)

    def forward(signum, _frame):
        try:
            proc.send_signal(signum)
        except OSError:
            pass

    for s in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(s, forward)
        except (ValueError, OSError):
            pass
    rc = proc.wait()
    if not os.path.exists(os.path.join(d, "printed")):
        time.sleep(float(env.get("TGN_WATCHER_GRACE_S", "1.0")) + 0.5)     # a watcher may be about to print
    if not os.path.exists(os.path.join(d, "printed")):
        recs = []
        for r in range(n):
            try:
                recs.append(json.load(open(os.path.join(d, f"rank{r}.json"))))
            except (OSError, ValueError):
                pass
        bad = [r for r in recs if r.get("error")]
        first = min(bad, key=lambda r: r.get("failed_at", 0.0)) if bad else None
        sys.stdout.write(json.dumps({
            "error": first["error"] if first else f"torchrun exited with status {rc} before rank 0 printed a result",
            "stage": first["stage"] if first else (recs[0].get("stage", "spawn") if recs else "spawn"),
            "failed_rank": first["rank"] if first else None, "metric": metric, "n_gpus": n, "ranks_reporting": len(recs),
            "ranks": recs, "reported_by": "supervisor"}) + "\n")
        sys.stdout.flush()
        rc = rc or 1
    raise SystemExit(rc)


def require_world(n, world):
    if world != max(int(n), 1):
        msg = (f"--gpus {n} but {world} rank(s) are running (WORLD_SIZE={os.environ.get('WORLD_SIZE', '<unset>')}): "
               f"refusing to report a {world}-rank measurement as a {n}-GPU one")
        print("error: " + msg, file=sys.stderr)
        if _RUN.dir is not None:
            fail(msg, stage_name="spawn", code=2)
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
    from . import sharding
    info["numa_pin"] = sharding.pin_record()
    return info


def rccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        return None


def describe_ranks(device):
    """{"ranks": [one record per rank, in rank order], "backend", "rccl_version", "distinct_devices", "self_spawned"} --
    the same on every rank.  One all_gather_object at set-up / tear-down time, never inside a timed region."""
    from . import sharding
    mine = _this_rank(device)
    note(device_index=mine["device_index"], pci_bus_id=mine["pci_bus_id"], name=mine["name"])
    live = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if live:
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, mine)
        backend = sharding.backend_description()
    else:
        ranks, backend = [mine], "none (single process)"
    devs = {(r["host"], r["pci_bus_id"] or r["uuid"] or r["device_index"]) for r in ranks}
    return {"ranks": ranks, "backend": backend, "rccl_version": rccl_version() if backend.startswith("nccl") or not live else None,
            "distinct_devices": len(devs), "self_spawned": os.environ.get("TGN_SELF_SPAWNED") == "1"}


def shutdown():
    stage("done")
    _close_watcher()
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
