#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json:

    meshes/sec (24k-pt FPS+ball_query+group fwd) at 1/2/4/8 GPU; HBM GB/s vs peak

A "step" is one pass of the hot path (toothgroupnetwork_amd.hotpath.HotPath, Shape A of SURVEY.md 8:
N=24000, npoint=[4096,1024,256], nsample=32, radii [0.05,0.1,0.2], D=[6,128,512]) over one batch of
synthetic scans that is already resident in HBM.  Multi-GPU = one process per GPU, every rank works on
its own batch (weak scaling; meshes are independent, no data-path collective), barrier + max-over-ranks
timing, one RCCL all_gather of the per-rank timing vector at the end.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` without torchrun starts its N ranks itself (toothgroupnetwork_amd.launch); a rank
count other than --gpus is an error.  Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel
(FPS level 1), timed with HIP events on the launch stream inside the timed region, against the measured
latency floor of one FPS iteration (tools/fps_floor.hip); `cpu_baseline` is the reference's CPU path restated
torch call for torch call (oracle/torch_cpu.py) on a bounded sample of the same workload, with the oracle's
C/OpenMP port beside it; `ranks` / `backend` / `rccl_version` say who took part.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # (toothgroupnetwork_amd/_lib.py: the schedule's streams each want a hardware queue)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import hotpath, launch, sharding, synth  # noqa: E402

METRIC = "meshes/sec (24k-pt FPS+ball_query+group fwd)"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable


_SCANS = {}


def make_inputs(B, device, seed, shape, unique=0):
    """B synthetic 24 000-point scans (xyz + normals) and synthetic level-2/3 features, resident in HBM.  Every scan of the batch is
    a distinct draw (FPS is data dependent and a step ends with its slowest workgroup: the maximum over B draws, not over a
    few repeated ones); `unique` > 0 tiles that many distinct scans instead (rounds 1-5 used 16: generation is ~20 ms per scan)."""
    n_unique = min(B, unique) if unique > 0 else B
    key = (n_unique, shape["n"], seed)
    if key not in _SCANS:                       # (tools/secondary_bench.py asks for the same batch several times: ~5 s per 256 scans)
        _SCANS.clear()
        _SCANS[key] = synth.scan_batch(n_unique, shape["n"], "arch", seed=seed)
    scans = _SCANS[key]
    scans = np.concatenate([scans] * ((B + n_unique - 1) // n_unique), axis=0)[:B]
    pts = torch.from_numpy(scans).to(device)
    xyz = pts[:, :, :3].contiguous()
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = [pts]
    for S, D in zip(shape["npoint"][:-1], shape["d"][1:]):
        f = torch.randn(n_unique, S, D, generator=g).repeat((B + n_unique - 1) // n_unique, 1, 1)[:B]
        feats.append(f.to(device).contiguous())
    return xyz, feats, scans


def cpu_baseline_c_port(scans, budget_meshes, shape):
    """The oracle's C restatement of the algorithm on a bounded sample, all host cores (OpenMP): the strong CPU line."""
    from oracle import cpu as O
    cores = O.num_threads()
    sample = scans[:budget_meshes]
    t0 = time.perf_counter()
    xyz = np.ascontiguousarray(sample[:, :, :3])
    pts = sample
    rng = np.random.default_rng(0)
    xyz_first = shape.get("xyz_first", True)
    for li, (S, r, K, D) in enumerate(zip(shape["npoint"], shape["radius"], shape["nsample"], shape["d"])):
        fidx = O.farthest_point_sample(xyz, S)
        new_xyz = O.index_points(xyz, fidx)
        for rb, kb in hotpath._branches(r, K):
            gidx = O.query_ball_point(rb, kb, xyz, new_xyz)
            O.group_points(xyz, new_xyz, pts, gidx, xyz_first=xyz_first)
        xyz = np.ascontiguousarray(new_xyz)
        nxt = shape["d"][li + 1] if li + 1 < len(shape["d"]) else 0
        pts = rng.standard_normal((sample.shape[0], S, nxt), dtype=np.float32) if nxt else None
    dt = time.perf_counter() - t0
    return {"value": sample.shape[0] / dt, "unit": "meshes/s", "cores": cores, "kind": "port",
            "sample": f"{sample.shape[0]} scans x {len(shape['npoint'])} levels through oracle/pointops_oracle.c, "
                      f"{cores} OpenMP threads, {dt:.1f} s",
            "note": "C/OpenMP restatement of the algorithm (oracle/): far stronger than the reference's own torch-CPU path"}


def cpu_baseline(scans, shape, budget_s=8.0, max_meshes=4):
    """The reference's CPU path of this workload -- `farthest_point_sample_np` (start forced to 0), `query_ball_point`, the
    gather / centre / cat lines of `sample_and_group` (pointnet2_utils.py:103-169) -- restated torch call for torch call in
    oracle/torch_cpu.py (pinned bit for bit to fixtures the reference's own functions produced) and timed HERE, on the bench
    host's cores, one scan at a time like the reference's batch-1 loop: scans until `budget_s` is used (at least one).
    Single-branch shapes only (Shape A); the C/OpenMP port rides along as `c_openmp_port`."""
    from oracle import torch_cpu as TC
    # all the cores the process can really use, and one thread.  "All" is the affinity mask cut down by the
    # container's CPU quota (sharding.effective_cpus): torch's default of one thread per hardware thread (128 here) under a
    # 16-core quota spends its time being throttled -- 13.6 s for level 1's sampling instead of ~2 s
    cores = sharding.effective_cpus()
    before = torch.get_num_threads()
    per_scan, levels = [], None
    t_all = time.perf_counter()
    try:
        torch.set_num_threads(cores)
        for i in range(max_meshes):
            t0 = time.perf_counter()
            lv = TC.headline_levels(scans[i % scans.shape[0]], shape["npoint"], shape["radius"], shape["nsample"], shape["d"], seed=i)
            per_scan.append(time.perf_counter() - t0)
            levels = lv if levels is None else [tuple(a + b for a, b in zip(x, y)) for x, y in zip(levels, lv)]
            if time.perf_counter() - t_all > budget_s:
                break
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        TC.headline_levels(scans[0], shape["npoint"], shape["radius"], shape["nsample"], shape["d"], seed=0)
        one_thread_s = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    n = len(per_scan)
    dt = float(np.median(per_scan))
    return {"value": 1.0 / dt, "unit": "meshes/s", "cores": cores, "kind": "port",
            "sample": f"{n} scan(s) x {len(shape['npoint'])} levels, one at a time, {cores} torch threads, median {dt:.2f} s per scan; "
                      f"one more scan on 1 thread ({one_thread_s:.2f} s); {time.perf_counter() - t_all:.1f} s in all",
            "seconds_per_scan": [round(v, 3) for v in per_scan],
            "one_thread": {"value": 1.0 / one_thread_s, "unit": "meshes/s", "cores": 1},
            "seconds_per_level_fps_ball_group": [[round(v / n, 3) for v in x] for x in levels],
            "what": "the reference's own CPU path (pointnet2_utils.py:103-169) restated torch call for torch call (oracle/torch_cpu.py, "
                    "pinned to the reference's outputs by tests/test_oracle_golden.py); the reference checkout itself is not present on "
                    "the bench host"}


def fps_latency_floor():
    """The latency floor of one FPS iteration (tools/fps_floor.hip), measured on this GPU when the tool is built
    (tools/_bin/fps_floor, __graft_entry__.build()), else read from the committed run under profiles/."""
    import subprocess
    exe = os.path.join(REPO, "tools", "_bin", "fps_floor")
    try:
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return {**json.loads(line)["fps_floor"], "source": "tools/_bin/fps_floor run on this GPU after the timed region"}
    except Exception as e:  # noqa: BLE001
        try:
            for name in ("r06_fps_floor.txt", "r05_fps_floor.txt"):
                if not os.path.exists(os.path.join(REPO, "profiles", name)):
                    continue
                for l in open(os.path.join(REPO, "profiles", name)):
                    if l.startswith("{"):
                        return {**json.loads(l)["fps_floor"], "source": f"profiles/{name} (the tool did not run here: {type(e).__name__})"}
        except Exception:
            pass
        return {"error": f"fps_floor unavailable: {type(e).__name__}: {str(e)[:120]}"}


def _secondary():
    import importlib.util
    spec = importlib.util.spec_from_file_location("secondary_bench", os.path.join(REPO, "tools", "secondary_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_secondary(timeout_s):
    """tools/secondary_bench.py as a child process writing its dict to a file; {"error": ...} when it fails or overruns"""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".json", prefix="tgn_secondary_")
    os.close(fd)
    try:
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "secondary_bench.py"), "--json-out", path],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
        try:
            res = json.load(open(path))
        except Exception:
            res = {}
        if r.returncode != 0:
            res["error"] = f"secondary_bench.py exited with status {r.returncode}: {r.stderr[-400:]}"
        return res
    except subprocess.TimeoutExpired:
        try:
            res = json.load(open(path))      # (the child rewrites the file after every entry: keep what it finished)
        except Exception:
            res = {}
        res["error"] = f"secondary_bench.py did not finish within {timeout_s:.0f} s"
        return res
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def single_gpu_legs_only(args, world):
    """N > 1 measures the headline and nothing else: the CPU baseline, the secondary configurations and the prefix-identity
    re-run are single-GPU legs (they would run on rank 0 while the other ranks wait in a collective)"""
    if world > 1:
        args.secondary, args.cpu_meshes, args.no_alt = 0, 0, True


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="scans per step per GPU (one FPS workgroup per scan)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic scans per batch (0 = all of them distinct, the default; "
                    "16 = what rounds 1-5 tiled to 256)")
    ap.add_argument("--cpu-meshes", type=int, default=-1, help="CPU-baseline sample size (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-stride", type=int, default=4, help="per-kernel HIP events on every n-th timed step (each event is a barrier "
                    "packet in its queue: on every step they cost the pipelined schedule ~5 %%)")
    ap.add_argument("--pipeline", type=int, default=1, help="1: overlap the FPS chain of step k+1 with ball query / "
                    "grouping of step k on two HIP streams (double-buffered); 0: one stream")
    ap.add_argument("--fps-prefix", type=int, default=0, help="1: levels 2 and 3 use the FPS-of-an-FPS-result identity "
                    "(exact, certificate checked on the device) instead of iterating; reported separately, never the headline")
    ap.add_argument("--shape", default="A", choices=["A", "B"], help="A: the headline configuration (BASELINE.json config 2); "
                    "B: what the reference net instantiates (multi-scale grouping, SURVEY.md 8) -- not the headline")
    ap.add_argument("--fused", type=int, default=0, help="1: every level is a fused set-abstraction level with a two-layer shared MLP "
                    "([64,128], [256,512], [512,1024]; eval-mode BatchNorm folded): neither the grouped tensor nor a layer output of "
                    "size S*K is written -- a second, non-headline line bound by the fp32 matrix cores")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL; gloo lets several ranks share "
                    "one GPU, e.g. to exercise the N > 1 branch on a one-GPU box)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra with_fps_prefix_identity measurement (profiling runs)")
    ap.add_argument("--secondary", type=int, default=1, help="1 (default, N = 1 only): after the headline, also measure the non-headline "
                    "configurations of BASELINE.json (Shape B, the fused levels, kNN, large-cloud FPS, Point-Transformer forward, training "
                    "step), in a child process, and attach them as `secondary` (tools/secondary_bench.py; ~1-2 minutes); 0: skip")
    ap.add_argument("--group-max-blocks", type=int, default=None, help="bound the grouping kernels' grid (default: 256 in the phased schedule -- their "
                    "place beside the FPS level-1 workgroups --, unbounded on one stream); counter passes use --pipeline 0 --group-max-blocks 256")
    ap.add_argument("--grid-stream", default="own", choices=["F", "H", "own"], help="where the next step's level-1 ball-query grid is built in "
                    "phase 2: behind FPS level 3 (F), behind the last phase-2 query (H), or on a stream of its own")
    ap.add_argument("--fps23", default="bucket", choices=["bucket", "plain"], help="FPS levels 2-3 on the bucket-skipping kernel (few vector "
                    "instructions beside the ball queries) or the plain register-resident one")
    ap.add_argument("--secondary-timeout", type=float, default=420.0, help="seconds the secondary child process may take")
    args = ap.parse_args(argv)

    if argv is None:
        # `python bench.py --gpus N` (no torchrun): this process runs N ranks of the same command line under torchrun and exits with
        # their status (asking for more GPUs than the node has: the error line, status 2)
        launch.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:], backend=args.backend, metric=METRIC)
    launch.begin(METRIC)                            # this rank's status record; on rank 0 the watcher that guarantees ONE JSON line
    launch.require_world(args.gpus, sharding.env_rank_world()[2])    # a rank count other than --gpus is an error (status 2), never a warning
    if not torch.cuda.is_available():
        launch.fail("bench.py needs a ROCm GPU: the hot path has no CPU implementation", stage_name="spawn")
    rank, local_rank, world, device = sharding.init_from_env(backend=args.backend)   # stages numa_pin, rccl_init (gloo if RCCL cannot come up)
    # host threads: the cores this rank can really use (a 128-thread host under a 16-core quota runs small CPU ops 10x slower on 128
    # threads than on 16; torchrun's children get OMP_NUM_THREADS=1 anyway)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), sharding.cpus_for_this_rank(int(os.environ.get("LOCAL_WORLD_SIZE", world))))))
    who = launch.describe_ranks(device)             # per-rank device records + backend + RCCL version (set-up time, untimed)
    B = args.batch
    shape = hotpath.SHAPE_A if args.shape == "A" else hotpath.SHAPE_B
    single_gpu_legs_only(args, world)
    xyz, feats, scans = make_inputs(B, device, seed=100 + 1000 * rank, shape=shape, unique=args.unique)
    gopts = dict(fused=bool(args.fused), group_max_blocks=args.group_max_blocks, grid_stream=args.grid_stream, fps_low_valu=args.fps23 == "bucket")
    hp = hotpath.HotPath(B, device, shape=shape, pipeline=bool(args.pipeline), fps_prefix=bool(args.fps_prefix), **gopts)
    launch.stage("calibrate")
    for _ in range(max(args.warmup, 1)):                     # (the first run also measures the schedule's plan, hotpath.plan_schedule)
        hp.run(xyz, feats, inputs_on_current_stream=False)   # the synthetic scans are resident before any step
    torch.cuda.synchronize()
    if not args.no_kernel_timing:
        # inside the timed region only the dominant kernel is timed (two events per timed step around FPS level 1, on its stream);
        # the other kernels' table comes from a few extra steps AFTER the timed region (events around every launch cost ~4 %)
        hp.enable_kernel_timing(args.steps, stride=max(args.timing_stride, 1), only=("fps_l1",))

    launch.stage("timed")
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        hp.run(xyz, feats, inputs_on_current_stream=False, more=k + 1 < args.steps)   # (the last step is told that nothing follows it)
    torch.cuda.synchronize()
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    launch.stage("gather", seconds=round(elapsed, 6))
    per_rank_s = sharding.gather_metrics([elapsed], device=device).reshape(-1).cpu().tolist()   # the one collective of the run
    elapsed = max(per_rank_s)
    launch.stage("report")

    total_meshes = B * args.steps * world
    value = total_meshes / elapsed
    bytes_per_mesh, per_level = hotpath.algorithmic_bytes(fused=bool(args.fused), **shape)

    out = {
        "metric": METRIC,
        "value": value,
        "unit": "meshes/s",
        "n_gpus": world,
        "ranks": who["ranks"], "backend": who["backend"], "rccl_version": who["rccl_version"],
        "distinct_devices": who["distinct_devices"], "self_spawned": who["self_spawned"],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "per_rank_seconds": [round(v, 6) for v in per_rank_s],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": ("shape_B FUSED (NOT the headline configuration): the three multi-scale set-abstraction levels of the reference "
                                "PointNet++ net on 24000-pt scans, FPS + ball queries + chained two-layer MLP kernels") if (args.fused and args.shape == "B") else
                               ("shape_A FUSED (NOT the headline configuration): 24000-pt scans, npoint=[4096,1024,256], nsample=32, "
                                "radii=[0.05,0.1,0.2]; FPS + ball query + fused set-abstraction level (gather, centre, two-layer shared "
                                "MLP [9->64->128, 131->256->512, 515->512->1024] on the fp32 matrix cores, folded BatchNorms, ReLUs, max "
                                "over K); no grouped tensor and no layer output of size S*K is written, level l feeds level l+1") if args.fused else
                               ("shape_A: 24000-pt scans, npoint=[4096,1024,256], nsample=32, radii=[0.05,0.1,0.2], "
                                "D=[6,128,512]; FPS+ball_query+group forward, grouped tensors materialised") if args.shape == "A" else
                               ("shape_B (NOT the headline configuration): 24000-pt scans, npoint=[1024,512,256], multi-scale "
                                "radii [[.025,.05],[.05,.1],[.1,.2]], nsample [32,64], D=[6,256,1024]; FPS+ball_query+group forward"),
                   **({"fused_levels": "the reference network's own set-abstraction stack (pointnet_pp.py:13-15): per branch a two-layer "
                                       "shared MLP [9->128->128, 259->256->512, 1027->784->1024], one chained kernel per branch, the "
                                       "branches written side by side; nothing of size S*K is written"} if (args.fused and args.shape == "B") else {}),
                   "meshes_per_step_per_gpu": B, "distinct_scans_per_gpu": int(min(B, args.unique) if args.unique > 0 else B), "sharding": f"independent meshes x {world} ranks, no data-path collective",
                   "index_dtype": "int32",
                   "fps_levels_2_3": "identity shortcut (FPS of an FPS result; certificate checked on device)"
                   if args.fps_prefix else "iterated like level 1",
                   "schedule": hp.describe_schedule()},
        "path_hbm": {"algorithmic_bytes_per_mesh": bytes_per_mesh,
                     "achieved_GBs": bytes_per_mesh * value / world / 1e9,
                     "frac_of_peak": bytes_per_mesh * value / world / 1e9 / HBM_PEAK_GBS},
    }
    if rank == 0 and not args.no_kernel_timing:
        live = {k: float(np.mean(v)) for k, v in hp.kernel_times_ms().items() if v}      # FPS level 1, inside the timed region
        hp.enable_kernel_timing(8, stride=2)                                            # every kernel class, after it
        for _ in range(8):
            hp.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        avg = {k: float(np.mean(v)) for k, v in hp.kernel_times_ms().items() if v}
        out["kernel_timing"] = {"fps_l1_in_timed_region_ms": live.get("fps_l1"), "fps_l1_after_it_ms": avg.get("fps_l1"),
                                "note": "FPS level 1 (the dominant kernel) is timed with HIP events on its stream inside the timed region; "
                                        "kernel_ms_per_step comes from 8 more steps behind it with events around every launch"}
        avg.update(live)
        dom = max(avg, key=avg.get)
        lvl = int(dom.split("_l")[1]) - 1
        kind = dom.split("_l")[0]
        algo = per_level[lvl][kind] * B
        achieved = algo / (avg[dom] * 1e-3) / 1e9
        hbm_view = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": algo}
        if kind == "fps":
            # the dominant kernel is a LATENCY-bound serial chain (S-1 dependent block-wide arg-maxes, the cloud in VGPRs): its yardstick
            # is the time per iteration; the HBM figures the metric asks for sit beside it (`hbm_view`) and are tiny by construction
            S_ = shape["npoint"][lvl]
            us_iter = 1e3 * avg[dom] / max(S_ - 1, 1)
            floor = fps_latency_floor()
            out["roofline"] = {"kernel": dom, "bound": "latency", "achieved": us_iter, "peak": floor.get("chain_us"),
                               "unit": "us per FPS iteration (lower is better; frac = floor / achieved)",
                               "frac": (floor["chain_us"] / us_iter) if floor.get("chain_us") else None, "traffic": None,
                               "avg_launch_ms": avg[dom], "hbm_view": hbm_view, "floor": floor,
                               "frac_vs_chain_plus_one_bucket": (floor["chain_plus_one_bucket_us"] / us_iter)
                               if floor.get("chain_plus_one_bucket_us") else None,
                               "primitive_floor_us": floor.get("primitive_us"),
                               "frac_vs_primitive_floor": (floor["primitive_us"] / us_iter) if floor.get("primitive_us") else None,
                               "note": "FPS is bound by the serial chain of S-1 block-wide argmaxes (instruction-issue latency of lone waves), "
                                       "not by HBM or the matrix cores.  Three yardsticks, all measured on this GPU by tools/fps_floor.hip "
                                       "(8 waves, one workgroup per CU): primitive_floor_us = what no register-resident FPS can do without "
                                       "(a lane value -> 6-step DPP max -> ballot -> ONE 8-byte record per wave -> s_barrier -> 8 records -> "
                                       "3-step DPP -> key -> the winner's coordinates by one LDS read); peak = the production kernel's own "
                                       "chain with the data work removed (box test in front, coordinates in a second record, four readlanes, "
                                       "the result row); chain_plus_one_bucket = that plus the one bucket refresh the algorithm cannot skip.  "
                                       "achieved = this launch's time / (S-1), set-up included.  `hbm_view` prices the same launch against HBM "
                                       "as the metric demands (`traffic` = its counter bytes); `roofline_group` is the HBM-bound kernel of the path"}
        else:
            out["roofline"] = {"kernel": dom, "bound": "hbm", **hbm_view, "avg_launch_ms": avg[dom]}
        out["kernel_ms_per_step"] = {k: round(v, 4) for k, v in sorted(avg.items())}
        if kind == "fps":
            S = shape["npoint"][lvl]
            out["roofline"]["us_per_fps_iteration"] = 1e3 * avg[dom] / max(S - 1, 1)
            # the vector-ALU view of the same launch: what the reference's brute-force update -- 3 sub, 3 mul, 2 add, 1 min per
            # point and sample (sampling_cuda_kernel.cu:50-56) -- amounts to, against the fp32 vector peak
            n_in = shape["n"] if lvl == 0 else shape["npoint"][lvl - 1]
            flops = 9.0 * n_in * max(S - 1, 0) * B
            out["roofline"]["valu"] = {"algorithmic_flops_per_launch": flops, "achieved_TFLOPs": flops / (avg[dom] * 1e-3) / 1e12,
                                       "peak_TFLOPs_fp32_vector": 157.3, "frac": flops / (avg[dom] * 1e-3) / 1e12 / 157.3,
                                       "note": "9 flop per point and sample is the brute-force update; the bucket kernel skips most of "
                                               "it (~55 VALU instructions per wave and iteration instead of ~420, "
                                               "profiles/r02_pmc_sq_counters.txt) and is bound by the dependency chain of one "
                                               "iteration in the wave that holds the new sample, so this is not a utilisation figure"}
        # HBM bytes per launch: NOT measured in this run -- read from the PMC passes committed under profiles/ (tools/gpu_pmc.sh,
        # separate rocprofv3 --pmc runs of the same workload; labelled `traffic_source`)
        pmc = {}
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(REPO, "profiles", name)))
                pmc_name = name
                break
            except Exception:
                pass
        pmc_ok = B == 256 and args.shape == "A" and not args.fused
        if dom in pmc and pmc_ok:
            tgt = out["roofline"].get("hbm_view", out["roofline"])
            tgt["traffic"] = pmc[dom]["fetch"] + pmc[dom]["write"]
            tgt["traffic_source"] = f"profiles/{pmc_name} (committed PMC passes of the same workload, not this run)"
            out["roofline"]["traffic"] = tgt["traffic"]         # HBM bytes per launch of the dominant kernel (FETCH_SIZE + WRITE_SIZE)
            out["roofline"]["traffic_source"] = tgt["traffic_source"]
        # the HBM-bound kernel of the path, for reference next to the (latency-bound) dominant one
        gk = max((k for k in avg if k.startswith("group")), key=lambda k: avg[k])
        gl = int(gk.split("_l")[1]) - 1
        galgo = per_level[gl]["group"] * B
        out["roofline_group"] = {"kernel": gk, "bound": "hbm", "achieved": galgo / (avg[gk] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": galgo / (avg[gk] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": (pmc[gk]["fetch"] + pmc[gk]["write"]) if gk in pmc and pmc_ok else None,
                                 "algorithmic_bytes_per_launch": galgo, "avg_launch_ms": avg[gk]}
    if args.fused and rank == 0 and not args.no_kernel_timing:
        # the fused levels are bound by the fp32 matrix cores: flops of the first layers (per POINT where the transform commutes
        # with the gather, per gathered row in the direct form) and of the second layers (per gathered row), against 157.3 TFLOP/s
        fl = _secondary().fused_flops(shape)
        sa_ms = sum(v for k, v in avg.items() if k.startswith("group"))
        tf = fl * B / (sa_ms * 1e-3) / 1e12
        peak, what = _secondary()._sa_peak()     # bf16 dense peak / 6 in the default bf16x3 form, the fp32 MFMA peak in the exact form
        out["roofline"] = {"kernel": "set-abstraction kernels (sa_mlp2_max + sa_point_transform)", "bound": "mfma", "achieved": tf,
                           "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                           "algorithmic_flops_per_launch": fl * B, "avg_launch_ms": sa_ms, "fp32_mfma_peak": 157.3,
                           "note": what + "; timed under the group_l* keys"}
        out.pop("roofline_group", None)
    if world == 1 and not args.fps_prefix and not args.no_alt and not args.fused:
        # the same steps with levels 2 and 3 answered by the FPS-of-an-FPS-result identity (exact):
        # reported next to the headline, never as the headline
        del hp
        torch.cuda.empty_cache()
        hp2 = hotpath.HotPath(B, device, shape=shape, pipeline=bool(args.pipeline), fps_prefix=True, **gopts)
        for _ in range(max(args.warmup, 1)):
            hp2.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hp2.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["with_fps_prefix_identity"] = {"value": B * args.steps / dt, "unit": "meshes/s", "ms_per_step": 1e3 * dt / args.steps,
                                           "note": "levels 2-3 sample the previous level's FPS result: provably 0..S-1, "
                                                   "certificate checked per cloud on the device; same outputs"}
        del hp2
    if rank == 0 and world == 1 and args.cpu_meshes != 0:
        from oracle import cpu as O
        budget = args.cpu_meshes if args.cpu_meshes > 0 else max(8, 2 * O.num_threads())
        c_port = cpu_baseline_c_port(scans if scans.shape[0] >= budget else
                                     np.concatenate([scans] * (budget // scans.shape[0] + 1))[:budget], budget, shape)
        if args.shape == "A" and not args.fused:
            out["cpu_baseline"] = cpu_baseline(scans, shape)
            out["cpu_baseline"]["c_openmp_port"] = c_port
            out["speedup_vs_c_openmp_port"] = value / c_port["value"]
            try:   # the reference's OWN functions, timed where its checkout exists (tools/ref_cpu_baseline.py, build container)
                ref = json.load(open(os.path.join(REPO, "profiles", "r04_reference_cpu.json")))
                out["cpu_baseline"]["reference_in_build_container"] = {
                    "kind": "reference", "unit": "meshes/s", "source": "profiles/r04_reference_cpu.json (tools/ref_cpu_baseline.py; NOT this host)",
                    "host": ref["host"], **{k: v["meshes_per_s"] for k, v in ref["results"].items()}, "what": ref["what"]}
            except Exception:
                pass
        else:
            out["cpu_baseline"] = c_port
        # which baseline the ratio is over is part of its name: rounds 1-4 divided by the C/OpenMP port (still reported as
        # speedup_vs_c_openmp_port), rounds 5-6 by the reference's torch-CPU path restated call for call
        out["cpu_baseline"]["baseline_of_speedup"] = ("reference torch-CPU path, restated (oracle/torch_cpu.py)"
                                                      if (args.shape == "A" and not args.fused) else "C/OpenMP port (oracle/pointops_oracle.c)")
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if args.shape == "A" and not args.fused:
            out["speedup_vs_reference_torch_cpu"] = out["speedup_vs_cpu_baseline"]
    if rank == 0 and world == 1 and args.secondary and args.shape == "A" and not args.fused and not args.fps_prefix:
        # in a process of its own, under a time limit: a fault or a hang in a non-headline configuration (graph-captured training
        # step, the matrix-core kernels) must not take the headline line with it
        hp = xyz = feats = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out["secondary"] = run_secondary(args.secondary_timeout)
    if rank == 0:
        launch.emit(out)
    launch.shutdown()


if __name__ == "__main__":
    launch.guard(main)
