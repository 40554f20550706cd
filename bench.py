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

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (FPS level 1), timed with HIP
events on the launch stream inside the timed region; `cpu_baseline` is the C oracle (a scalar port of
the reference algorithm, OpenMP over clouds/queries) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import hotpath, sharding, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable


def make_inputs(B, device, seed, shape):
    """B synthetic 24 000-point scans (xyz + normals) and synthetic level-2/3 features, resident in HBM."""
    n_unique = min(B, 16)  # 16 distinct arch scans, tiled: generation cost stays bounded
    scans = synth.scan_batch(n_unique, shape["n"], "arch", seed=seed)
    scans = np.concatenate([scans] * ((B + n_unique - 1) // n_unique), axis=0)[:B]
    pts = torch.from_numpy(scans).to(device)
    xyz = pts[:, :, :3].contiguous()
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = [pts]
    for S, D in zip(shape["npoint"][:-1], shape["d"][1:]):
        f = torch.randn(n_unique, S, D, generator=g).repeat((B + n_unique - 1) // n_unique, 1, 1)[:B]
        feats.append(f.to(device).contiguous())
    return xyz, feats, scans


def cpu_baseline(scans, budget_meshes, shape):
    """The oracle (C port of the reference algorithm) on a bounded sample, all host cores (OpenMP)."""
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
            "note": "C/OpenMP port of the reference algorithm (oracle/), stronger than the reference's own torch-CPU path "
                    "(BASELINE.md section 2: ~5.8 s per mesh for level 1 alone on 8 threads); the reference checkout is not "
                    "present on the bench host, so its code cannot be timed here"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="scans per step per GPU (one FPS workgroup per scan)")
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
    ilist = lambda s: [int(v) for v in str(s).split(",")]
    ap.add_argument("--group-impl", type=ilist, default=[0], help="grouping kernel, one value or one per level: 0 choose, 1 = 4-B stores, "
                    "2 = 16-B stores through LDS, 3 = LDS-DMA ring, 7 = row pieces into an LDS image")
    ap.add_argument("--group-policy", type=ilist, default=[-1], help="cache policy of the 16-B grouping stores (0 plain, 2 nt, 16 sc1; -1 default)")
    ap.add_argument("--group-max-blocks", type=ilist, default=[-1], help="grid bound of the grouping kernel (-1: 256 = one wave per SIMD when pipelined, else none)")
    ap.add_argument("--fused", type=int, default=0, help="1: every level is a fused set-abstraction level (single-layer shared MLP "
                    "[128,512,1024], eval-mode BatchNorm folded): the grouped tensor is never written -- a second, non-headline line")
    ap.add_argument("--ball-stream", type=int, default=-1, help="-1: default (2 = phased: ball queries on a third stream beside FPS levels "
                    "2-3, fenced off from the next step's FPS level 1); 0: in line on the FPS stream; 1: third stream, free-running")
    ap.add_argument("--ball-split", type=int, default=-1, help="phased schedule: where the ball queries of levels 2-3 run (hotpath.py; -1 = default: the last level in front of the groupings)")
    ap.add_argument("--group-delay-us", type=int, default=-1, help="gated schedule: hold the groupings back by this long behind the start of "
                    "FPS level 1 (-1: default, ~100 us at 24 000 points: most of the FPS set-up)")
    ap.add_argument("--group-order", type=ilist, default=None, help="gated schedule: order of the grouping launches, e.g. 2,1,0")
    ap.add_argument("--grid-stream", type=int, default=0, help="phased schedule: the early level-1 grid on a stream of its own (experiment)")
    ap.add_argument("--low-valu", type=int, default=1, help="phased schedule: FPS level 2 on the bucket-skipping kernel (TGN_FPS_LOW_VALU)")
    ap.add_argument("--early-grid", type=int, default=-1, help="phased schedule: build the level-1 ball-query grid ahead of the fence (-1 default on)")
    ap.add_argument("--group-gate", type=int, default=-1, help="1: groupings of a step wait for its last ball query, i.e. run beside the next "
                    "step's FPS level 1 (-1: default on when pipelined)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL; gloo lets several ranks share "
                    "one GPU, e.g. to exercise the N > 1 branch on a one-GPU box)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra with_fps_prefix_identity measurement (profiling runs)")
    args = ap.parse_args()

    rank, local_rank, world, device = sharding.init_from_env(backend=args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the hot path has no CPU implementation")
    if world != max(args.gpus, 1):
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    B = args.batch
    shape = hotpath.SHAPE_A if args.shape == "A" else hotpath.SHAPE_B
    xyz, feats, scans = make_inputs(B, device, seed=100 + rank, shape=shape)
    one = lambda v: v[0] if len(v) == 1 else v
    mb = one(args.group_max_blocks)
    gopts = dict(group_impl=one(args.group_impl), group_policy=one(args.group_policy),
                 group_max_blocks=None if mb == -1 else mb, fused=bool(args.fused),
                 ball_stream=None if args.ball_stream < 0 else args.ball_stream,
                 group_gate=None if args.group_gate < 0 else bool(args.group_gate),
                 early_grid=None if args.early_grid < 0 else bool(args.early_grid), ball_split=None if args.ball_split < 0 else int(args.ball_split), grid_stream=bool(args.grid_stream),
                 low_valu=bool(args.low_valu), group_order=args.group_order, group_delay_us=None if args.group_delay_us < 0 else args.group_delay_us)
    if args.fused and args.shape != "A":
        raise SystemExit("--fused is defined for shape A (single-scale levels)")
    hp = hotpath.HotPath(B, device, shape=shape, pipeline=bool(args.pipeline), fps_prefix=bool(args.fps_prefix), **gopts)
    for _ in range(max(args.warmup, 0)):
        hp.run(xyz, feats, inputs_on_current_stream=False)   # the synthetic scans are resident before any step
    torch.cuda.synchronize()
    if not args.no_kernel_timing:
        hp.enable_kernel_timing(args.steps, stride=max(args.timing_stride, 1))

    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hp.run(xyz, feats, inputs_on_current_stream=False)
    torch.cuda.synchronize()
    sharding.barrier()
    elapsed = time.perf_counter() - t0
    per_rank_s = sharding.gather_metrics([elapsed], device=device).reshape(-1).cpu().tolist()   # the one collective of the run
    elapsed = max(per_rank_s)

    total_meshes = B * args.steps * world
    value = total_meshes / elapsed
    bytes_per_mesh, per_level = hotpath.algorithmic_bytes(fused=bool(args.fused), **shape)

    out = {
        "metric": "meshes/sec (24k-pt FPS+ball_query+group fwd)",
        "value": value,
        "unit": "meshes/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "per_rank_seconds": [round(v, 6) for v in per_rank_s],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": ("shape_A FUSED (NOT the headline configuration): 24000-pt scans, npoint=[4096,1024,256], nsample=32, "
                                "radii=[0.05,0.1,0.2]; FPS + ball query + fused set-abstraction level (gather, centre, 1x1 conv "
                                "[9->128, 131->512, 515->1024] on the fp32 matrix cores, folded BatchNorm, ReLU, max over K); "
                                "the grouped tensor is never written, level l feeds level l+1") if args.fused else
                               ("shape_A: 24000-pt scans, npoint=[4096,1024,256], nsample=32, radii=[0.05,0.1,0.2], "
                                "D=[6,128,512]; FPS+ball_query+group forward, grouped tensors materialised") if args.shape == "A" else
                               ("shape_B (NOT the headline configuration): 24000-pt scans, npoint=[1024,512,256], multi-scale "
                                "radii [[.025,.05],[.05,.1],[.1,.2]], nsample [32,64], D=[6,256,1024]; FPS+ball_query+group forward"),
                   "meshes_per_step_per_gpu": B, "sharding": f"independent meshes x {world} ranks, no data-path collective",
                   "index_dtype": "int32",
                   "fps_levels_2_3": "identity shortcut (FPS of an FPS result; certificate checked on device)"
                   if args.fps_prefix else "iterated like level 1",
                   "schedule": hp.describe_schedule()},
        "path_hbm": {"algorithmic_bytes_per_mesh": bytes_per_mesh,
                     "achieved_GBs": bytes_per_mesh * value / world / 1e9,
                     "frac_of_peak": bytes_per_mesh * value / world / 1e9 / HBM_PEAK_GBS},
    }
    if rank == 0 and not args.no_kernel_timing:
        times = hp.kernel_times_ms()
        avg = {k: float(np.mean(v)) for k, v in times.items() if v}
        dom = max(avg, key=avg.get)
        lvl = int(dom.split("_l")[1]) - 1
        kind = dom.split("_l")[0]
        algo = per_level[lvl][kind] * B
        achieved = algo / (avg[dom] * 1e-3) / 1e9
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": algo, "avg_launch_ms": avg[dom],
                           "note": "FPS is bound by the serial chain of S-1 block-wide argmaxes and fp32 VALU issue, not by HBM "
                                   "(the cloud lives in VGPRs); the HBM fraction is reported because the metric asks for it"}
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
        # HBM bytes per launch from the PMC passes committed under profiles/ (tools/gpu_pmc.sh; same workload)
        pmc = {}
        for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(REPO, "profiles", name)))
                break
            except Exception:
                pass
        pmc_ok = B == 256 and args.shape == "A" and not args.fused
        if dom in pmc and pmc_ok:
            out["roofline"]["traffic"] = pmc[dom]["fetch"] + pmc[dom]["write"]
        # the HBM-bound kernel of the path, for reference next to the (latency-bound) dominant one
        gk = max((k for k in avg if k.startswith("group")), key=lambda k: avg[k])
        gl = int(gk.split("_l")[1]) - 1
        galgo = per_level[gl]["group"] * B
        out["roofline_group"] = {"kernel": gk, "bound": "hbm", "achieved": galgo / (avg[gk] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": galgo / (avg[gk] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": (pmc[gk]["fetch"] + pmc[gk]["write"]) if gk in pmc and pmc_ok else None,
                                 "algorithmic_bytes_per_launch": galgo, "avg_launch_ms": avg[gk]}
    if args.fused and rank == 0 and not args.no_kernel_timing:
        # matrix-core side of the fused levels: flops of the per-point transforms / direct contractions per step
        fl = 0
        Nl = shape["n"]
        for S, K, D, C1 in zip(shape["npoint"], shape["nsample"], shape["d"], shape["c_out"]):
            direct = (3 + D) <= 16
            fl += 2 * (S * K if direct else Nl) * (3 + D) * C1
            Nl = S
        sa_ms = sum(v for k, v in avg.items() if k.startswith("group"))
        out["fused_levels"] = {"flops_per_mesh": fl, "sa_kernels_ms_per_step": sa_ms,
                               "achieved_TFLOPs_fp32": fl * B / (sa_ms * 1e-3) / 1e12, "peak_TFLOPs_fp32_mfma": 157.3,
                               "note": "set-abstraction kernels (per-point transform + gather-max, or the direct kernel) are timed under the "
                                       "group_l* keys; fp32 MFMA = exact fp32, 1/16 of the bf16 rate"}
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
        out["cpu_baseline"] = cpu_baseline(scans if scans.shape[0] >= budget else
                                           np.concatenate([scans] * (budget // scans.shape[0] + 1))[:budget], budget, shape)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
