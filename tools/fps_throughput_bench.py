#!/usr/bin/env python3
"""FPS level 1 of the headline (24 000 -> 4096 per scan) as a THROUGHPUT problem: the register-resident bucket kernel (one workgroup per
CU) against the owner-wave kernel out of an L2-resident workspace (TGN_FPS_THROUGHPUT: four workgroups per CU), for batches of 256 ...
1024 distinct scans, alone and beside a stream of device-to-device copies (what the groupings do to the memory system).

    python tools/fps_throughput_bench.py [--batches 256,512,768,1024] [--reps 5]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402
from toothgroupnetwork_amd import _lib, hotpath  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="256,512,768,1024")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--n", type=int, default=24000)
    ap.add_argument("--s", type=int, default=4096)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    shape = dict(hotpath.SHAPE_A, n=args.n)
    out = []
    big = max(int(b) for b in args.batches.split(","))
    xyz_all, _, _ = bench.make_inputs(big, dev, seed=100, shape=shape)
    side = torch.cuda.Stream(device=dev)
    src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    for B in [int(b) for b in args.batches.split(",")]:
        xyz = xyz_all[:B].contiguous()
        idx = {k: torch.empty(B, args.s, dtype=torch.int32, device=dev) for k in ("reg", "l2")}
        nx = {k: torch.empty(B, args.s, 3, device=dev) for k in ("reg", "l2")}
        nbytes = int(L.tgn_fps_throughput_workspace_bytes(B, args.n))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

        def run(kind):
            flags = _lib.FPS_LOCAL_INDEX | (_lib.FPS_THROUGHPUT if kind == "l2" else 0)
            _lib.check(L.tgn_furthestsampling_dense_ws(B, args.n, args.s, _lib.ptr(xyz), _lib.ptr(ws), nbytes, _lib.ptr(idx[kind]),
                                                       _lib.ptr(nx[kind]), flags, _lib.stream()), "fps")

        def timed(kind, copies):
            ts = []
            for _ in range(args.reps):
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if copies:
                    with torch.cuda.stream(side):
                        for _ in range(copies):
                            dst.copy_(src, non_blocking=True)
                a.record()
                run(kind)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            return sorted(ts)[len(ts) // 2]

        run("reg"), run("l2")
        torch.cuda.synchronize()
        same = bool(torch.equal(idx["reg"], idx["l2"]) and torch.equal(nx["reg"], nx["l2"]))
        row = dict(B=B, same=same)
        for kind in ("reg", "l2"):
            ms = timed(kind, 0)
            ms_c = timed(kind, 40 * B // 256)       # ~0.5 GB of traffic per ms of FPS
            row[kind] = dict(ms=round(ms, 3), us_per_scan=round(1e3 * ms / B, 2), ms_beside_copies=round(ms_c, 3),
                             us_per_scan_beside_copies=round(1e3 * ms_c / B, 2))
        print(json.dumps(row), flush=True)
        out.append(row)


if __name__ == "__main__":
    main()
