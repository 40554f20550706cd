#!/usr/bin/env python3
"""Randomised stress of the round-6 kernels against the CPU oracle, beyond the seeds tests/test_gpu_random_sweep.py pins:
  fps   fps_lean_kernel in all four arithmetic / tie-order modes on ragged packed batches of 257 .. 4096-point clouds (quantised
        coordinates: exact ties inside lanes, across lanes and across waves; NaN rows; over-sampled clouds), `fps_lean` = 2;
  ball  the chunked ball-query kernel on clouds of 2 049 .. 24 000 points (clustered sheets, quantised lattices, duplicates).
    python tests/random_stress.py [--fps 200] [--ball 60] [--seed 0]
Test infrastructure (lives under tests/: it imports oracle/)."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (tests/ is one level below the repository root)
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import cpu as oracle  # noqa: E402
from toothgroupnetwork_amd import _lib, pointnet2_utils as U  # noqa: E402


def cloud(rng, n, style):
    if style == 0:
        return rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    if style == 1:
        return (rng.integers(-6, 7, size=(n, 3)) / 8.0).astype(np.float32)
    if style == 2:
        c = rng.uniform(-1, 1, size=(max(n // 40, 1), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(scale=0.03, size=(n, 3))
        p[:, 2] *= 0.05
        return p.astype(np.float32)
    return np.full((n, 3), 0.25, np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fps", type=int, default=200)
    ap.add_argument("--ball", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(777 + args.seed)
    bad = 0
    for it in range(args.fps):
        b = int(rng.integers(1, 6))
        cap = int(rng.choice([512, 1024, 2048, 4096]))
        sizes = [int(rng.integers(cap // 2 + 1, cap + 1))] + [int(rng.integers(1, cap + 1)) for _ in range(b - 1)]
        ms = [int(rng.integers(1, n + 40)) if rng.random() < 0.15 else int(rng.integers(1, max(2, n // 3))) for n in sizes]
        cl = [cloud(rng, n, int(rng.integers(0, 4))) for n in sizes]
        for c in cl:
            if c.shape[0] > 10 and rng.random() < 0.2:
                c[rng.integers(0, c.shape[0], 3)] = np.nan
        xyz_np = np.concatenate(cl)
        off, noff = np.cumsum(sizes).astype(np.int32), np.cumsum(ms).astype(np.int32)
        mode = int(rng.integers(0, 4))
        flags = (_lib.FPS_FMA if mode & 1 else 0) | (_lib.FPS_TREE_TIES if mode & 2 else 0)
        idx = torch.full((int(noff[-1]),), -7, dtype=torch.int32, device=dev)
        nx = torch.full((int(noff[-1]), 3), -7.0, device=dev)
        xyz, off_d, noff_d = T(xyz_np), T(off), T(noff)      # (kept alive: the launch is asynchronous)
        with _lib.tuning(fps_lean=2, fps_bucket_min=1000000):
            _lib.check(L.tgn_furthestsampling(b, max(sizes), _lib.ptr(xyz), _lib.ptr(off_d), _lib.ptr(noff_d), None, _lib.ptr(idx),
                                              _lib.ptr(nx), flags, _lib.stream()))
        want = oracle.furthestsampling(xyz_np, off, noff, mode=mode)
        ok = np.array_equal(idx.cpu().numpy(), want) and np.array_equal(nx.cpu().numpy(), xyz_np[want.astype(np.int64)], equal_nan=True)
        if not ok:
            bad += 1
            print("FPS MISMATCH", it, sizes, ms, mode, flush=True)
    print(f"fps: {args.fps} ragged batches, mismatches {bad}", flush=True)
    bad_b = 0
    for it in range(args.ball):
        B = int(rng.integers(1, 4))
        N = int(rng.choice([2049, 4096, 6000, 8191, 8193, 16384, 24000]))
        S = int(rng.choice([17, 256, 1024, 4096]))
        K = int(rng.choice([1, 16, 32, 64]))
        style = int(rng.choice([0, 1, 2]))
        xyz = np.stack([cloud(rng, N, style) for _ in range(B)])
        q = np.stack([np.concatenate([xyz[b_][rng.integers(0, N, S // 2 + 1)], cloud(rng, S, 0)])[:S] for b_ in range(B)])
        radius = float(rng.choice([0.02, 0.05, 0.1, 0.2, 0.4]))
        got = U.query_ball_point(radius, K, T(xyz), T(q)).cpu().numpy()
        if not np.array_equal(got, oracle.query_ball_point(radius, K, xyz, q)):
            bad_b += 1
            print("BALL MISMATCH", it, B, N, S, K, style, radius, flush=True)
    print(f"ball: {args.ball} cases, mismatches {bad_b}", flush=True)
    sys.exit(1 if (bad or bad_b) else 0)


if __name__ == "__main__":
    main()
