#!/usr/bin/env python3
"""Time the FPS kernel shapes (threads x points-per-lane) on the GPU box: us per iteration and ms per launch.

    python tools/fps_sweep.py [--batch 256]
Forces each instantiated shape with _lib.set_tuning("fps_config", (nt, p)) and times it with HIP events on the launch stream."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import _lib, synth  # noqa: E402

CONFIGS = [(64, 1), (64, 2), (64, 4), (64, 8), (64, 16), (256, 8), (512, 8), (256, 16),
           (512, 16), (512, 24), (512, 32), (1024, 24), (512, 48), (512, 56)]


def time_fps(B, N, S, cfg, flags=0, reps=3):
    dev = torch.device("cuda")
    xyz = torch.from_numpy(np.stack([synth.arch_cloud(N, s, False) for s in range(min(B, 4))])).to(dev)
    xyz = xyz.repeat((B + 3) // 4, 1, 1)[:B].contiguous()
    idx = torch.empty(B, S, dtype=torch.int32, device=dev)
    _lib.set_tuning("fps_config", cfg if cfg else 0)
    L = _lib.lib()
    ts = []
    for r in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.tgn_furthestsampling_dense(B, N, S, _lib.ptr(xyz), None, _lib.ptr(idx), None,
                                                _lib.FPS_LOCAL_INDEX | flags, _lib.stream()))
        b.record()
        torch.cuda.synchronize()
        if r:
            ts.append(a.elapsed_time(b))
    return min(ts), idx


def bucket_sweep(batch):
    """bucket-skipping kernel (fps_bucket.hip) per shape vs the plain kernel."""
    shapes = [(256, 8), (256, 16), (512, 16), (512, 24), (512, 32), (512, 48), (512, 56)]
    for (N, S) in [(24000, 4096), (4096, 1024), (6000, 1500), (3072, 768)]:
        with _lib.tuning(fps_plain=1):
            ms1, ref = time_fps(batch, N, S, None)
        print(f"N={N:6d} S={S:5d} B={batch:4d} plain            {ms1:9.3f} ms {1e3 * ms1 / (S - 1):7.3f} us/iter", flush=True)
        for nt, p in shapes:
            if nt * p < N or nt * p > 4 * N:
                continue
            with _lib.tuning(fps_bucket_config=(nt, p), fps_bucket_min=0):
                for B in sorted({1, batch}):
                    ms, idx = time_fps(B, N, S, None)
                    print(f"N={N:6d} S={S:5d} B={B:4d} bucket {nt:4d}x{p:<2d}   {ms:9.3f} ms {1e3 * ms / (S - 1):7.3f} us/iter  "
                          f"same_idx={bool(torch.equal(idx[0], ref[0]))}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--bucket", action="store_true")
    args = ap.parse_args()
    if args.bucket:
        return bucket_sweep(args.batch)
    _lib.set_tuning("fps_plain", 1)
    for (N, S) in [(4096, 1024), (1024, 256), (6000, 1500), (3072, 768), (1500, 375)]:
        ref = None
        for cfg in CONFIGS:
            if cfg[0] * cfg[1] < N or cfg[0] * cfg[1] > 2 * max(N, 64):
                continue
            for B in sorted({1, args.batch}):
                ms, idx = time_fps(B, N, S, cfg)
                if ref is None:
                    ref = idx[0].clone()
                ok = bool(torch.equal(idx[0], ref))
                print(f"N={N:6d} S={S:5d} B={B:4d} cfg={cfg[0]:4d}x{cfg[1]:<2d} {ms:9.3f} ms  "
                      f"{1e3 * ms / (S - 1):7.3f} us/iter  {B / ms * 1e3:10.0f} clouds/s  same_idx={ok}", flush=True)


if __name__ == "__main__":
    main()
