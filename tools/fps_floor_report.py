#!/usr/bin/env python3
"""Where FPS level 1 (24 000 -> 4096, 256 scans per launch) stands against its latency floor.

  floor      tools/_bin/fps_floor (tools/fps_floor.hip): the dependent chain of one iteration with the data work removed
  best case  the PRODUCTION kernel on a cloud where every iteration can touch only the bucket that holds the new sample:
             375 clusters of 64 points, each inside one Z-order cell, the clusters far apart compared with their size
             (after the first ~375 samples a new sample's box distance to every other cluster exceeds that cluster's maximum)
  bench      the PRODUCTION kernel on the benchmark's arch scans (8.6 of 375 buckets touched per iteration on average)
Per-iteration time = (launch of S samples - launch of 2 samples) / (S - 2): the set-up is measured and taken out."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import _lib, synth  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cluster_cloud(n=24000, seed=0):
    rng = np.random.default_rng(seed)
    ncl = n // 64
    cells = rng.permutation(np.arange(1, 16 * 16 * 16 - 1))[:ncl]
    cells[0], cells[1] = 0, 16 * 16 * 16 - 1                                  # the two corner cells hold a cluster each
    cx, cy, cz = cells % 16, (cells // 16) % 16, cells // 256
    centre = (np.stack([cx, cy, cz], 1) + 0.5) / 16.0 * 2.0 - 1.0           # one cluster per cell of the 16^3 grid over [-1, 1]^3
    pts = centre[:, None, :] + rng.uniform(-1e-3, 1e-3, size=(ncl, 64, 3))
    # one point of each corner cluster sits in the corner itself: the bounding box is [-1, 1]^3 exactly, so the kernel's 16^3
    # Z-order cells are the cells above and every bucket (64 consecutive sorted positions) is one cluster
    pts[0, 0], pts[1, 0] = (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)
    pts = pts.reshape(-1, 3)
    return pts[rng.permutation(pts.shape[0])].astype(np.float32)


def time_fps(L, xyz, S, reps=5):
    B, N = xyz.shape[:2]
    idx = torch.empty(B, S, dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty(B, S, 3, device=xyz.device)

    def run():
        _lib.check(L.tgn_furthestsampling_dense(B, N, S, _lib.ptr(xyz), None, _lib.ptr(idx), _lib.ptr(new_xyz), _lib.FPS_LOCAL_INDEX,
                                                _lib.stream()), "fps")
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), idx


def main():
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    B, N, S = 256, 24000, 4096
    r = subprocess.run([os.path.join(REPO, "tools", "_bin", "fps_floor")], capture_output=True, text=True, timeout=300)
    print(r.stdout.rstrip())
    floor = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["fps_floor"]
    rows = []
    for name, cloud in (("bench: arch scans", np.stack([synth.arch_cloud(N, 100 + i, False) for i in range(16)])),
                        ("best case: 375 far-apart clusters of 64 points", np.stack([cluster_cloud(N, i) for i in range(16)]))):
        xyz = torch.from_numpy(np.concatenate([cloud] * (B // 16))).to(dev).contiguous()
        full, idx = time_fps(L, xyz, S)
        setup, _ = time_fps(L, xyz, 2)
        us = 1e3 * (full - setup) / (S - 2)
        rows.append((name, full, setup, us))
        assert int(idx.max()) < N and len(set(idx[0].tolist())) == S
    print(f"\n# production kernel fps_bucket_kernel<512,48>, {B} scans x {N} -> {S} per launch (one workgroup per CU)")
    print(f"{'':58s} {'launch ms':>10s} {'set-up ms':>10s} {'us / iteration':>15s} {'chain / it':>11s} {'chain+1 / it':>13s}")
    for name, full, setup, us in rows:
        print(f"{name:58s} {full:10.3f} {setup:10.3f} {us:15.4f} {floor['chain_us'] / us:11.3f} {floor['chain_plus_one_bucket_us'] / us:13.3f}")
    name, full, setup, us = rows[0]
    print(f"\nroofline.frac as bench.py reports it (set-up included): floor {floor['chain_us']:.4f} us / "
          f"({full:.3f} ms / {S - 1}) = {floor['chain_us'] / (1e3 * full / (S - 1)):.3f}")
    print(json.dumps({"fps_floor": floor, "production": [dict(cloud=n_, launch_ms=f, setup_ms=s_, us_per_iteration=u) for n_, f, s_, u in rows]}))


if __name__ == "__main__":
    main()
