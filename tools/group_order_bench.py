#!/usr/bin/env python3
"""Does the ORDER in which a scan's queries are processed matter for the grouping kernel?  Level-2 shape of the headline
path (N=4096 level-1 samples with 128 features, S=1024 queries, K=32) on real ball-query output: queries in FPS order
(far apart: consecutive queries share no rows) vs the same queries sorted along a Z-order curve (neighbours share rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from toothgroupnetwork_amd import _lib, pointnet2_utils as U, synth
dev = torch.device("cuda"); L = _lib.lib(); B = 256
pts = torch.from_numpy(synth.scan_batch(16, 24000, "arch", 100)).to(dev).repeat(B // 16, 1, 1)
xyz0 = pts[:, :, :3].contiguous()
_, xyz1 = U._fps_dense(xyz0, 4096, want_coords=True)                      # level-1 samples = level-2 support
for (S, r, D) in [(1024, 0.1, 128)]:
    N = xyz1.shape[1]
    _, new_xyz = U._fps_dense(xyz1, S, want_coords=True)
    idx = U.query_ball_point(r, 32, xyz1, new_xyz).to(torch.int32).contiguous()
    feats = torch.randn(B, N, D, device=dev)
    out = torch.empty(B, S, 32, 3 + D, device=dev)
    def morton(p):
        q = ((p - p.amin(1, keepdim=True)) / (p.amax(1, keepdim=True) - p.amin(1, keepdim=True) + 1e-9) * 1023).long().clamp(0, 1023)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
            return v
        return spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
    perm = morton(new_xyz).argsort(dim=1)
    nx2 = torch.gather(new_xyz, 1, perm[..., None].expand(-1, -1, 3)).contiguous()
    idx2 = torch.gather(idx, 1, perm[..., None].expand(-1, -1, 32)).contiguous()
    def run(nx, ix):
        _lib.check(L.tgn_group_points(B, N, S, 32, D, _lib.ptr(xyz1), _lib.ptr(nx), _lib.ptr(feats), _lib.ptr(ix), 0, 1, _lib.ptr(out), _lib.stream()))
    for name, nx, ix in (("FPS order", new_xyz, idx), ("Z-order", nx2, idx2)):
        for _ in range(2): run(nx, ix)
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(nx, ix); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print(f"N={N} S={S} D={D} queries in {name:10s}: {min(ts):.3f} ms  ({out.numel() * 4 / min(ts) / 1e6:.0f} GB/s stored)", flush=True)
