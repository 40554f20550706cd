#!/usr/bin/env python3
"""Time tgn_group_points at the three Shape-A levels (256 scans) and report achieved GB/s of the output store."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import _lib
dev = torch.device("cuda"); L = _lib.lib(); B = 256
for (N, S, K, D) in [(24000, 4096, 32, 6), (4096, 1024, 32, 128), (1024, 256, 32, 512)]:
    xyz = torch.rand(B, N, 3, device=dev); new_xyz = xyz[:, :S].contiguous(); pts = torch.randn(B, N, D, device=dev)
    # neighbours of query s: a window of nearby indices (spatially coherent like a real ball query)
    base = torch.randint(0, N, (B, S, 1), device=dev)
    idx = ((base + torch.randint(0, 256, (B, S, K), device=dev)) % N).to(torch.int32).contiguous()
    out = torch.empty(B, S, K, 3 + D, device=dev)
    def run():
        _lib.check(L.tgn_group_points(B, N, S, K, D, _lib.ptr(xyz), _lib.ptr(new_xyz), _lib.ptr(pts), _lib.ptr(idx), 0, 1, _lib.ptr(out), _lib.stream()))
    for _ in range(2): run()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = min(ts); gb = out.numel() * 4 / 1e9
    print(f"N={N} S={S} K={K} D={D}: {ms:.3f} ms  store {gb / ms * 1e3:.0f} GB/s  (vec1={os.environ.get('TGN_GROUP_VEC1', '0')})", flush=True)
