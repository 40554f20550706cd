#!/usr/bin/env python3
"""Time tgn_group_points_ex at the three Shape-A levels (256 scans, real ball-query neighbourhoods of synthetic arch
scans) for every kernel variant: implementation, store policy, grid bound.  Prints achieved GB/s of algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from toothgroupnetwork_amd import _lib, hotpath

dev = torch.device("cuda")
L = _lib.lib()
B = int(os.environ.get("B", "256"))
# inputs: one real chain (FPS + ball query) so that the neighbourhoods have the spatial structure of the benchmark
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

xyz, feats, _ = bench.make_inputs(B, dev, 100, hotpath.SHAPE_A)
hp = hotpath.HotPath(B, dev)
levels = hp.run(xyz, feats)
torch.cuda.synchronize()
_, per_level = hotpath.algorithmic_bytes(**hotpath.SHAPE_A)
variants = ([(1, -1, 0)] + [(2, p, mb) for p in (2, 16) for mb in (0, 1024, 512)]
            + [(7, p, mb) for p in (0, 2, 16) for mb in (0, 1024, 256)])
if len(sys.argv) > 1 and sys.argv[1] == "pairs":   # level 1: the pairs kernel (impl 10) against the per-element / staged kernels
    variants = [(1, -1, 0), (2, 16, 0)] + [(10, p, mb) for p in (16, 0, 2) for mb in (0, 2048, 1024, 512, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "rows":    # the row-piece kernel only: nt / write-through stores, whole chip / the grid it gets beside FPS
    variants = [(7, p, mb) for p in (2, 16) for mb in (0, 256)]
print(f"{'impl':>4} {'pol':>3} {'maxb':>5} | " + " | ".join(f"L{i + 1} ms   GB/s" for i in range(3)))
for impl, pol, mb in variants:
    row = []
    cur = xyz
    for i, lv in enumerate(levels):
        pts = feats[i]
        out = lv["grouped"]

        def run():
            _lib.check(L.tgn_group_points_ex(B, lv["N"], lv["S"], lv["K"], lv["D"], _lib.ptr(cur), _lib.ptr(lv["new_xyz"]),
                                             _lib.ptr(pts), _lib.ptr(lv["group_idx"]), 0, 1, _lib.ptr(out), impl, pol, mb,
                                             _lib.stream()))
        for _ in range(2):
            run()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ms = min(ts)
        row.append(f"{ms:6.3f} {per_level[i]['group'] * B / ms / 1e6:6.0f}")
        cur = lv["new_xyz"]
    print(f"{impl:>4} {pol:>3} {mb:>5} | " + " | ".join(row), flush=True)
