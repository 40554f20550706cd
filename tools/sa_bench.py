#!/usr/bin/env python3
"""Eval-mode PointNetSetAbstractionMsg forward at the reference's sa1/sa2/sa3 shapes (pointnet_pp.py:13-15, scale=4),
fused first layer vs materialised grouping; batch of 8 scans."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import pointnet2_utils as U, synth
dev = torch.device("cuda"); B = 8
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 1).transpose(0, 2, 1).copy()).to(dev)
sa1 = U.PointNetSetAbstractionMsg(1024, [0.025, 0.05], [32, 64], 6, [[128, 128], [128, 128]]).to(dev).eval()
sa2 = U.PointNetSetAbstractionMsg(512, [0.05, 0.1], [32, 64], 256, [[256, 512], [256, 512]]).to(dev).eval()
sa3 = U.PointNetSetAbstractionMsg(256, [0.1, 0.2], [32, 64], 1024, [[784, 1024], [784, 1024]]).to(dev).eval()
with torch.no_grad():
    x1, f1 = sa1(pts[:, :3].contiguous(), pts); x2, f2 = sa2(x1, f1)
    for name, m, a in (("sa1", sa1, (pts[:, :3].contiguous(), pts)), ("sa2", sa2, (x1, f1)), ("sa3", sa3, (x2, f2))):
        U.FUSED_SA = True; tf = timeit(lambda: m(*a))
        U.FUSED_SA = False; tp = timeit(lambda: m(*a))
        print(f"{name}: fused {tf:.3f} ms  materialised {tp:.3f} ms  ({tp / tf:.2f}x)  batch {B}", flush=True)
