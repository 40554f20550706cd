#!/usr/bin/env python3
"""Top GPU kernels of the PointNet++ MSG forward (the reference network's shape, 8 scans) -- torch.profiler table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from toothgroupnetwork_amd import nets, synth
dev = torch.device("cuda")
net = nets.PointNetPPSeg().to(dev).eval()
B = int(os.environ.get("B", "8"))
pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 3).transpose(0, 2, 1).copy()).to(dev)
with torch.no_grad():
    for _ in range(3):
        net([pts])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net([pts])
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
