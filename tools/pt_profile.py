#!/usr/bin/env python3
"""Top GPU kernels of the Point-Transformer U-Net eval forward on one 24 000-point scan (torch.profiler table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from toothgroupnetwork_amd import point_transformer as PT, synth
dev = torch.device("cuda")
torch.manual_seed(0)
B = int(os.environ.get("B", "1"))
net = PT.PointTransformerUNet().to(dev).eval()
inp = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 3).transpose(0, 2, 1).copy()).to(dev)
with torch.no_grad():
    for _ in range(3):
        net(inp)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        net(inp)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=26, max_name_column_width=70))
