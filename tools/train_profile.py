#!/usr/bin/env python3
"""Top GPU kernels of one steady-state training step of the tgnet_fps first-stage network (torch.profiler, kernels only)."""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
spec = importlib.util.spec_from_file_location("tsb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_bench.py"))
tsb = importlib.util.module_from_spec(spec); spec.loader.exec_module(tsb)
dev = torch.device("cuda")
feat, xyz, label = tsb.make_scan(24000, 3, dev)
torch.manual_seed(0)
net = tsb.FirstStage().to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
def step():
    offset, sem = net(feat)
    loss, _ = tsb.losses(offset, sem, xyz, label)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=80))
