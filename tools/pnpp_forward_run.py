#!/usr/bin/env python3
"""nets.PointNetPPSeg (pointnet_pp.py get_model) forward, 8 x 24 000-point scans, eval: N forwards for a rocprofv3 kernel
trace (tools/gpu_r5_evidence.sh -> profiles/r05_pnpp_forward_kernel_stats.csv).  Prints ms per forward (HIP events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from toothgroupnetwork_amd import nets, synth  # noqa: E402

B, reps = int(os.environ.get("B", "8")), int(os.environ.get("REPS", "10"))
dev = torch.device("cuda")
torch.manual_seed(0)
net = nets.PointNetPPSeg().to(dev).eval()
pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 5).transpose(0, 2, 1).copy()).to(dev)
with torch.no_grad():
    for _ in range(3):
        net([pts])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        net([pts])
    b.record()
    torch.cuda.synchronize()
print(f"PointNetPPSeg forward, {B} x 24000 points, eval: {a.elapsed_time(b) / reps:.3f} ms per forward ({reps} forwards after 3 warm-ups)")
