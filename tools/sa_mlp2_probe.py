#!/usr/bin/env python3
"""One chained set-abstraction level (the reference net's sa3, K = 64: 256 queries x 64 neighbours, 1027 -> 784 -> 1024) on 64 scans,
launched a few times: the target of the SQ counter passes in tools/gpu_pmc_sa.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import pointnet2_utils as U, synth
dev = torch.device("cuda")
B = int(os.environ.get("B", "64"))
torch.manual_seed(0)
xyz = torch.from_numpy(synth.scan_batch(B, 512, "arch", 1)[:, :, :3].copy()).to(dev)
pts = torch.randn(B, 512, 1024, device=dev)
m = U.PointNetSetAbstractionMsg(256, [0.1, 0.2], [32, 64], 1024, [[784, 1024], [784, 1024]]).to(dev).eval()
with torch.no_grad():
    _, new_xyz = U._fps_dense(xyz, 256, want_coords=True)
    idx = U.query_ball_point(0.2, 64, xyz, new_xyz).to(torch.int32)
    for _ in range(int(os.environ.get("REPS", "4"))):
        U.sa_level_mlp2_max(xyz, new_xyz, pts, idx, m.conv_blocks[1], m.bn_blocks[1], False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); U.sa_level_mlp2_max(xyz, new_xyz, pts, idx, m.conv_blocks[1], m.bn_blocks[1], False); b.record(); torch.cuda.synchronize()
    print(f"sa3 K=64 level, {B} scans: {a.elapsed_time(b):.3f} ms")
