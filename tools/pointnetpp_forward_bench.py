#!/usr/bin/env python3
"""BASELINE config 2 end to end: the PointNet++-MSG segmentation network the reference instantiates
(models/modules/pointnet_pp.py:13-20: sa1/sa2/sa3 + fp3/fp2/fp1, scale=4), assembled from THIS repo's drop-in
modules, forward only, 24 000-point scans, eval mode.  Reports ms per forward for batch 1 and 8, fused first layer on/off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from toothgroupnetwork_amd import pointnet2_utils as U, synth

class Net(nn.Module):
    def __init__(self, scale=4, cin=6):
        super().__init__()
        s = scale
        self.sa1 = U.PointNetSetAbstractionMsg(1024, [0.025, 0.05], [32, 64], cin, [[32 * s, 32 * s], [32 * s, 32 * s]])
        self.sa2 = U.PointNetSetAbstractionMsg(512, [0.05, 0.1], [32, 64], 64 * s, [[64 * s, 128 * s], [64 * s, 128 * s]])
        self.sa3 = U.PointNetSetAbstractionMsg(256, [0.1, 0.2], [32, 64], 256 * s, [[196 * s, 256 * s], [196 * s, 256 * s]])
        self.fp3 = U.PointNetFeaturePropagation((512 + 256) * s, [256 * s, 256 * s])
        self.fp2 = U.PointNetFeaturePropagation((256 + 64) * s, [128 * s, 128 * s])
        self.fp1 = U.PointNetFeaturePropagation(128 * s + cin, [64 * s, 32 * s])
        self.head = nn.Sequential(nn.Conv1d(32 * s, 17, 1), nn.BatchNorm1d(17), nn.ReLU(), nn.Conv1d(17, 17, 1))
    def forward(self, pts):
        xyz = pts[:, :3, :].contiguous()
        x1, f1 = self.sa1(xyz, pts); x2, f2 = self.sa2(x1, f1); x3, f3 = self.sa3(x2, f2)
        f2 = self.fp3(x2, x3, f2, f3); f1 = self.fp2(x1, x2, f1, f2); f0 = self.fp1(xyz, x1, pts, f1)
        return self.head(f0)

dev = torch.device("cuda"); net = Net().to(dev).eval()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
with torch.no_grad():
    for B in (1, 8):
        pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 3).transpose(0, 2, 1).copy()).to(dev)
        for fused in (True, False):
            U.FUSED_SA = fused
            ms = timeit(lambda: net(pts))
            print(f"pointnet++ MSG forward, 24000 pts, batch {B}, fused_first_layer={fused}: {ms:.2f} ms ({B / ms * 1e3:.1f} scans/s)", flush=True)

    # the same forward captured in a HIP graph (every launch of this package goes to the current stream through the
    # C ABI, nothing synchronises with the host in the dense path): launch overhead and Python time disappear
    U.FUSED_SA = True
    for B in (1, 8):
        pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 3).transpose(0, 2, 1).copy()).to(dev)
        static_in = pts.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                ref = net(static_in)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = net(static_in)
            g.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(static_out, ref))
            ms = timeit(g.replay, reps=10)
            print(f"pointnet++ MSG forward, 24000 pts, batch {B}, HIP graph replay: {ms:.2f} ms ({B / ms * 1e3:.1f} scans/s), "
                  f"output identical to eager: {same}", flush=True)
        except Exception as e:  # noqa: BLE001
            print("graph capture failed:", type(e).__name__, str(e)[:300], flush=True)
