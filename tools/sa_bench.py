#!/usr/bin/env python3
"""Eval-mode PointNetSetAbstractionMsg forward at the reference's sa1/sa2/sa3 shapes (pointnet_pp.py:13-15, scale=4):
the chained two-layer kernel (tgn_sa_mlp2_max: nothing of size S*K written, no torch convolution) against the round-2
form (fused first layer, (B,S,K,C1) written, layer 2 + max in torch) and the materialised path (group + torch);
then the chained kernel alone per branch with its fp32-MFMA rate.  Batch of 8 scans (and 64 for the kernel rates)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import pointnet2_utils as U, synth
dev = torch.device("cuda")
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
def build(B):
    pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 1).transpose(0, 2, 1).copy()).to(dev)
    sa1 = U.PointNetSetAbstractionMsg(1024, [0.025, 0.05], [32, 64], 6, [[128, 128], [128, 128]]).to(dev).eval()
    sa2 = U.PointNetSetAbstractionMsg(512, [0.05, 0.1], [32, 64], 256, [[256, 512], [256, 512]]).to(dev).eval()
    sa3 = U.PointNetSetAbstractionMsg(256, [0.1, 0.2], [32, 64], 1024, [[784, 1024], [784, 1024]]).to(dev).eval()
    with torch.no_grad():
        x1, f1 = sa1(pts[:, :3].contiguous(), pts); x2, f2 = sa2(x1, f1)
    return (("sa1", sa1, (pts[:, :3].contiguous(), pts)), ("sa2", sa2, (x1, f1)), ("sa3", sa3, (x2, f2)))
keep = U._mlp2_shape_ok
with torch.no_grad():
    B = 8
    for name, m, a in build(B):
        U.FUSED_SA = True; U._mlp2_shape_ok = keep; tc = timeit(lambda: m(*a))
        U._mlp2_shape_ok = lambda K, C1: False; t1 = timeit(lambda: m(*a))
        U._mlp2_shape_ok = keep; U.FUSED_SA = False; tp = timeit(lambda: m(*a))
        U.FUSED_SA = True
        print(f"{name}: chained {tc:.3f} ms | first layer fused + torch tail {t1:.3f} ms ({t1 / tc:.2f}x) | materialised {tp:.3f} ms "
              f"({tp / tc:.2f}x)  batch {B}", flush=True)
    # the chained kernel alone (sampling and ball query outside the timed region)
    B = 64
    for name, m, a in build(B):
        xyz = a[0].permute(0, 2, 1).contiguous(); pts = a[1].permute(0, 2, 1).contiguous()
        _, new_xyz = U._fps_dense(xyz, m.npoint, want_coords=True)
        for i, (r, K) in enumerate(zip(m.radius_list, m.nsample_list)):
            idx = U.query_ball_point(r, K, xyz, new_xyz).to(torch.int32)
            convs, bns = m.conv_blocks[i], m.bn_blocks[i]
            ms = timeit(lambda: U.sa_level_mlp2_max(xyz, new_xyz, pts, idx, convs, bns, False), reps=8)
            C1, C2 = convs[0].out_channels, convs[1].out_channels
            fl2 = 2.0 * B * m.npoint * K * C1 * C2
            D = pts.shape[2]
            direct = 3 + D <= 16
            fl1 = 2.0 * B * (m.npoint * K if direct else xyz.shape[1]) * (3 + D) * C1
            print(f"  {name} K={K}: level (first-layer {'direct' if direct else 'transform + commuted'} + chained kernel) {ms:.3f} ms "
                  f"for {B} scans = {(fl1 + fl2) / ms / 1e9:.1f} TFLOP/s fp32 (layer 2 alone {fl2 / 1e9 / B:.2f} GFLOP per scan)", flush=True)
