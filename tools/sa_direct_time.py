#!/usr/bin/env python3
"""The direct-form chained set-abstraction kernel (level 1: 3 + D <= 16 input channels) alone, at the two shapes that use it:
Shape A level 1 (256 scans, S = 4096, K = 32, 9 -> 64 -> 128) and the reference net's sa1 (8 scans, S = 1024, K = 32 / 64, 9 -> 128 -> 128).
HIP events around back-to-back launches of tgn_sa_mlp2_max_bf16x3; TFLOP/s fp32-equivalent."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import _lib, pointnet2_utils as U, synth  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
for B, S, K, D, C1, C2, r in ((256, 4096, 32, 6, 64, 128, 0.05), (8, 1024, 32, 6, 128, 128, 0.025), (8, 1024, 64, 6, 128, 128, 0.05)):
    N = 24000
    scans = synth.scan_batch(min(B, 8), N, "arch", 3)
    scans = np.concatenate([scans] * (B // min(B, 8)))
    pts = torch.from_numpy(scans).to(dev)
    xyz = pts[:, :, :3].contiguous()
    fidx = U.farthest_point_sample(xyz, S)
    new_xyz = U.index_points(xyz, fidx).contiguous()
    gidx = U.query_ball_point(r, K, xyz, new_xyz).int().contiguous()
    g = torch.Generator().manual_seed(0)
    C1p = (C1 + 15) // 16 * 16
    Wd = torch.zeros(16, C1p)
    Wd[:9, :C1] = torch.randn(9, C1, generator=g) / 3.0
    b1 = torch.zeros(C1p)
    W2 = torch.zeros(C2, C1p)
    W2[:, :C1] = torch.randn(C2, C1, generator=g) / C1 ** 0.5
    W2f = W2.view(C2, C1p // 8, 8).permute(1, 0, 2).contiguous().to(dev)
    W2s = U.split_second_layer(W2f)
    b2 = torch.zeros(C2, device=dev)
    Wd, b1 = Wd.to(dev), b1.to(dev)
    out = torch.empty(B, S, C2, device=dev)

    def run():
        _lib.check(L.tgn_sa_mlp2_max_bf16x3(B, N, S, K, D, C1p, C2, None, _lib.ptr(xyz), _lib.ptr(pts), _lib.ptr(new_xyz), _lib.ptr(Wd), _lib.ptr(b1),
                                            _lib.ptr(gidx), 0, _lib.ptr(W2s), _lib.ptr(b2), _lib.ptr(out), C2, _lib.stream()), "sa")
    def run_fp32():     # the same level with the second layer on v_mfma_f32_32x32x2_f32 (no three-way bf16 split of the activations)
        _lib.check(L.tgn_sa_mlp2_max(B, N, S, K, D, C1p, C2, None, _lib.ptr(xyz), _lib.ptr(pts), _lib.ptr(new_xyz), _lib.ptr(Wd), _lib.ptr(b1),
                                     _lib.ptr(gidx), 0, _lib.ptr(W2f), _lib.ptr(b2), _lib.ptr(out), C2, _lib.stream()), "sa fp32")

    fl = 2.0 * B * S * K * (9 * C1 + C1 * C2)
    for name, fn in (("bf16x3", run), ("fp32 MFMA", run_fp32)):
        if os.environ.get("TGN_SA_TIME_ONLY", name) != name:
            continue
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 4)
        ms = sorted(ts)[3]
        print(f"B={B:4d} S={S} K={K} 9->{C1}->{C2} {name:10s}: {ms * 1e3:9.1f} us   {fl / ms / 1e9:7.1f} TFLOP/s fp32-equivalent   checksum {float(out.double().sum()):.6e}")
