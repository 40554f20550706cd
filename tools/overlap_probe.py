#!/usr/bin/env python3
"""Can the (latency-bound, register-hungry) FPS level-1 kernel and the (HBM-bound, 26-VGPR) group kernel share CUs?
Runs both on separate streams in both enqueue orders and compares with the serial time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import _lib, hotpath, synth
import numpy as np
dev = torch.device("cuda"); L = _lib.lib(); B = 256
_lib.set_tuning("fps_bucket_config", (512, 47))
scans = synth.scan_batch(4, 24000, "arch", 5)
xyz = torch.from_numpy(np.concatenate([scans[:, :, :3]] * 64)).to(dev).contiguous()
idx = torch.empty(B, 4096, dtype=torch.int32, device=dev); nx = torch.empty(B, 4096, 3, device=dev)
N2, S2, K, D = 4096, 1024, 32, 128
x2 = torch.rand(B, N2, 3, device=dev); nx2 = x2[:, :S2].contiguous(); p2 = torch.randn(B, N2, D, device=dev)
gi = ((torch.randint(0, N2, (B, S2, 1), device=dev) + torch.randint(0, 256, (B, S2, K), device=dev)) % N2).int().contiguous()
out = torch.empty(B, S2, K, 3 + D, device=dev)
def fps(st): _lib.check(L.tgn_furthestsampling_dense(B, 24000, 4096, _lib.ptr(xyz), None, _lib.ptr(idx), _lib.ptr(nx), _lib.FPS_LOCAL_INDEX, st))
def grp(st): _lib.check(L.tgn_group_points(B, N2, S2, K, D, _lib.ptr(x2), _lib.ptr(nx2), _lib.ptr(p2), _lib.ptr(gi), 0, 1, _lib.ptr(out), st))
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
sa = torch.cuda.Stream(priority=-1); sb = torch.cuda.Stream(priority=0)
pa = _lib.c_void_p(sa.cuda_stream); pb = _lib.c_void_p(sb.cuda_stream)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(2): fps(pa); grp(pb)
print("fps alone      %.2f ms" % timed(lambda: fps(pa)))
print("3x group alone %.2f ms" % timed(lambda: [grp(pb) for _ in range(3)]))
print("fps first, then 3x group on other stream  %.2f ms" % timed(lambda: (fps(pa), [grp(pb) for _ in range(3)])))
print("3x group first, then fps on other stream  %.2f ms" % timed(lambda: ([grp(pb) for _ in range(3)], fps(pa))))
print("same stream serial                         %.2f ms" % timed(lambda: (fps(pa), [grp(pa) for _ in range(3)])))
