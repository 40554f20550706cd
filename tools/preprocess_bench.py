#!/usr/bin/env python3
"""preprocess_data.py-size FPS: N_raw -> 24000; ms per launch, us per iteration, scans/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from toothgroupnetwork_amd import pointops as P, synth
dev = torch.device("cuda")
for (N, S, B) in [(100000, 24000, 1), (100000, 24000, 64), (200000, 24000, 64)]:
    base = np.stack([synth.arch_cloud(N, s, False) for s in range(2)])
    xyz = torch.from_numpy(np.concatenate([base] * (B // 2 + 1))[:B].reshape(-1, 3)).to(dev)
    off = (torch.arange(1, B + 1, device=dev) * N).int(); noff = (torch.arange(1, B + 1, device=dev) * S).int()
    P.furthestsampling(xyz, off, noff); torch.cuda.synchronize()
    t0 = time.perf_counter(); idx = P.furthestsampling(xyz, off, noff); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N_raw={N} -> {S}, B={B}: {dt * 1e3:.1f} ms  {dt / (S - 1) * 1e6:.2f} us/iter  {B / dt:.1f} scans/s", flush=True)
