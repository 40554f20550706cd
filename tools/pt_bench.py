#!/usr/bin/env python3
"""Point-Transformer-shape operator timings (SURVEY.md 3.3 / BASELINE config 4): kNN, queryandgroup, interpolation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from toothgroupnetwork_amd import pointops as P, synth

dev = torch.device("cuda")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        P.knn_cache_clear()  # time the kernels, not the memo
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)

xyz = torch.from_numpy(synth.arch_cloud(24000, 1, False)).to(dev)
off = torch.tensor([24000], dtype=torch.int32, device=dev)
levels = [xyz]; offs = [off]
for n in (6000, 1500, 375, 93):
    noff = torch.tensor([n], dtype=torch.int32, device=dev)
    idx = P.furthestsampling(levels[-1], offs[-1], noff)
    levels.append(levels[-1][idx.long()].contiguous()); offs.append(noff)
for (qi, si, k) in [(0, 0, 36), (1, 0, 24), (1, 1, 24), (2, 1, 24), (2, 2, 24), (0, 1, 3), (0, 0, 1), (0, 0, 16)]:
    q, s = levels[qi], levels[si]
    ms = timeit(lambda: P.knnquery(k, s, q, offs[si], offs[qi]))
    print(f"knn m={q.shape[0]:6d} n={s.shape[0]:6d} k={k:3d}: {ms:8.3f} ms  ({q.shape[0] * s.shape[0] / ms / 1e6:8.1f} G pair/s)", flush=True)
feat = torch.randn(24000, 32, device=dev)
idx, _ = P.knnquery(36, xyz, xyz, off, off)
print("queryandgroup (24000,36,35):", timeit(lambda: P.queryandgroup(36, xyz, xyz, feat, idx, off, off)), "ms")
print("interpolation 6000->24000 c=32:", timeit(lambda: P.interpolation(levels[1], xyz, torch.randn(6000, 32, device=dev), offs[1], off)), "ms")
