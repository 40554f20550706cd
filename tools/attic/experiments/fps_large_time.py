"""Large-cloud FPS (raw scans -> 24 000): kernel time of one launch for 1 / 32 / 64 scans of ~108 000 points and one of
250 000.  (profiles/r03_fps_large_cloud.txt holds the A/B against round 2's touched-list kernel, which was removed afterwards.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from toothgroupnetwork_amd import pointops as P, synth

dev = torch.device("cuda")
form = "owner-wave"


def run(ns, m, reps=3):
    xs = [np.ascontiguousarray(synth.arch_cloud(n, seed=i, with_normals=False), dtype=np.float32) for i, n in enumerate(ns)]
    pts = torch.from_numpy(np.concatenate(xs)).to(dev)
    off = torch.from_numpy(np.cumsum(ns).astype(np.int32)).to(dev)
    noff = torch.arange(1, len(ns) + 1, dtype=torch.int32, device=dev) * m
    P.furthestsampling(pts, off, noff)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        idx = P.furthestsampling(pts, off, noff)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, int(idx.sum().item())


for label, ns, m in (("1 scan of 108 000", [108000], 24000), ("32 scans", [108000 - 1000 * (i % 5) for i in range(32)], 24000),
                     ("64 scans", [108000 - 1000 * (i % 5) for i in range(64)], 24000), ("1 scan of 50 000", [50000], 24000),
                     ("1 scan of 250 000", [250000], 24000)):
    ms, chk = run(ns, m)
    print(f"{form:12s} {label:20s} {ms:8.2f} ms   ({ms * 1e3 / m:.2f} us per iteration)   checksum {chk}")
