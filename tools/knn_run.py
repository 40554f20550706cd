#!/usr/bin/env python3
"""pointops.knnquery(36) on one 24 000-point scan, a few launches: the workload of secondary.knn_24000_k36, for counter passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from toothgroupnetwork_amd import pointops as P, synth  # noqa: E402

dev = torch.device("cuda", 0)
xyz = torch.from_numpy(synth.arch_cloud(24000, 1, False)).to(dev)
off = torch.tensor([24000], dtype=torch.int32, device=dev)
for _ in range(4):
    P.knn_cache_clear()
    P.knnquery(36, xyz, xyz, off, off)
torch.cuda.synchronize()
print("ok")
