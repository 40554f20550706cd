#!/usr/bin/env python3
"""Bucket-FPS instrumentation: touched buckets per iteration and per-phase cycles of wave 0 (flag 0x100)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from toothgroupnetwork_amd import _lib, synth

dev = torch.device("cuda")
L = _lib.lib()
for (N, S, B) in [(24000, 4096, 1), (24000, 4096, 256), (4096, 1024, 256)]:
    xyz = torch.from_numpy(np.stack([synth.arch_cloud(N, s, False) for s in range(min(B, 4))])).to(dev).repeat((B + 3) // 4, 1, 1)[:B].contiguous()
    idx = torch.empty(B, S, dtype=torch.int32, device=dev)
    st = torch.zeros(max(B * N, 64), dtype=torch.float32, device=dev)
    _lib.check(L.tgn_furthestsampling_dense(B, N, S, _lib.ptr(xyz), _lib.ptr(st), _lib.ptr(idx), None, _lib.FPS_LOCAL_INDEX | 0x100, _lib.stream()))
    torch.cuda.synchronize()
    v = st[:32].view(torch.int64).cpu().numpy()
    it = max(int(v[6]), 1)
    if v[10]:
        print(f"   slowest wave per iteration: {v[10] / it:.0f} cycles before the barrier; it is last iteration's winner in {100 * v[11] / it:.0f} % "
              f"of the iterations, searched for a new candidate in {100 * v[12] / it:.0f} %, had {v[13] / it:.2f} touched buckets; "
              f"last iteration's winner needs {v[14] / it:.0f} cycles")
    print(f"N={N} S={S} B={B}: refresh-skipped/iter/cloud={v[9] / it / B:.2f} touched buckets/iter/cloud={v[0] / it / B:.2f}  waves touched/iter/cloud={v[1] / it / B:.2f}  "
          f"cycles/iter wave0: A={v[2] / it:.0f} update={v[3] / it:.0f} cand={v[4] / it:.0f} C={v[5] / it:.0f} (rec write {v[7] / it:.0f}, barrier wait {v[8] / it:.0f})")
