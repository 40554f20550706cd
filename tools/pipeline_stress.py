#!/usr/bin/env python3
"""Race check of the pipelined schedule at full size: 40 steps over 4 different resident batches, every step's outputs
compared with the one-stream results of the same batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import hotpath, synth
dev = torch.device("cuda"); B = 256
batches = []
for s in range(4):
    pts = torch.from_numpy(synth.scan_batch(8, 24000, "arch", 500 + s)).to(dev).repeat(B // 8, 1, 1).contiguous()
    feats = [pts, torch.randn(B, 4096, 128, device=dev), torch.randn(B, 1024, 512, device=dev)]
    batches.append((pts[:, :, :3].contiguous(), feats))
ref = hotpath.HotPath(B, dev)
sums = []
for xyz, feats in batches:
    lv = ref.run(xyz, feats)
    sums.append([(l["fps_idx"].long().sum(), l["group_idx"].long().sum(), l["grouped"].double().sum()) for l in lv])
torch.cuda.synchronize()
# The planner double-buffers: the results of step k live until step k+2 is enqueued (a caller that reads them
# asynchronously has to order that read before its call k+2 itself).  So: enqueue k and k+1 back to back -- they overlap on
# the two streams -- then compare both result sets, 20 pairs per configuration.
for fp in (False, True):
    hp = hotpath.HotPath(B, dev, pipeline=True, fps_prefix=fp)
    bad = 0
    for pair in range(20):
        outs = []
        for step in (2 * pair, 2 * pair + 1):
            xyz, feats = batches[(step * 3 + pair) % 4]
            outs.append((hp.run(xyz, feats, inputs_on_current_stream=False), sums[(step * 3 + pair) % 4]))
        torch.cuda.synchronize()
        for lv, want in outs:
            for l, (a, b, c) in zip(lv, want):
                bad += int(l["fps_idx"].long().sum() != a) + int(l["group_idx"].long().sum() != b) + int(l["grouped"].double().sum() != c)
    print(f"pipelined, fps_prefix={fp}: 40 steps in overlapping pairs, mismatching checksums: {bad}")
    # chains of five steps without a host synchronisation in between: buffer set p is rewritten by step k+2 while the
    # groupings of step k+1 are still running beside it -- only the last two steps of a chain can be read back
    bad = 0
    for chain in range(8):
        outs = []
        for step in range(5):
            xyz, feats = batches[(chain + step * 3) % 4]
            outs.append((hp.run(xyz, feats, inputs_on_current_stream=False), sums[(chain + step * 3) % 4]))
        torch.cuda.synchronize()
        for lv, want in outs[-2:]:
            for l, (a, b, c) in zip(lv, want):
                bad += int(l["fps_idx"].long().sum() != a) + int(l["group_idx"].long().sum() != b) + int(l["grouped"].double().sum() != c)
    print(f"pipelined, fps_prefix={fp}: 8 chains of 5 steps, mismatching checksums in the last two steps of each: {bad}")
