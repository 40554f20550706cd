#!/usr/bin/env python3
"""Pipelined (phased, three streams) vs one-stream HotPath on odd configurations: batch sizes that are not multiples of 8,
int64 indices, Shape B (multi-scale, [features, xyz] layout), with and without the FPS identity shortcut.  Every output of
five back-to-back pipelined steps over alternating inputs must equal the one-stream result bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toothgroupnetwork_amd import hotpath, synth
dev = torch.device("cuda")
bad = 0
for shape_name, shape in (("A", hotpath.SHAPE_A), ("B", hotpath.SHAPE_B)):
    for B, idt, prefix in ((5, torch.int32, False), (20, torch.int64, False), (36, torch.int32, True), (100, torch.int32, False)):
        ins = []
        for s in range(2):
            pts = torch.from_numpy(synth.scan_batch(min(B, 4), shape["n"], "arch", 700 + s)).to(dev).repeat((B + 3) // 4, 1, 1)[:B].contiguous()
            feats = [pts] + [torch.randn(B, S, D, device=dev) for S, D in zip(shape["npoint"][:-1], shape["d"][1:])]
            ins.append((pts[:, :, :3].contiguous(), feats))
        ref = hotpath.HotPath(B, dev, shape=shape, index_dtype=idt, fps_prefix=prefix)
        want = []
        for xyz, feats in ins:
            lv = ref.run(xyz, feats)
            torch.cuda.synchronize()
            want.append([[t.clone() for l in lv for t in [l["fps_idx"], l["new_xyz"]] + [x for br in l["branches"] for x in (br["group_idx"], br["grouped"])]]])
        hp = hotpath.HotPath(B, dev, shape=shape, index_dtype=idt, pipeline=True, fps_prefix=prefix)
        outs = []
        for step in range(5):
            xyz, feats = ins[step & 1]
            lv = hp.run(xyz, feats, inputs_on_current_stream=False)
            outs.append((lv, step & 1))
        torch.cuda.synchronize()
        n_bad = 0
        for lv, which in outs[-2:]:
            got = [t for l in lv for t in [l["fps_idx"], l["new_xyz"]] + [x for br in l["branches"] for x in (br["group_idx"], br["grouped"])]]
            n_bad += sum(int(not torch.equal(a, b)) for a, b in zip(got, want[which][0]))
        print(f"shape {shape_name} B={B} {str(idt).replace('torch.', '')} prefix={prefix}: mismatching tensors {n_bad}")
        bad += n_bad
        del ref, hp, want, outs
        torch.cuda.empty_cache()
print("TOTAL mismatches", bad)
sys.exit(1 if bad else 0)
