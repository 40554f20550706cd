"""Seeded parameter fill shared by the golden generators (build container, reference classes) and the GPU tests (drop-in
mirrors): every tensor of a state_dict is drawn from its own CPU generator, seeded by crc32(name) ^ seed, so two modules
with the same parameter names and shapes get bit-identical values whatever order their constructors ran in -- the
whole-network fixtures (35 MB and 31 MB of weights) need no weight files.  BatchNorm running statistics are drawn too
(eval mode would otherwise be the identity), and the zero-initialised head weights of pointnet_pp.py:34-35 get values."""
import zlib

import torch


def seeded_fill(module, seed):
    sd = module.state_dict()
    names = []
    for name, t in sd.items():
        if not t.is_floating_point():
            continue                                     # num_batches_tracked
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = torch.rand(t.shape, generator=g) * 1.5 + 0.5
        elif leaf == "running_mean":
            v = torch.randn(t.shape, generator=g) * 0.2
        elif leaf == "weight" and t.dim() == 1:          # BatchNorm gamma
            v = torch.rand(t.shape, generator=g) + 0.5
        elif leaf == "bias":
            v = torch.randn(t.shape, generator=g) * 0.1
        else:                                            # Linear / Conv weight: variance-preserving
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.4 / fan_in ** 0.5)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
        names.append(f"{name}:{'x'.join(map(str, t.shape))}")
    return names
