"""Per-shape time of the fused BatchNorm(+ReLU) rows kernels (csrc/bnorm.hip) against nn.BatchNorm1d + relu, forward and backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from toothgroupnetwork_amd import point_transformer as PT
dev = torch.device("cuda")

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for rows, C in ((864000, 32), (864000, 4), (864000, 3), (144000, 64), (144000, 8), (36000, 128), (24000, 32), (6000, 64), (1500, 128), (375, 256), (93, 512)):
    x = torch.randn(rows, C, device=dev, requires_grad=True)
    dy = torch.randn(rows, C, device=dev)
    bn = torch.nn.BatchNorm1d(C).to(dev).train()
    res = []
    for fused in (True, False):
        PT.BN_ROWS = fused
        y = PT.bn_rows(bn, x, relu=True)
        f = t(lambda: PT.bn_rows(bn, x, relu=True))
        b = t(lambda: torch.autograd.grad(y, (x, bn.weight, bn.bias), dy, retain_graph=True))
        res += [f, b]
    print(f"rows {rows:7d} C {C:4d}: fused fwd {res[0]:7.1f} us bwd {res[1]:7.1f} us | torch fwd {res[2]:7.1f} us bwd {res[3]:7.1f} us   ({rows * C * 4 / 1e6:.1f} MB)")
