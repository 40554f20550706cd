"""ctypes front-end to oracle/_ref/libpointops_ref.so: the REFERENCE's own kernels
(external_libs/pointops/src/*/*_cuda_kernel.cu), compiled for gfx950 from the sources where they lie
(oracle/Makefile target `ref`, -ffp-contract=off).  TEST INFRASTRUCTURE ONLY, GPU box only.

It lets the `-m gpu` tests pin the HIP kernels against the reference's code running on the same MI355X:
FPS (tree tie order), kNN (heap order), grouping / interpolation / subtraction / aggregation fwd+bwd.
The launchers use the null stream and report nothing (as in the reference), so callers synchronise.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "libpointops_ref.so")
_lib = None


def available():
    return os.path.exists(SO) and torch.cuda.is_available()


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _call(name, ints, tensors):
    torch.cuda.synchronize()
    fn = getattr(lib(), name)
    fn.restype = None
    fn(*[ctypes.c_int(int(v)) for v in ints], *[_p(t) for t in tensors])
    torch.cuda.synchronize()


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:10-24 executed with the reference kernel."""
    off = offset.cpu().tolist()
    n_max, prev = 0, 0
    for v in off:
        n_max, prev = max(n_max, v - prev), v
    m = int(new_offset[-1].item())
    idx = torch.zeros(m, dtype=torch.int32, device=xyz.device)
    tmp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
    _call("furthestsampling_cuda_launcher", (offset.shape[0], n_max), (xyz, offset, new_offset, tmp, idx))
    return idx


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    m = new_xyz.shape[0]
    idx = torch.zeros(m, nsample, dtype=torch.int32, device=xyz.device)
    dist2 = torch.zeros(m, nsample, dtype=torch.float32, device=xyz.device)
    _call("knnquery_cuda_launcher", (m, nsample), (xyz, new_xyz, offset, new_offset, idx, dist2))
    return idx, dist2


def grouping_forward(inp, idx):
    m, ns = idx.shape
    c = inp.shape[1]
    out = torch.empty(m, ns, c, dtype=torch.float32, device=inp.device)
    _call("grouping_forward_cuda_launcher", (m, ns, c), (inp, idx, out))
    return out


def grouping_backward(grad_output, idx, n):
    m, ns, c = grad_output.shape
    gi = torch.zeros(n, c, dtype=torch.float32, device=grad_output.device)
    _call("grouping_backward_cuda_launcher", (m, ns, c), (grad_output, idx, gi))
    return gi


def interpolation_forward(inp, idx, weight):
    n, k = idx.shape
    c = inp.shape[1]
    out = torch.zeros(n, c, dtype=torch.float32, device=inp.device)
    _call("interpolation_forward_cuda_launcher", (n, c, k), (inp, idx, weight, out))
    return out


def interpolation_backward(grad_output, idx, weight, m):
    n, c = grad_output.shape
    gi = torch.zeros(m, c, dtype=torch.float32, device=grad_output.device)
    _call("interpolation_backward_cuda_launcher", (n, c, idx.shape[1]), (grad_output, idx, weight, gi))
    return gi


def subtraction_forward(input1, input2, idx):
    n, c = input1.shape
    ns = idx.shape[1]
    out = torch.zeros(n, ns, c, dtype=torch.float32, device=input1.device)
    _call("subtraction_forward_cuda_launcher", (n, ns, c), (input1, input2, idx, out))
    return out


def subtraction_backward(idx, grad_output):
    n, ns, c = grad_output.shape
    g1 = torch.zeros(n, c, dtype=torch.float32, device=grad_output.device)
    g2 = torch.zeros(n, c, dtype=torch.float32, device=grad_output.device)
    _call("subtraction_backward_cuda_launcher", (n, ns, c), (idx, grad_output, g1, g2))
    return g1, g2


def aggregation_forward(inp, position, weight, idx):
    n, ns, c = position.shape
    out = torch.zeros(n, c, dtype=torch.float32, device=inp.device)
    _call("aggregation_forward_cuda_launcher", (n, ns, c, weight.shape[-1]), (inp, position, weight, idx, out))
    return out


def aggregation_backward(inp, position, weight, idx, grad_output):
    n, ns, c = position.shape
    gi, gp, gw = torch.zeros_like(inp), torch.zeros_like(position), torch.zeros_like(weight)
    _call("aggregation_backward_cuda_launcher", (n, ns, c, weight.shape[-1]),
          (inp, position, weight, idx, grad_output, gi, gp, gw))
    return gi, gp, gw
