"""``pointops_cuda`` -- the module name the reference's pybind extension is installed under
(external_libs/pointops/setup.py:14, src/pointops_api.cpp:12-23).

Same ten functions, same argument order, tensors in / nothing out; each forwards the raw device
pointers to the identically named ``*_cuda_launcher`` symbol of libtgn_pointops.so
(include/tgn_pointops.h section 1) on the current HIP stream.  Code written against the
reference's native module (e.g. the reference's own pointops.py) runs on it unchanged.
"""
import torch

from toothgroupnetwork_amd._lib import c_void_p, lib, ptr, require_cuda


def _launch(name, ints, tensors):
    require_cuda(*tensors)
    L = lib()
    L.tgn_set_default_stream(c_void_p(torch.cuda.current_stream().cuda_stream))
    getattr(L, name)(*[int(v) for v in ints], *[ptr(t) for t in tensors])


def furthestsampling_cuda(b, n, xyz, offset, new_offset, tmp, idx):
    _launch("furthestsampling_cuda_launcher", (b, n), (xyz, offset, new_offset, tmp, idx))


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    _launch("knnquery_cuda_launcher", (m, nsample), (xyz, new_xyz, offset, new_offset, idx, dist2))


def grouping_forward_cuda(m, nsample, c, input, idx, output):
    _launch("grouping_forward_cuda_launcher", (m, nsample, c), (input, idx, output))


def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):
    _launch("grouping_backward_cuda_launcher", (m, nsample, c), (grad_output, idx, grad_input))


def interpolation_forward_cuda(n, c, k, input, idx, weight, output):
    _launch("interpolation_forward_cuda_launcher", (n, c, k), (input, idx, weight, output))


def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):
    _launch("interpolation_backward_cuda_launcher", (n, c, k), (grad_output, idx, weight, grad_input))


def subtraction_forward_cuda(n, nsample, c, input1, input2, idx, output):
    _launch("subtraction_forward_cuda_launcher", (n, nsample, c), (input1, input2, idx, output))


def subtraction_backward_cuda(n, nsample, c, idx, grad_output, grad_input1, grad_input2):
    _launch("subtraction_backward_cuda_launcher", (n, nsample, c), (idx, grad_output, grad_input1, grad_input2))


def aggregation_forward_cuda(n, nsample, c, w_c, input, position, weight, idx, output):
    _launch("aggregation_forward_cuda_launcher", (n, nsample, c, w_c), (input, position, weight, idx, output))


def aggregation_backward_cuda(n, nsample, c, w_c, input, position, weight, idx, grad_output, grad_input,
                              grad_position, grad_weight):
    _launch("aggregation_backward_cuda_launcher", (n, nsample, c, w_c),
            (input, position, weight, idx, grad_output, grad_input, grad_position, grad_weight))
