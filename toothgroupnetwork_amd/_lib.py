"""ctypes binding of libtgn_pointops.so (the C ABI declared in include/tgn_pointops.h).

This is the ONLY compute path of the package: there is no CPU or eager-PyTorch fallback.
If the library has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C toothgroupnetwork_amd/csrc``) every operator raises, loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtgn_pointops.so")

_lib = None

c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/tgn_pointops.h one to one
_P = c_void_p
SIGNATURES = {
    "tgn_version": (ctypes.c_char_p, []),
    "tgn_last_error": (ctypes.c_char_p, []),
    "tgn_set_default_stream": (None, [_P]),
    # section 1: the reference's launchers
    "furthestsampling_cuda_launcher": (None, [c_int, c_int, _P, _P, _P, _P, _P]),
    "knnquery_cuda_launcher": (None, [c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "grouping_forward_cuda_launcher": (None, [c_int, c_int, c_int, _P, _P, _P]),
    "grouping_backward_cuda_launcher": (None, [c_int, c_int, c_int, _P, _P, _P]),
    "interpolation_forward_cuda_launcher": (None, [c_int, c_int, c_int, _P, _P, _P, _P]),
    "interpolation_backward_cuda_launcher": (None, [c_int, c_int, c_int, _P, _P, _P, _P]),
    "subtraction_forward_cuda_launcher": (None, [c_int, c_int, c_int, _P, _P, _P, _P]),
    "subtraction_backward_cuda_launcher": (None, [c_int, c_int, c_int, _P, _P, _P, _P]),
    "aggregation_forward_cuda_launcher": (None, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "aggregation_backward_cuda_launcher": (None, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    # section 2
    "tgn_furthestsampling": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "tgn_furthestsampling_dense": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "tgn_fps_resident_capacity": (c_int, []),
    "tgn_fps_workspace_bytes": (c_size_t, [c_int, c_int]),
    "tgn_furthestsampling_ws": (c_int, [c_int, c_int, _P, _P, _P, _P, c_size_t, _P, _P, c_int, _P]),
    "tgn_furthestsampling_dense_ws": (c_int, [c_int, c_int, c_int, _P, _P, c_size_t, _P, _P, c_int, _P]),
    "tgn_furthestsampling_prefix": (c_int, [c_int, c_int, _P, _P, _P, _P, c_size_t, _P, _P, _P, _P, _P, c_int, _P]),
    "tgn_furthestsampling_dense_prefix": (c_int, [c_int, c_int, c_int, _P, _P, c_size_t, _P, _P, _P, _P, _P, c_int, _P]),
    "tgn_knnquery": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "tgn_knnquery_workspace_bytes": (c_size_t, [c_int]),
    "tgn_knnquery_ws": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "tgn_knnquery_grid_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tgn_knnquery_grid": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "tgn_grouping_forward": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P]),
    "tgn_grouping_backward": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P]),
    "tgn_interpolation_forward": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "tgn_interpolation_backward": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "tgn_subtraction_forward": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "tgn_subtraction_backward": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "tgn_aggregation_forward": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "tgn_aggregation_backward": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    # section 3
    "tgn_ball_query_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tgn_ball_query": (c_int, [c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "tgn_group_points": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_sa_first_layer": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_sa_first_layer_max": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_gather_points": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    "tgn_scatter_add_points": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    "tgn_three_nn": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "tgn_three_interpolate": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "tgn_take_index_error": (c_int, [_P]),
    "tgn_square_distance": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P]),
}

FPS_FMA = 1
FPS_LOCAL_INDEX = 2
FPS_INDEX64 = 4
FPS_TREE_TIES = 8
FPS_CUDA_COMPAT = FPS_FMA | FPS_TREE_TIES


class TgnLibraryError(RuntimeError):
    pass


def lib():
    """Load the HIP library; raise if it is missing (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgnLibraryError(
                f"{LIB_PATH} not found: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C toothgroupnetwork_amd/csrc`). "
                "toothgroupnetwork_amd has no CPU or eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().tgn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libtgn_pointops {what} failed (status {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    """Every operator needs its tensors on the GPU: this package ships no CPU path."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "toothgroupnetwork_amd operators run only on a ROCm GPU (got a CPU tensor); "
                "there is deliberately no CPU fallback -- the CPU restatement lives in oracle/ and is test-only.")


def as_int(v):
    """The reference passes python ints, numpy ints or 0-d CUDA tensors interchangeably (basic_operators.py:22,30)."""
    if isinstance(v, torch.Tensor):
        return int(v.item())
    return int(v)
