"""ctypes binding of libtgn_pointops.so (the C ABI declared in include/tgn_pointops.h).

This is the ONLY compute path of the package: there is no CPU or eager-PyTorch fallback.
If the library has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C toothgroupnetwork_amd/csrc``) every operator raises, loudly.
"""
import ctypes
import os
import threading

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a
# queue run one after the other.  The phased HotPath schedule alone keeps four streams busy; with the caller's own stream and a
# copy stream that is six, and a second HotPath in the process used to lose its overlap entirely (7.7 instead of 4.9 ms per
# step, profiles/r06_hw_queues.txt).  The runtime reads the variable once, when the first HIP call initialises it -- after this
# import in every entry point of the package.  An explicit setting of the caller's wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

from . import config

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TGN_LIB_PATH") or os.path.join(_HERE, "csrc", "libtgn_pointops.so")   # TGN_LIB_PATH: A/B builds (tools/)

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
    "tgn_set_fps_mode": (None, [c_int]),
    "tgn_get_fps_mode": (c_int, []),
    "tgn_set_tuning": (c_int, [ctypes.c_char_p, c_int]),
    "tgn_get_tuning": (c_int, [ctypes.c_char_p, c_int]),
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
    "tgn_fps_throughput_workspace_bytes": (c_size_t, [c_int, c_int]),
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
    "tgn_pt_attention_forward": (c_int, [c_int, c_int, c_int, c_int] + [_P] * 18 + [_P]),
    "tgn_pt_softmax_aggregate_forward": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "tgn_pt_softmax_aggregate_backward": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    # section 3
    "tgn_ball_query_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "tgn_ball_query": (c_int, [c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "tgn_stream_delay": (c_int, [c_int, _P]),
    "tgn_slice_columns": (c_int, [ctypes.c_longlong, c_int, c_int, c_int, _P, _P, _P]),
    "tgn_ball_query_build": (c_int, [c_int, c_int, c_int, c_int, c_float, _P, _P, c_size_t, _P]),
    "tgn_ball_query_prebuilt": (c_int, [c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "tgn_group_points": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_group_points_ex": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "tgn_sa_first_layer": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_sa_first_layer_max": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_sa_point_transform": (c_int, [ctypes.c_longlong, c_int, c_int, _P, _P, _P, _P, _P]),
    "tgn_sa_gather_max": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_sa_gather_act": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_sa_direct_supported": (c_int, [c_int, c_int, c_int]),
    "tgn_sa_mlp2_direct_supported": (c_int, [c_int, c_int]),
    "tgn_linear_wgrad_workspace_bytes": (c_size_t, [ctypes.c_longlong, c_int, c_int]),
    "tgn_linear_wgrad": (c_int, [ctypes.c_longlong, c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "tgn_bn_rows_workspace_bytes": (c_size_t, [c_int]),
    "tgn_bn_rows_forward": (c_int, [ctypes.c_longlong, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "tgn_bn_rows_backward": (c_int, [ctypes.c_longlong, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "tgn_sa_mlp2_max": (c_int, [c_int] * 7 + [_P] * 7 + [c_int, _P, _P, _P, c_int, _P]),
    "tgn_sa_mlp2_split_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "tgn_sa_mlp2_split_weights": (c_int, [c_int, c_int, _P, _P, _P]),
    "tgn_sa_point_transform_bf16x3": (c_int, [ctypes.c_longlong, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "tgn_sa_mlp2_max_bf16x3": (c_int, [c_int] * 7 + [_P] * 7 + [c_int, _P, _P, _P, c_int, _P]),
    "tgn_sa_all_chunks": (c_int, [c_int]),
    "tgn_sa_all_mlp2_max": (c_int, [c_int] * 5 + [_P] * 9 + [c_int, _P]),
    "tgn_sa_direct_max": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "tgn_gather_points": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    "tgn_scatter_add_points": (c_int, [c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P]),
    "tgn_three_nn": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "tgn_three_interpolate": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "tgn_three_interpolate_ex": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, _P, _P]),
    "tgn_take_index_error": (c_int, [_P]),
    "tgn_take_index_error_device": (c_int, []),
    "tgn_clear_index_error": (c_int, [_P]),
    "tgn_square_distance": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P]),
    # section 4 (host pointers)
    "tgn_obj_count": (c_int, [ctypes.c_char_p, _P, _P]),
    "tgn_obj_read": (c_int, [ctypes.c_char_p, _P, _P, ctypes.c_longlong, ctypes.c_longlong, _P, _P]),
    "tgn_vertex_normals": (c_int, [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _P]),
    "tgn_scan_open": (c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double, ctypes.c_double, _P, _P, ctypes.c_char_p, c_int]),
    "tgn_scan_take": (c_int, [_P, _P, _P]),
    "tgn_scan_pool_trim": (c_int, []),
}

ERR_INVALID_ARGUMENT, ERR_LAUNCH, ERR_UNSUPPORTED = 1, 2, 3     # TGN_ERR_* of include/tgn_pointops.h

FPS_FMA = 1
FPS_LOCAL_INDEX = 2
FPS_INDEX64 = 4
FPS_THROUGHPUT = 32   # scheduling hint: 4 097 - 32 768-point clouds out of an L2-resident workspace, four workgroups per CU
FPS_LOW_VALU = 16   # scheduling hint (include/tgn_pointops.h): small clouds on the bucket-skipping kernel too
FPS_TREE_TIES = 8
FPS_CUDA_COMPAT = FPS_FMA | FPS_TREE_TIES


# Which farthest-point-sampling arithmetic the Python operators ask for:
#   ties  'first' (default): equal distances -> lowest point index, what torch-CPU's farthest_point_sample_np does
#                 (pointnet2_utils.py:103-118) -- the canonical mode, pinned against the reference's torch code;
#         'tree': equal distances resolved like the reference CUDA kernel's shared-memory tree
#                 (sampling_cuda_kernel.cu:5-10,64-123) -- pinned against that kernel compiled for gfx950 (oracle/_ref);
#   fma   False (default): d = ((dx*dx)+(dy*dy))+(dz*dz), the source order of sampling_cuda_kernel.cu:54;
#         True: fma(dz,dz,fma(dy,dy,dx*dx)), nvcc's presumed contraction (parity unpinned: no CUDA device here).
# Environment: TGN_FPS_TIES=first|tree, TGN_FPS_FMA=0|1; or set_fps_mode() at run time.
_fps_mode = {"ties": os.environ.get("TGN_FPS_TIES", "first").lower(), "fma": os.environ.get("TGN_FPS_FMA", "0") == "1"}


def set_fps_mode(ties=None, fma=None):
    """Select the FPS tie order ('first' | 'tree') and distance contraction used by pointops.furthestsampling,
    pointnet2_utils.farthest_point_sample and the modules built on them.  Returns the previous (ties, fma)."""
    prev = (_fps_mode["ties"], _fps_mode["fma"])
    if ties is not None:
        if ties not in ("first", "tree"):
            raise ValueError("ties must be 'first' or 'tree'")
        _fps_mode["ties"] = ties
    if fma is not None:
        _fps_mode["fma"] = bool(fma)
    if _lib is not None or os.path.exists(LIB_PATH):
        lib().tgn_set_fps_mode(fps_flags())   # the reference-signature launcher (pointops_cuda shim) follows too
    return prev


def get_fps_mode():
    return _fps_mode["ties"], _fps_mode["fma"]


def set_tuning(key, value):
    """Select a kernel variant inside the library (include/tgn_pointops.h: tgn_set_tuning; keys "fps_plain", "fps_config",
    "fps_bucket_config", "fps_cell_bits", "fps_bucket_min", "ball_bitmap", "sa_tile", "knn_memset", "knn_grid_scale").
    (nt, p) pairs are given as tuples.  Returns the previous value.  Experiments and parity tests only."""
    if isinstance(value, (tuple, list)):
        value = int(value[0]) * 256 + int(value[1])
    k = key.encode()
    prev = lib().tgn_get_tuning(k, 0)
    check(lib().tgn_set_tuning(k, int(value)), "tgn_set_tuning")
    return prev


class tuning:
    """`with tuning(fps_plain=1, fps_bucket_min=2048): ...` -- kernel-variant switches for the duration of a block."""

    def __init__(self, **kv):
        self.kv, self.prev = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.prev[k] = set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_tuning(k, v)
        return False


def fps_flags(cuda_compat=False):
    """Flag bits of tgn_furthestsampling* for the current mode; cuda_compat=True forces tree ties + FMA."""
    if cuda_compat:
        return FPS_CUDA_COMPAT
    if _fps_mode["ties"] not in ("first", "tree"):
        raise ValueError(f"TGN_FPS_TIES={_fps_mode['ties']!r}: expected 'first' or 'tree'")
    return (FPS_TREE_TIES if _fps_mode["ties"] == "tree" else 0) | (FPS_FMA if _fps_mode["fma"] else 0)


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


# What the gather family does about an index outside [-N, N) -- where the reference's advanced indexing raises
# (pointnet2_utils.py:56-60; an empty ball yields index N, :136-141).  The kernels latch a flag that belongs to the
# (device, STREAM) they were launched on;
#   TGN_INDEX_CHECK=sync (default): index_points / group_points read their stream's flag after their launch (one 4-byte
#       device->host copy + stream sync) and raise IndexError like the reference's CPU path;
#   TGN_INDEX_CHECK=off: no check, no sync (rows with a bad index are filled from point 0 / zeros);
#       take_index_error() reads the CURRENT stream's flag by hand at a synchronisation point; launches that ran on other
#       streams (HotPath's three, a graph replayed elsewhere than it was captured) are covered by
#       take_index_error_device(), which synchronises the device and reads every flag of it.
#   The mode is config.cfg.index_check; _lib.INDEX_CHECK stays as a live alias of it.


def take_index_error():
    """True (and the flag is cleared) if a gather launched on the CURRENT torch stream of the current device saw an out-of-range
    index since the last call.  Synchronises that stream."""
    return bool(lib().tgn_take_index_error(stream()))


def take_index_error_device():
    """True (and every flag of the device is cleared) if a gather on ANY stream of the current device saw an out-of-range index
    since the flags were last read.  Synchronises the device."""
    return bool(lib().tgn_take_index_error_device())


_check_state = threading.local()


def _checking():
    return config.cfg.index_check != "off" and not torch.cuda.is_current_stream_capturing()   # (a host read cannot be captured)


def begin_index_check():
    """In front of a checked launch: drop whatever UNCHECKED launches latched before it (HotPath, captured graphs, the
    training path's trusted gathers, TGN_INDEX_CHECK=off sections), in stream order and without a synchronisation, so that
    the IndexError raised afterwards belongs to this operator.  No-op inside a deferred_index_check() section."""
    if _checking() and not getattr(_check_state, "depth", 0):
        lib().tgn_clear_index_error(stream())


def raise_on_index_error(what):
    """Behind a checked launch: one 4-byte device->host copy + stream synchronisation (that is the price of the
    reference's error behaviour; TGN_INDEX_CHECK=off drops it).  Inside deferred_index_check() the read is left to the
    end of the section."""
    if not _checking() or getattr(_check_state, "depth", 0):
        return
    if take_index_error():
        raise IndexError(f"{what}: index out of range for the gathered dimension "
                         "(the reference's advanced indexing raises here too, pointnet2_utils.py:56-60)")


class deferred_index_check:
    """`with deferred_index_check("set abstraction"):` -- ONE flag read (one synchronisation) for all the gather launches of
    the section instead of one per launch: the fused eval path of a set-abstraction module issues up to six."""

    def __init__(self, what):
        self.what = what

    def __enter__(self):
        if not getattr(_check_state, "depth", 0):
            begin_index_check()
        _check_state.depth = getattr(_check_state, "depth", 0) + 1
        return self

    def __exit__(self, exc_type, exc, tb):
        _check_state.depth -= 1
        if exc_type is None and not _check_state.depth:
            raise_on_index_error(self.what)
        return False


def as_int(v):
    """The reference passes python ints, numpy ints or 0-d CUDA tensors interchangeably (basic_operators.py:22,30)."""
    if isinstance(v, torch.Tensor):
        return int(v.item())
    return int(v)


config.legacy_attributes(__name__, {"INDEX_CHECK": "index_check"})
