"""MI355X implementation of the reference's ``external_libs/pointnet2_utils/pointnet2_utils.py``.

Every public name of the reference module (pointnet2_utils.py:8-352) is provided with the same
signature, layouts (dense ``(B, N, C)`` / channel-first modules) and dtypes (int64 indices), so
``models/modules/pointnet_pp.py``, ``tsg_centroid_module.py``, ``tsg_seg_module.py``, ``tsegnet.py``,
``tgn_loss.py`` and ``tsg_loss.py`` import it unchanged.

What differs is how the work is done: FPS, ball query, the gather+centre+concat of grouping, three_nn
and three_interpolate are single HIP kernels (include/tgn_pointops.h section 3) instead of chains of
torch kernels that materialise (B,S,N) matrices and sort them.  The shared MLPs stay nn.Conv/BatchNorm
layers with the reference's parameter names, so state_dicts are interchangeable.

Retained reference fallbacks.  About 25 lines of the reference's own torch logic live on here, on paths the fast kernels do
not cover, because they ARE the semantics to preserve there: the three ``nn.Module`` constructors (sub-module names and order
decide the state_dict keys), the train-mode tails ``F.relu(bn(conv(x)))`` of the set-abstraction / feature-propagation
stacks (pointnet2_utils.py:229-236, :289-294, :345-351 -- BatchNorm needs batch statistics there), the general-C form of
``square_distance`` (:36-40; the kernel covers C = 3) and the sort-based three-nearest-neighbour weights for DIFFERENTIABLE
coordinates (:333-340; no reference model has them).  Everything on the eval and C = 3 paths is this package's own.
"""
import os
from time import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _derived, _lib, config
from . import _fps_prefix
from ._fps_prefix import PrefixBook
from ._lib import as_int, check, lib, ptr, require_cuda, stream

_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


def timeit(tag, t):
    print("{}: {}s".format(tag, time() - t))
    return time()


def pc_normalize(pc):
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    pc = pc / m
    return pc


def _f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------
# square_distance
# ---------------------------------------------------------------------------------------------
class _SquareDistance3(Function):
    """square_distance for 3-D points with the reference's exact (torch-CPU) arithmetic:
    dot = fma(z1,z2,fma(y1,y2,x1*x2)); d = ((-2*dot) + |src|^2) + |dst|^2   (pointnet2_utils.py:20-41)."""

    @staticmethod
    @_fwd
    def forward(ctx, src, dst):
        src, dst = _f32c(src), _f32c(dst)
        B, N, _ = src.shape
        if dst.shape[0] != B:
            raise RuntimeError(f"square_distance: batch sizes differ ({B} vs {dst.shape[0]})")
        M = dst.shape[1]
        out = torch.empty(B, N, M, dtype=torch.float32, device=src.device)
        check(lib().tgn_square_distance(B, N, M, ptr(src), ptr(dst), ptr(out), stream()), "square_distance")
        ctx.save_for_backward(src, dst)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        src, dst = ctx.saved_tensors
        g = g.float()
        g_src = g_dst = None
        if ctx.needs_input_grad[0]:
            g_src = 2.0 * (src * g.sum(-1, keepdim=True) - torch.matmul(g, dst))
        if ctx.needs_input_grad[1]:
            g_dst = 2.0 * (dst * g.sum(-2).unsqueeze(-1) - torch.matmul(g.transpose(1, 2), src))
        return g_src, g_dst


def square_distance(src, dst):
    """All-pairs squared distances, src (B,N,C) x dst (B,M,C) -> (B,N,M), in the expanded form
    -2*src.dst + |src|^2 + |dst|^2 of the reference (so values can be slightly negative, as there)."""
    require_cuda(src, dst)
    if src.shape[-1] == 3 and dst.shape[-1] == 3:
        return _SquareDistance3.apply(src, dst)
    # C != 3 never occurs in the reference models; same formula, generic C (pointnet2_utils.py:38-41)
    B, N, _ = src.shape
    _, M, _ = dst.shape
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


# ---------------------------------------------------------------------------------------------
# index_points
# ---------------------------------------------------------------------------------------------
class _IndexPoints(Function):
    @staticmethod
    @_fwd
    def forward(ctx, points, idx):
        B, N, C = points.shape
        M = idx.numel() // B if B else 0
        out = torch.empty(tuple(idx.shape) + (C,), dtype=torch.float32, device=points.device)
        _lib.begin_index_check()
        check(lib().tgn_gather_points(B, N, M, C, ptr(points), ptr(idx), int(idx.dtype == torch.int64), ptr(out),
                                      stream()), "gather_points")
        _lib.raise_on_index_error("index_points")
        ctx.shape = (B, N, M, C)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        B, N, M, C = ctx.shape
        g = g.contiguous().float()
        grad = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        check(lib().tgn_scatter_add_points(B, N, M, C, ptr(g), ptr(idx), int(idx.dtype == torch.int64), ptr(grad),
                                           stream()), "scatter_add_points")
        return grad, None


def index_points(points, idx):
    """points (B,N,C) gathered with idx (B,S) or (B,S,K) (int64/int32, per-cloud indices) -> (B,S[,K],C);
    differentiable w.r.t. points (scatter-add backward).  Negative indices wrap and an index outside [-N, N) raises
    IndexError, as torch's advanced indexing does in the reference (pointnet2_utils.py:56-60); the check costs one
    stream synchronisation per call (TGN_INDEX_CHECK=off drops it: such rows are then zero-filled and
    _lib.take_index_error() reports them)."""
    require_cuda(points, idx)
    if idx.dtype not in (torch.int64, torch.int32):
        idx = idx.long()
    points = _f32c(points)
    out = _IndexPoints.apply(points, idx.contiguous())
    return out


# ---------------------------------------------------------------------------------------------
# FPS of an FPS result is the identity (include/tgn_pointops.h, tgn_furthestsampling_dense_prefix).  Every
# set-abstraction level after the first samples the previous level's new_xyz (pointnet2_utils.py:160 /:276), so the
# kernel that produced new_xyz also leaves a per-cloud certificate.  A later FPS whose input has the shape of a recent
# result is offered that result and its certificate; the kernel compares the input with the stored coordinates bit for
# bit, per cloud, and only then writes 0..S-1 instead of iterating -- provenance by content, nothing is assumed about
# how the tensor travelled (module round trips, index_points(xyz, fps_idx), copies).  Opt-in per call site: the
# set-abstraction modules in eval mode ask for it, the plain operators do not (TGN_FPS_PREFIX=1 / 0 forces it on / off).
# ---------------------------------------------------------------------------------------------
FPS_PREFIX = _fps_prefix.FORCE     # None: opt-in per call site (the modules below); True / False: forced (TGN_FPS_PREFIX)
_fps_book = PrefixBook()
fps_prefix_stats = _fps_book.stats


def fps_prefix_clear():
    _fps_book.clear()


# ---------------------------------------------------------------------------------------------
# farthest point sampling
# ---------------------------------------------------------------------------------------------
def _fps_dense(xyz, npoint, want_coords=False, cuda_compat=False, prefix=False, mode=None):
    """mode: explicit tie-order / contraction flag bits (None: the process-wide _lib.set_fps_mode() setting)."""
    require_cuda(xyz)
    npoint = as_int(npoint)
    xyz = _f32c(xyz.detach())
    B, N, _ = xyz.shape
    mode = _lib.fps_flags(cuda_compat) if mode is None else int(mode)
    # prefix: this call site chains sampling levels (FPS of an FPS result is the identity); the tree tie order breaks it
    use_prefix = _fps_prefix.use_prefix(prefix, FPS_PREFIX) and not (mode & _lib.FPS_TREE_TIES)
    idx = torch.empty(B, npoint, dtype=torch.int64, device=xyz.device)
    new_xyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device) if (want_coords or use_prefix) else None
    if B == 0 or npoint == 0:
        return idx, (new_xyz if want_coords else None)
    from .pointops import fps_workspace
    ws, nbytes = fps_workspace(B, N, B * N, xyz.device)
    flags = _lib.FPS_LOCAL_INDEX | _lib.FPS_INDEX64 | mode
    if nbytes == 0 and B >= 3 * torch.cuda.get_device_properties(xyz.device).multi_processor_count:
        # three or more clouds per CU: the L2-resident form shares a CU between four workgroups and finishes the BATCH sooner
        # (24 000 -> 4096: 10.9 against 12.7 us per scan at 768 scans, 9.7 at 1024; profiles/r06_fps_throughput.txt); same results
        nbytes = int(lib().tgn_fps_throughput_workspace_bytes(B, N))
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
            flags |= _lib.FPS_THROUGHPUT
    cert_in = ref = cert_out = None
    if use_prefix:
        cert_in, ref = _fps_book.offer((B, N, mode), xyz.device)
        cert_out = torch.empty(B, dtype=torch.int32, device=xyz.device)
    check(lib().tgn_furthestsampling_dense_prefix(B, N, npoint, ptr(xyz), ptr(ws), nbytes, ptr(idx), ptr(new_xyz),
                                                  ptr(cert_in), ptr(ref), ptr(cert_out), flags, stream()),
          "tgn_furthestsampling_dense")
    if use_prefix:
        _fps_book.record((B, npoint, mode), xyz.device, new_xyz, cert_out, shared=want_coords)
    return idx, (new_xyz if want_coords else None)


def farthest_point_sample(xyz, npoint):
    """(B,N,3) -> (B,npoint) int64 indices local to each cloud; the first sample is point 0, as in the kernel the
    reference calls (pointnet2_utils.py:87-98 / sampling_cuda_kernel.cu:39).  Tie order / contraction follow
    _lib.set_fps_mode() (TGN_FPS_TIES, TGN_FPS_FMA): 'first' = torch-CPU semantics (default), 'tree' = the
    reference CUDA kernel's reduction order."""
    return _fps_dense(xyz, npoint)[0]


def farthest_point_sample_np(xyz, npoint):
    """numpy in / numpy out variant with a RANDOM first sample (pointnet2_utils.py:103-118), indices bit-identical
    to the reference's torch-CPU loop for the same torch RNG state -- ties included -- for float32 input.  (float64 input:
    the reference keeps the numpy dtype, computes the squared distances in float64 and rounds them to float32 only when
    it stores them (:113-116); here the coordinates are rounded to float32 first, so the arg-max may differ where two
    candidates are closer than float32 resolution.)

    The start is drawn with the reference's own call, torch.randint(0, N, (B,), dtype=torch.long) (:109).  The GPU
    kernel always starts at a cloud's first point, so each cloud is sampled as [p_start, p_0, ..., p_{N-1}]: the copy
    in front makes p_start the first sample, every later arg-max sees the original points in their original order
    (first-index ties as in torch.max, :117), and the copy itself sits at distance 0 like p_start.  Index 0 of the
    padded cloud maps back to `start` for the first sample and to point 0 afterwards (an exhausted or all-NaN cloud,
    where torch.max returns index 0)."""
    xyz_t = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32))
    B, N, _ = xyz_t.shape
    farthest = torch.randint(0, N, (B,), dtype=torch.long)
    first = torch.gather(xyz_t, 1, farthest.view(B, 1, 1).expand(B, 1, 3))
    padded = torch.cat([first, xyz_t], dim=1).to(torch.device("cuda"))
    # this function IS the torch-CPU semantics (first-index ties, unfused distance), whatever mode the kernels are in:
    # the flags go with the call, the process-wide mode is not touched (other threads may be sampling)
    idx = _fps_dense(padded, npoint, mode=0)[0].cpu() - 1
    if idx.shape[1] > 0:
        idx[:, 0] = farthest
    idx.clamp_(min=0)
    return idx.numpy()


# ---------------------------------------------------------------------------------------------
# ball query and grouping
# ---------------------------------------------------------------------------------------------
_ws_cache = {}


def _workspace(nbytes, device):
    if nbytes == 0:
        return None
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def query_ball_point(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3) cloud, new_xyz (B,S,3) centres -> (B,S,nsample) int64: per centre the first `nsample` point
    indices, in ascending index order, whose expanded-form squared distance is <= radius**2; short rows repeat
    their first hit (reference pointnet2_utils.py:120-144, without its (B,S,N) matrix and sort)."""
    require_cuda(xyz, new_xyz)
    nsample = as_int(nsample)
    xyz, new_xyz = _f32c(xyz.detach()), _f32c(new_xyz.detach())
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    # `sqrdists > radius ** 2`: python evaluates radius**2 in double, torch compares in fp32
    r2 = float(torch.tensor(float(radius) ** 2, dtype=torch.float32).item())
    group_idx = torch.empty(B, S, nsample, dtype=torch.int64, device=xyz.device)
    nbytes = int(lib().tgn_ball_query_workspace_bytes(B, N, S))
    ws = _workspace(nbytes, xyz.device)
    check(lib().tgn_ball_query(B, N, S, nsample, r2, ptr(xyz), ptr(new_xyz), ptr(group_idx), 1, ptr(ws), nbytes,
                               stream()), "tgn_ball_query")
    return group_idx


class _GroupPoints(Function):
    """out[b,s,k,:] = concat(xyz[b,idx]-new_xyz[b,s], points[b,idx]) (xyz_first) or the Msg order
    (points first) -- pointnet2_utils.py:162-169 / 281-285 -- in one kernel."""

    @staticmethod
    @_fwd
    def forward(ctx, xyz, new_xyz, points, idx, xyz_first):
        B, N, _ = xyz.shape
        _, S, K = idx.shape
        D = 0 if points is None else points.shape[2]
        out = torch.empty(B, S, K, 3 + D, dtype=torch.float32, device=xyz.device)
        _lib.begin_index_check()
        check(lib().tgn_group_points(B, N, S, K, D, ptr(xyz), ptr(new_xyz), ptr(points), ptr(idx),
                                     int(idx.dtype == torch.int64), int(xyz_first), ptr(out), stream()),
              "group_points")
        _lib.raise_on_index_error("group_points")
        ctx.dims = (B, N, S, K, D, bool(xyz_first))
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        B, N, S, K, D, xyz_first = ctx.dims
        g = g.contiguous().float()
        g_rel = (g[..., :3] if xyz_first else g[..., D:]).contiguous()
        is64 = int(idx.dtype == torch.int64)
        g_xyz = g_new = g_pts = None
        if ctx.needs_input_grad[0]:
            g_xyz = torch.zeros(B, N, 3, dtype=torch.float32, device=g.device)
            check(lib().tgn_scatter_add_points(B, N, S * K, 3, ptr(g_rel), ptr(idx), is64, ptr(g_xyz), stream()),
                  "scatter_add_points")
        if ctx.needs_input_grad[1]:
            g_new = -g_rel.sum(2)
        if D and ctx.needs_input_grad[2]:
            g_f = (g[..., 3:] if xyz_first else g[..., :D]).contiguous()
            g_pts = torch.zeros(B, N, D, dtype=torch.float32, device=g.device)
            check(lib().tgn_scatter_add_points(B, N, S * K, D, ptr(g_f), ptr(idx), is64, ptr(g_pts), stream()),
                  "scatter_add_points")
        return g_xyz, g_new, g_pts, None, None


def group_points(xyz, new_xyz, points, idx, xyz_first=True):
    """Fused gather + centre + concat: (B,S,K,3+D) = [xyz[idx]-new_xyz, points[idx]] (xyz_first, reference :162-169)
    or [points[idx], xyz[idx]-new_xyz] (:281-285).  Raises IndexError where the reference's indexing would -- an
    empty ball yields index N (pointnet2_utils.py:136-141) -- after one stream synchronisation per call;
    TGN_INDEX_CHECK=off drops check and sync (rows with a bad index are then filled from point 0)."""
    require_cuda(xyz, new_xyz, idx)
    out = _GroupPoints.apply(_f32c(xyz), _f32c(new_xyz), None if points is None else _f32c(points),
                             idx.contiguous(), bool(xyz_first))
    return out


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """FPS -> ball query -> fused grouping.  xyz (B,N,3), points (B,N,D) or None ->
    new_xyz (B,npoint,3), new_points (B,npoint,nsample,3+D) in [centred xyz, features] order (reference :168);
    with returnfps also the gathered absolute coordinates and the FPS indices."""
    fps_idx = farthest_point_sample(xyz, npoint)  # [B, npoint]
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    new_points = group_points(xyz, new_xyz, points, idx, xyz_first=True)
    if returnfps:
        grouped_xyz = index_points(xyz, idx)
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """One group holding the whole cloud: new_xyz (B,1,3) zeros, new_points (B,1,N,3+D) = [xyz, points]."""
    device = xyz.device
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C).to(device)
    grouped_xyz = xyz.view(B, 1, N, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points


# ---------------------------------------------------------------------------------------------
# three_nn / three_interpolate
# ---------------------------------------------------------------------------------------------
def three_nn(xyz1, xyz2):
    """3 nearest points of xyz2 (B,S,3) for every point of xyz1 (B,N,3): (dist (B,N,3) squared,
    expanded form, ascending by (dist, index); idx (B,N,3) int64) -- pointnet2_utils.py:333-335."""
    require_cuda(xyz1, xyz2)
    xyz1, xyz2 = _f32c(xyz1.detach()), _f32c(xyz2.detach())
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    dist = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device)
    idx = torch.empty(B, N, 3, dtype=torch.int64, device=xyz1.device)
    check(lib().tgn_three_nn(B, N, S, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), 1, stream()), "three_nn")
    return dist, idx


class _ThreeInterpolate(Function):
    @staticmethod
    @_fwd
    def forward(ctx, points2, dist, idx):
        B, S, C = points2.shape
        N = dist.shape[1]
        out = torch.empty(B, N, C, dtype=torch.float32, device=points2.device)
        weight = torch.empty(B, N, 3, dtype=torch.float32, device=points2.device)
        check(lib().tgn_three_interpolate(B, N, S, C, ptr(points2), ptr(dist), ptr(idx), int(idx.dtype == torch.int64),
                                          ptr(out), ptr(weight), stream()), "three_interpolate")
        ctx.dims = (B, N, S, C)
        ctx.save_for_backward(idx, weight)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, g):
        idx, weight = ctx.saved_tensors
        B, N, S, C = ctx.dims
        g = g.contiguous().float()
        # global row indices into the flattened (B*S, C) support features
        gidx = (idx + (torch.arange(B, device=idx.device, dtype=idx.dtype) * S).view(B, 1, 1)).to(torch.int32)
        grad = torch.zeros(B * S, C, dtype=torch.float32, device=g.device)
        check(lib().tgn_interpolation_backward(B * N, C, 3, ptr(g), ptr(gidx.contiguous()), ptr(weight), ptr(grad),
                                               stream()), "interpolation bwd")
        return grad.view(B, S, C), None, None


def linear_relu(x, W, b):
    """relu(x @ W.T + b) over the last axis of contiguous fp32 rows, the ReLU in the GEMM's epilogue (hipBLASLt through
    torch._addmm_activation: bit-identical to F.linear followed by relu_, one pass over the output less); any other input takes
    those two calls."""
    if (hasattr(torch, "_addmm_activation") and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and W.dtype == torch.float32
            and b is not None and not (torch.is_grad_enabled() and (x.requires_grad or W.requires_grad or b.requires_grad))):
        y = torch._addmm_activation(b, x.reshape(-1, x.shape[-1]), W.t(), use_gelu=False)
        return y.view(*x.shape[:-1], W.shape[0])
    return torch.relu_(F.linear(x, W, b))


def three_interpolate_add_relu(points2, dist, idx, add=None, relu=False):
    """[relu](three_interpolate(points2, dist, idx) (+ add)) in ONE kernel, no autograd: the epilogue of the eval-mode feature
    propagation (first convolution commuted onto the coarse points; `add` = the skip features' share of that convolution plus the
    folded bias).  `add` (B,N,C) is overwritten with the result when given (it is a temporary there)."""
    require_cuda(points2, dist, idx)
    if idx.dtype not in (torch.int64, torch.int32):
        idx = idx.long()
    points2, dist, idx = _f32c(points2), _f32c(dist), idx.contiguous()
    B, S, C = points2.shape
    N = dist.shape[1]
    if add is not None:
        add = _f32c(add)
        assert tuple(add.shape) == (B, N, C)
    out = add if add is not None else torch.empty(B, N, C, dtype=torch.float32, device=points2.device)
    check(lib().tgn_three_interpolate_ex(B, N, S, C, ptr(points2), ptr(dist), ptr(idx), int(idx.dtype == torch.int64), ptr(add),
                                         int(bool(relu)), ptr(out), None, stream()), "three_interpolate_ex")
    return out


def three_interpolate(points2, dist, idx):
    """inverse-(squared)-distance weighted sum of the 3 neighbours (pointnet2_utils.py:337-340) -> (B,N,C)."""
    require_cuda(points2, dist, idx)
    if idx.dtype not in (torch.int64, torch.int32):
        idx = idx.long()
    return _ThreeInterpolate.apply(_f32c(points2), _f32c(dist), idx.contiguous())


# ---------------------------------------------------------------------------------------------
# fused set abstraction (eval mode): the first shared-MLP layer without the grouped (B,S,K,3+D) tensor
# ---------------------------------------------------------------------------------------------
# The switches of this module live in toothgroupnetwork_amd.config (cfg.fused_sa, cfg.commute_fp -- feature propagation: first
# convolution on the coarse points, below --, cfg.sa_bf16x3 -- second layer of the chained set-abstraction kernel on the bf16 matrix
# cores at fp32 accuracy: three-way bf16 split of both operands, six products, include/tgn_pointops.h tgn_sa_mlp2_max_bf16x3; False:
# the exact-fp32 MFMA form).  FUSED_SA / COMMUTE_FP / SA_BF16X3 remain as live aliases of those fields (reads and writes).
config.legacy_attributes(__name__, {"FUSED_SA": "fused_sa", "COMMUTE_FP": "commute_fp", "SA_BF16X3": "sa_bf16x3"})


def _can_fuse(module, *tensors):
    """Eval-mode, no autograd through the inputs: BatchNorm is a fixed affine map and can be folded."""
    if not config.cfg.fused_sa or module.training:
        return False
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        return False
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        return False
    return True


def _fuse_pays(N, S, K, c_in, c1):
    """A MULTI-layer MLP still needs its (B,S,K,C1) first-layer output; the commuted first layer trades the grouped
    tensor (S*K rows of c_in) for a per-point one (N rows of c1): worth it once the former is the larger (sa1 of the
    reference net, c_in = 9 -> 128 channels: it is not; measured 0.89x)."""
    return S * K * c_in > N * c1


def fold_first_layer(conv, bn, D, xyz_first):
    """The first Conv2d(1x1) + eval-mode BatchNorm2d of a shared MLP as the operands of the fused kernels
    (include/tgn_pointops.h): scale = gamma/sqrt(var+eps) goes into the weight columns, shift + scale*bias into b2.
      Wt  (D+3, C1) rows [features..., x, y, z]   -- tgn_sa_point_transform
      Wxs (3, C1)   the x, y, z rows of Wt        -- centre term of tgn_sa_gather_max / tgn_sa_gather_act
      Wd  (16, C1)  rows [x, y, z, features..., 0] -- tgn_sa_direct_max
      b2  (C1,)
    Memoised on `bn` until a parameter / running statistic of the two modules changes (_derived.cached)."""
    return _derived.cached(bn, "first_layer", _derived.sources(conv, bn), (D, bool(xyz_first)),
                           lambda: _fold_first_layer(conv, bn, D, xyz_first))


def _fold_first_layer(conv, bn, D, xyz_first):
    C1 = conv.out_channels
    W = conv.weight.detach().reshape(C1, -1).float()                       # (C1, 3+D) in the module's channel order
    bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(C1, device=W.device)
    scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float()
    shift = (bn.bias.detach() - bn.running_mean * scale).float()
    Wx, Wp = (W[:, :3], W[:, 3:]) if xyz_first else (W[:, D:], W[:, :D])   # sample_and_group: [xyz, feat]; Msg: [feat, xyz]
    Ws = W.new_empty(C1, 3 + D)
    Ws[:, :D], Ws[:, D:] = Wp * scale[:, None], Wx * scale[:, None]
    Wt = Ws.t().contiguous()                                               # (D+3, C1): [features..., x, y, z]
    Wd = W.new_zeros(16, C1)
    if 3 + D <= 16:
        Wd[:3], Wd[3:3 + D] = Wt[D:], Wt[:D]
    return dict(Wt=Wt, Wxs=Wt[D:].contiguous(), Wd=Wd, b2=(shift + scale * bias).contiguous(), C1=C1)


def split_point_transform(Wt):
    """The bf16 x 3 image of a per-point first-layer matrix Wt (D+3, C1) for tgn_sa_point_transform_bf16x3: rows padded with zeros
    to a multiple of 16, arranged (Kp/8, C1, 8) and split like a second layer (tgn_sa_mlp2_split_weights)."""
    Kc, C1 = Wt.shape
    Kp = (Kc + 15) // 16 * 16
    Wp = Wt.new_zeros(Kp, C1)
    Wp[:Kc] = Wt
    return split_second_layer(Wp.view(Kp // 8, 8, C1).permute(0, 2, 1).contiguous()), Kp


def sa_point_transform(xyz, points, Wt, Wts=None):
    """A[b,n,:] = [points[b,n,:], xyz[b,n,:]] @ Wt -- the per-POINT half of a fused first layer, on the matrix cores.
    xyz (B,N,3), points (B,N,D) or None, Wt (D+3, C1) -> (B,N,C1).  Wts = split_point_transform(Wt): the bf16 x 3 form
    (tgn_sa_point_transform_bf16x3, fp32-class rounding at up to 2.7x the rate); None: exact fp32 MFMA (tgn_sa_point_transform)."""
    B, N, _ = xyz.shape
    D = 0 if points is None else points.shape[2]
    C1 = Wt.shape[1]
    A = torch.empty(B, N, C1, dtype=torch.float32, device=xyz.device)
    if Wts is not None and B * N <= 65535 * 128:
        img, Kp = Wts
        check(lib().tgn_sa_point_transform_bf16x3(B * N, D, Kp, C1, ptr(xyz), ptr(points), ptr(img), ptr(A), stream()), "sa_point_transform_bf16x3")
    else:
        check(lib().tgn_sa_point_transform(B * N, D, C1, ptr(xyz), ptr(points), ptr(Wt), ptr(A), stream()), "sa_point_transform")
    return A


def sa_level_max(xyz, new_xyz, points, idx, conv, bn, xyz_first):
    """A whole single-layer set-abstraction level after sampling and ball query:
        max_k relu(bn(conv([xyz[idx]-new_xyz, points[idx]])))  ->  (B,S,C1)
    (pointnet2_utils.py:162-169 + 229-236, or 281-294 for Msg) with nothing of size S*K ever written: narrow inputs go
    through the direct kernel (gather -> matrix cores -> max), wide ones through the per-point transform + gather-max."""
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    f = fold_first_layer(conv, bn, D, xyz_first)
    C1 = f["C1"]
    idx = idx.contiguous()
    out = torch.empty(B, S, C1, dtype=torch.float32, device=xyz.device)
    L = lib()
    _lib.begin_index_check()
    if L.tgn_sa_direct_supported(K, D, C1):
        check(L.tgn_sa_direct_max(B, N, S, K, D, C1, ptr(xyz), ptr(new_xyz), ptr(points), ptr(f["Wd"]), ptr(f["b2"]), ptr(idx),
                                  int(idx.dtype == torch.int64), 1, ptr(out), stream()), "sa_direct_max")
    else:
        A = sa_point_transform(xyz, points, f["Wt"])
        check(L.tgn_sa_gather_max(B, N, S, K, C1, ptr(A), ptr(new_xyz), ptr(f["Wxs"]), ptr(f["b2"]), ptr(idx),
                                  int(idx.dtype == torch.int64), 1, ptr(out), stream()), "sa_gather_max")
    _lib.raise_on_index_error("set abstraction (grouping)")
    return out


def sa_first_layer(xyz, new_xyz, points, idx, conv, bn, xyz_first, reduce_max=False):
    """relu(bn(conv(grouped))) of the FIRST shared-MLP layer without ever building `grouped`
    (pointnet2_utils.py:162-169 + 229-233, or 281-292 for Msg).  The 1x1 convolution commutes with the gather:
        W*[points[idx], xyz[idx]-c] + b = (W_p*points + W_x*xyz)[idx] + (b - W_x*c)
    so the contraction runs over the N points (tgn_sa_point_transform, fp32 MFMA) instead of the S*K grouped rows and
    the per-query kernel only gathers, adds the centre term and applies ReLU.  Returns (B,S,K,C1), or (B,S,C1) with
    reduce_max (= sa_level_max).  Eval-mode BatchNorm statistics are folded in."""
    if reduce_max:
        return sa_level_max(xyz, new_xyz, points, idx, conv, bn, xyz_first)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    f = fold_first_layer(conv, bn, D, xyz_first)
    C1 = f["C1"]
    idx = idx.contiguous()
    A = sa_point_transform(xyz, points, f["Wt"])
    out = torch.empty(B, S, K, C1, dtype=torch.float32, device=xyz.device)
    _lib.begin_index_check()
    check(lib().tgn_sa_gather_act(B, N, S, K, C1, ptr(A), ptr(new_xyz), ptr(f["Wxs"]), ptr(f["b2"]), ptr(idx),
                                  int(idx.dtype == torch.int64), 1, ptr(out), stream()), "sa_gather_act")
    _lib.raise_on_index_error("set abstraction (grouping)")
    return out


def fold_second_layer(conv, bn, C1p):
    """The second Conv2d(1x1) + eval-mode BatchNorm2d of a shared MLP as the operands of tgn_sa_mlp2_max:
      W2f (C1p/8, C2, 8): W2f[kb, c, i] = scale[c] * W[c, 8*kb + i], zero for the padded input channels
      b2  (C2,)          = shift + scale * bias"""
    return _derived.cached(bn, "second_layer", _derived.sources(conv, bn), C1p, lambda: _fold_second_layer(conv, bn, C1p))


def _fold_second_layer(conv, bn, C1p):
    C2, C1 = conv.out_channels, conv.in_channels
    W = conv.weight.detach().reshape(C2, C1).float()
    bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros(C2, device=W.device)
    scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float()
    shift = (bn.bias.detach() - bn.running_mean * scale).float()
    Wp = W.new_zeros(C2, C1p)
    Wp[:, :C1] = W * scale[:, None]
    W2f = Wp.view(C2, C1p // 8, 8).permute(1, 0, 2).contiguous()
    return W2f, (shift + scale * bias).contiguous()


def split_second_layer(W2f):
    """The bf16 x 3 image of a folded second-layer weight matrix W2f (C1p/8, C2, 8) for tgn_sa_mlp2_max_bf16x3."""
    C1p, C2 = W2f.shape[0] * 8, W2f.shape[1]
    img = torch.empty(int(lib().tgn_sa_mlp2_split_bytes(C1p, C2)), dtype=torch.uint8, device=W2f.device)
    check(lib().tgn_sa_mlp2_split_weights(C1p, C2, ptr(W2f), ptr(img), stream()), "sa_mlp2_split_weights")
    return img


def _pad_cols(t, C1p):
    if t.shape[-1] == C1p:
        return t.contiguous()
    out = t.new_zeros(t.shape[:-1] + (C1p,))
    out[..., :t.shape[-1]] = t
    return out


def sa_level_mlp2_max(xyz, new_xyz, points, idx, convs, bns, xyz_first, out=None):
    """A whole set-abstraction level with a TWO-layer shared MLP after sampling and ball query:
        max_k relu(bn2(conv2(relu(bn1(conv1([xyz[idx]-new_xyz, points[idx]]))))))  ->  (B,S,C2)
    (pointnet2_utils.py:162-169 + 229-236, or 281-294 for Msg) in ONE kernel after the per-point transform of the first
    layer (wide inputs) or with the first layer computed from the gathered rows (3+D <= 16): nothing of size S*K is
    written, no torch convolution runs (tgn_sa_mlp2_max; the second layer on the fp32 matrix cores).  out: optional (B,S,C2)
    view into a wider row-major tensor (last stride 1) -- a multi-scale level writes its branches side by side."""
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    L = lib()
    direct = bool(L.tgn_sa_mlp2_direct_supported(K, D))

    def operands():   # (memoised on the second BatchNorm: ~40 small launches per branch otherwise, every forward)
        f = fold_first_layer(convs[0], bns[0], D, xyz_first)
        C1p = (f["C1"] + 15) // 16 * 16
        W2f, b2 = fold_second_layer(convs[1], bns[1], C1p)
        return dict(C1p=C1p, W2f=W2f, b2=b2, b1=_pad_cols(f["b2"], C1p), W2s=split_second_layer(W2f) if config.cfg.sa_bf16x3 else None,
                    W1=_pad_cols(f["Wd"] if direct else f["Wxs"], C1p),      # direct: (16, C1p) rows [x, y, z, features..., 0]
                    Wt=None if direct else _pad_cols(f["Wt"], C1p),
                    Wts=split_point_transform(_pad_cols(f["Wt"], C1p)) if (config.cfg.sa_bf16x3 and not direct) else None)
    ops = _derived.cached(bns[1], "mlp2", _derived.sources(convs[0], bns[0], convs[1], bns[1]), (D, bool(xyz_first), direct, config.cfg.sa_bf16x3),
                          operands)
    C1p, W2f, b2, b1, W1 = ops["C1p"], ops["W2f"], ops["b2"], ops["b1"], ops["W1"]
    C2 = b2.shape[0]
    idx = idx.contiguous()
    if out is None:
        out = torch.empty(B, S, C2, dtype=torch.float32, device=xyz.device)
    assert out.shape == (B, S, C2) and out.stride(2) == 1 and out.stride(0) == S * out.stride(1) and out.dtype == torch.float32
    A1 = None if direct else sa_point_transform(xyz, points, ops["Wt"], ops["Wts"])      # (B, N, C1p)
    _lib.begin_index_check()
    if ops["W2s"] is not None:
        check(L.tgn_sa_mlp2_max_bf16x3(B, N, S, K, D, C1p, C2, ptr(A1), ptr(xyz), ptr(points), ptr(new_xyz), ptr(W1), ptr(b1), ptr(idx),
                                       int(idx.dtype == torch.int64), ptr(ops["W2s"]), ptr(b2), ptr(out), out.stride(1), stream()),
              "sa_mlp2_max_bf16x3")
    else:
        check(L.tgn_sa_mlp2_max(B, N, S, K, D, C1p, C2, ptr(A1), ptr(xyz), ptr(points), ptr(new_xyz), ptr(W1), ptr(b1), ptr(idx),
                                int(idx.dtype == torch.int64), ptr(W2f), ptr(b2), ptr(out), out.stride(1), stream()), "sa_mlp2_max")
    _lib.raise_on_index_error("set abstraction (grouping)")
    return out


def sa_all_mlp2_max(xyz, points, convs, bns):
    """PointNetSetAbstraction(group_all=True) with a two-layer shared MLP, eval mode (pointnet2_utils.py:178-195 + 229-236; the one
    instantiation is tsg_seg_module.py:28, 515 -> [256, 512] over 256 points):
        max_n relu(bn2(conv2(relu(bn1(conv1([xyz_n, points_n]))))))  ->  (B, C2)
    The first layer runs once per point on the fp32 matrix cores (tgn_sa_point_transform; 3+D <= 16: inside the kernel), the second
    layer and the maximum over the cloud in tgn_sa_all_mlp2_max: no (B,1,N,.) tensor, no torch convolution."""
    B, N, _ = xyz.shape
    D = 0 if points is None else points.shape[2]
    L = lib()
    direct = bool(L.tgn_sa_mlp2_direct_supported(min(N, 64), D))

    def operands():
        f = fold_first_layer(convs[0], bns[0], D, True)
        C1p = (f["C1"] + 15) // 16 * 16
        W2f, b2 = fold_second_layer(convs[1], bns[1], C1p)
        return dict(C1p=C1p, W2f=W2f, b2=b2, b1=_pad_cols(f["b2"], C1p), Wd=_pad_cols(f["Wd"], C1p) if direct else None,
                    Wt=None if direct else _pad_cols(f["Wt"], C1p))
    ops = _derived.cached(bns[1], "all_mlp2", _derived.sources(convs[0], bns[0], convs[1], bns[1]), (D, direct), operands)
    C2 = ops["b2"].shape[0]
    out = torch.empty(B, C2, dtype=torch.float32, device=xyz.device)
    chunks = int(L.tgn_sa_all_chunks(N))
    part = torch.empty(B, chunks, C2, dtype=torch.float32, device=xyz.device) if chunks > 1 else None
    A1 = None if direct else sa_point_transform(xyz, points, ops["Wt"])
    check(L.tgn_sa_all_mlp2_max(B, N, D, ops["C1p"], C2, ptr(A1), ptr(xyz), ptr(points), ptr(ops["Wd"]), ptr(ops["b1"]),
                                ptr(ops["W2f"]), ptr(ops["b2"]), ptr(part), ptr(out), C2, stream()), "sa_all_mlp2_max")
    return out


def _mlp2_shape_ok(K, C1):
    """tgn_sa_mlp2_max keeps the per-query constants of its 4 (K <= 32) or 2 queries in LDS next to the tile buffers."""
    return 1 <= K <= 64 and (4 if K <= 32 else 2) * ((C1 + 15) // 16 * 16) <= 8192


def _fused_level(xyz_c, new_xyz, points_c, idx, convs, bns, xyz_first, N, S, K):
    """The fused eval-mode forms of one (radius, nsample) branch, or None when the shared MLP / shape has none:
    one layer -> sa_level_max; two layers (every level of the reference networks) -> sa_level_mlp2_max; more -> fused first
    layer (where it is cheaper than grouping) + the remaining layers in torch.  Returns (B, C', S)."""
    n = len(convs)
    if n == 0 or not _fused_shape_ok(K, convs[0].out_channels):
        return None
    if n == 1:
        return sa_level_max(xyz_c, new_xyz, points_c, idx, convs[0], bns[0], xyz_first).permute(0, 2, 1)
    if n == 2 and _mlp2_shape_ok(K, convs[0].out_channels):
        return sa_level_mlp2_max(xyz_c, new_xyz, points_c, idx, convs, bns, xyz_first).permute(0, 2, 1)
    if _fuse_pays(N, S, K, 3 + (0 if points_c is None else points_c.shape[2]), convs[0].out_channels):
        y = sa_first_layer(xyz_c, new_xyz, points_c, idx, convs[0], bns[0], xyz_first=xyz_first)
        return _mlp_tail_and_max(y, convs, bns, 1)
    return None


def _fused_shape_ok(K, C1):
    return K <= 64 and C1 % 4 == 0


def _mlp_tail_and_max(x_bskc, convs, bns, first):
    """Layers `first`.. of the shared MLP on a (B,S,K,C) tensor, then max over K -> (B,C',S)."""
    x = x_bskc.permute(0, 3, 2, 1)  # [B, C, K, S] view, exactly what the reference feeds its Conv2d stack
    for j in range(first, len(convs)):
        x = F.relu(bns[j](convs[j](x)))
    return torch.max(x, 2)[0]


# ---------------------------------------------------------------------------------------------
# modules (constructor signatures and parameter names of pointnet2_utils.py:198-352)
# ---------------------------------------------------------------------------------------------
class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super(PointNetSetAbstraction, self).__init__()
        self.npoint = npoint
        self.radius = radius
        self.nsample = nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last_channel = out_channel
        self.group_all = group_all

    def forward(self, xyz, points):
        """channel-first in, channel-first out: xyz (B,3,N), points (B,D,N) or None ->
        new_xyz (B,3,S) sampled centres, features (B,D',S)."""
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        if not self.group_all and len(self.mlp_convs) > 0 and _can_fuse(self, xyz, points):
            # eval fast path: FPS (+coordinates) -> ball query -> fused shared MLP + max; `grouped` is never built
            xyz_c = _f32c(xyz)
            points_c = None if points is None else _f32c(points)
            _, new_xyz = _fps_dense(xyz_c, self.npoint, want_coords=True, prefix=True)
            idx = query_ball_point(self.radius, self.nsample, xyz_c, new_xyz)
            new_points = _fused_level(xyz_c, new_xyz, points_c, idx, self.mlp_convs, self.mlp_bns, True, xyz_c.shape[1],
                                      self.npoint, self.nsample)
            if new_points is None:      # no fused form for this MLP / shape: group, then the reference's stack
                new_points = _mlp_tail_and_max(group_points(xyz_c, new_xyz, points_c, idx, xyz_first=True), self.mlp_convs,
                                               self.mlp_bns, 0)
            return new_xyz.permute(0, 2, 1), new_points
        if (self.group_all and len(self.mlp_convs) == 2 and xyz.shape[1] >= 1 and _can_fuse(self, xyz, points)
                and _mlp2_shape_ok(min(xyz.shape[1], 64), self.mlp_convs[0].out_channels)):
            # eval fast path of the form the reference builds (tsg_seg_module.py:28): the whole cloud is one group
            y = sa_all_mlp2_max(_f32c(xyz), None if points is None else _f32c(points), self.mlp_convs, self.mlp_bns)
            return xyz.new_zeros(xyz.shape[0], xyz.shape[2], 1), y.unsqueeze(2)
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points)
        new_points = new_points.permute(0, 3, 2, 1)  # [B, C+D, nsample, npoint]
        for i, conv in enumerate(self.mlp_convs):
            bn = self.mlp_bns[i]
            new_points = F.relu(bn(conv(new_points)))
        new_points = torch.max(new_points, 2)[0]
        new_xyz = new_xyz.permute(0, 2, 1)
        return new_xyz, new_points


class PointNetSetAbstractionMsg(nn.Module):
    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super(PointNetSetAbstractionMsg, self).__init__()
        self.npoint = npoint
        self.radius_list = radius_list
        self.nsample_list = nsample_list
        self.conv_blocks = nn.ModuleList()
        self.bn_blocks = nn.ModuleList()
        for i in range(len(mlp_list)):
            convs = nn.ModuleList()
            bns = nn.ModuleList()
            last_channel = in_channel + 3
            for out_channel in mlp_list[i]:
                convs.append(nn.Conv2d(last_channel, out_channel, 1))
                bns.append(nn.BatchNorm2d(out_channel))
                last_channel = out_channel
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        """channel-first in, channel-first out: xyz (B,3,N), points (B,D,N) or None ->
        new_xyz (B,3,S) sampled centres, features (B,D',S)."""
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        S = self.npoint
        # one FPS launch yields both the indices and the sampled coordinates (:276)
        if xyz.requires_grad:
            new_xyz = index_points(xyz, farthest_point_sample(xyz, S))
        else:
            _, new_xyz = _fps_dense(xyz, S, want_coords=True, prefix=not self.training)
        xyz_c = _f32c(xyz)
        points_c = None if points is None else _f32c(points)
        fuse = _can_fuse(self, xyz, points)
        new_points_list = []
        chained = fuse and all(len(c) == 2 and _fused_shape_ok(k, c[0].out_channels) and _mlp2_shape_ok(k, c[0].out_channels)
                               for c, k in zip(self.conv_blocks, self.nsample_list))
        with _lib.deferred_index_check("set abstraction (grouping)"):      # one flag read for all branches
            if chained:
                # every branch is one chained kernel: they write their columns of the concatenated output side by side (:296-298)
                widths = [c[1].out_channels for c in self.conv_blocks]
                cat = torch.empty(xyz_c.shape[0], S, sum(widths), dtype=torch.float32, device=xyz_c.device)
                col = 0
                for i, radius in enumerate(self.radius_list):
                    group_idx = query_ball_point(radius, self.nsample_list[i], xyz_c, new_xyz)
                    sa_level_mlp2_max(xyz_c, new_xyz, points_c, group_idx, self.conv_blocks[i], self.bn_blocks[i], False,
                                      out=cat[:, :, col:col + widths[i]])
                    col += widths[i]
                return new_xyz.permute(0, 2, 1), cat.permute(0, 2, 1)
            for i, radius in enumerate(self.radius_list):
                K = self.nsample_list[i]
                group_idx = query_ball_point(radius, K, xyz_c, new_xyz)
                convs, bns = self.conv_blocks[i], self.bn_blocks[i]
                new_points = _fused_level(xyz_c, new_xyz, points_c, group_idx, convs, bns, False, xyz_c.shape[1], S, K) if fuse else None
                if new_points is None:
                    grouped_points = group_points(xyz_c, new_xyz, points_c, group_idx, xyz_first=False)  # [feat, rel_xyz] (:285)
                    new_points = _mlp_tail_and_max(grouped_points, convs, bns, 0)                        # (:286-294)
                new_points_list.append(new_points)
        new_xyz = new_xyz.permute(0, 2, 1)
        new_points_concat = torch.cat(new_points_list, dim=1)
        return new_xyz, new_points_concat


class PointNetFeaturePropagation(nn.Module):
    def __init__(self, in_channel, mlp):
        super(PointNetFeaturePropagation, self).__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel

    def forward(self, xyz1, xyz2, points1, points2):
        """xyz1 (B,3,N) dense level, xyz2 (B,3,S) coarse level, points1 (B,D1,N) skip features or None,
        points2 (B,D2,S) coarse features -> (B,D',N): inverse-distance interpolation of points2 onto xyz1,
        concatenated after points1, then the Conv1d/BN/ReLU stack."""
        xyz1 = xyz1.permute(0, 2, 1)
        xyz2 = xyz2.permute(0, 2, 1)
        points2 = points2.permute(0, 2, 1)
        B, N, C = xyz1.shape
        _, S, _ = xyz2.shape

        if (config.cfg.commute_fp and S > 1 and S < N and len(self.mlp_convs) > 0 and not (xyz1.requires_grad or xyz2.requires_grad)
                and points2.dtype == torch.float32):
            # The first 1x1 convolution commutes with the interpolation (a weighted sum whose weights do not depend on the
            # features): W * [p1, sum_i w_i f2[idx_i]] + b = W1 * p1 + sum_i w_i (W2 * f2)[idx_i] + b.  The coarse features are
            # transformed ONCE per coarse point (S rows instead of N: fp1 of the reference net 1024 instead of 24 000 rows of a
            # 512-wide contraction), and neither the interpolated (B,N,D2) tensor nor the concatenated (B,D1+D2,N) one exists.
            # Exact algebra; fp32 rounding differs in summation order only.  Differentiable like the plain form.
            conv0 = self.mlp_convs[0]
            W = conv0.weight.squeeze(-1)                                        # (C1, D1 + D2), columns [points1, points2] (:345)
            D2 = points2.shape[2]
            D1 = W.shape[1] - D2
            dist, idx = three_nn(xyz1, xyz2)
            if _can_fuse(self, xyz1, xyz2, points1, points2):
                # eval: every BatchNorm folded into its 1x1 convolution (memoised), the whole stack on channel-last rows -- one GEMM
                # with bias + an in-place ReLU per layer, no normalisation kernels and no (B,N,C) <-> (B,C,N) copies in between; the
                # result goes out as the channel-first VIEW of the channel-last tensor
                def fold():
                    out = []
                    for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                        sc = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float()
                        sh = (bn.bias.detach() - bn.running_mean * sc).float()
                        Wc = conv.weight.detach().squeeze(-1).float()
                        bias = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(sh)
                        out.append(((Wc * sc[:, None]).contiguous(), (bias * sc + sh).contiguous()))
                    return out
                srcs = _derived.sources(*self.mlp_convs, *self.mlp_bns)
                layers = _derived.cached(self.mlp_bns[0], "fp_eval", srcs, None, fold)
                W0, b0 = layers[0]
                # relu(interpolation of the transformed coarse rows + the skip features' share + bias): the bias rides in the COARSE
                # GEMM (S contiguous rows; the interpolation weights sum to one, so it comes through unchanged up to one rounding --
                # the skip features arrive as a channel-first view, where torch would add a bias in a pass of its own), sum and ReLU
                # in the interpolation kernel -- no elementwise pass over the (B, N, C1) tensor
                coarse = F.linear(points2, W0[:, D1:], b0)
                skip = F.linear(points1.permute(0, 2, 1), W0[:, :D1]) if points1 is not None else None
                y = three_interpolate_add_relu(coarse, dist, idx, add=skip, relu=True)      # (B, N, C1)
                for Wi, bi in layers[1:]:
                    y = linear_relu(y, Wi, bi)
                return y.permute(0, 2, 1)
            y = three_interpolate(F.linear(points2, W[:, D1:]), dist, idx)      # (B, N, C1)
            if points1 is not None:
                y = y + F.linear(points1.permute(0, 2, 1), W[:, :D1])
            if conv0.bias is not None:
                y = y + conv0.bias
            new_points = F.relu(self.mlp_bns[0](y.permute(0, 2, 1)))
            for i in range(1, len(self.mlp_convs)):
                new_points = F.relu(self.mlp_bns[i](self.mlp_convs[i](new_points)))
            return new_points
        if S == 1:
            interpolated_points = points2.repeat(1, N, 1)
        elif xyz1.requires_grad or xyz2.requires_grad:
            # differentiable coordinates (never the case in the reference models): same maths in torch so
            # that the interpolation weights carry gradient as they would in the reference (:333-340)
            dists = square_distance(xyz1, xyz2)
            dists, idx = dists.sort(dim=-1)
            dists, idx = dists[:, :, :3], idx[:, :, :3]
            dist_recip = 1.0 / (dists + 1e-8)
            norm = torch.sum(dist_recip, dim=2, keepdim=True)
            weight = dist_recip / norm
            interpolated_points = torch.sum(index_points(points2, idx) * weight.view(B, N, 3, 1), dim=2)
        else:
            dist, idx = three_nn(xyz1, xyz2)
            interpolated_points = three_interpolate(points2, dist, idx)

        if points1 is not None:
            points1 = points1.permute(0, 2, 1)
            new_points = torch.cat([points1, interpolated_points], dim=-1)
        else:
            new_points = interpolated_points

        new_points = new_points.permute(0, 2, 1)
        for i, conv in enumerate(self.mlp_convs):
            bn = self.mlp_bns[i]
            new_points = F.relu(bn(conv(new_points)))
        return new_points
