"""ctypes front-end to oracle/pointops_oracle.c (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Function names follow the reference
operators they check (external_libs/pointops/functions/pointops.py and
external_libs/pointnet2_utils/pointnet2_utils.py of limhoyeon/ToothGroupNetwork).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "pointops_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t)


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def num_threads():
    return int(lib().oracle_num_threads())


def opt_n_threads(n):
    return int(lib().oracle_opt_n_threads(int(n)))


def furthestsampling(xyz, offset, new_offset, mode=0, block_size=0):
    """pointops.furthestsampling (pointops.py:10-27): xyz (n,3), cumulative offsets -> idx (m,) int32.

    mode bit 0 = FMA-chain arithmetic, bit 1 = shared-memory-tree tie order of sampling_cuda_kernel.cu.
    mode 0: canonical (torch-CPU arithmetic of pointnet2_utils.py:103-118, start index 0);
    mode 3: "cuda-compat"; mode 2: the reference source without contraction (== oracle/_ref)."""
    xyz, offset, new_offset = _f32(xyz), _i32(offset), _i32(new_offset)
    b = offset.shape[0]
    m = int(new_offset[-1]) if b else 0
    idx = np.zeros(m, dtype=np.int32)
    lib().oracle_furthestsampling(b, _p(xyz, _f32p), _p(offset, _i32p), _p(new_offset, _i32p),
                                  _p(idx, _i32p), int(mode), int(block_size))
    return idx


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.knnquery (pointops.py:30-45): returns (idx (m,k) int32, dist (m,k) = sqrt(dist2))."""
    xyz = _f32(xyz)
    new_xyz = xyz if new_xyz is None else _f32(new_xyz)
    offset, new_offset = _i32(offset), _i32(new_offset)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    dist2 = np.zeros((m, nsample), dtype=np.float32)
    lib().oracle_knnquery(offset.shape[0], m, int(nsample), _p(xyz, _f32p), _p(new_xyz, _f32p),
                          _p(offset, _i32p), _p(new_offset, _i32p), _p(idx, _i32p), _p(dist2, _f32p))
    return idx, np.sqrt(dist2)


def grouping_forward(inp, idx):
    inp, idx = _f32(inp), _i32(idx)
    m, ns = idx.shape
    c = inp.shape[1]
    out = np.empty((m, ns, c), dtype=np.float32)
    lib().oracle_grouping_forward(m, ns, c, _p(inp, _f32p), _p(idx, _i32p), _p(out, _f32p))
    return out


def grouping_backward(grad_output, idx, n):
    grad_output, idx = _f32(grad_output), _i32(idx)
    m, ns, c = grad_output.shape
    gi = np.zeros((n, c), dtype=np.float32)
    lib().oracle_grouping_backward(m, ns, c, _p(grad_output, _f32p), _p(idx, _i32p), _p(gi, _f32p))
    return gi


def interpolation_forward(inp, idx, weight):
    inp, idx, weight = _f32(inp), _i32(idx), _f32(weight)
    n, k = idx.shape
    c = inp.shape[1]
    out = np.zeros((n, c), dtype=np.float32)
    lib().oracle_interpolation_forward(n, c, k, _p(inp, _f32p), _p(idx, _i32p), _p(weight, _f32p), _p(out, _f32p))
    return out


def interpolation_backward(grad_output, idx, weight, m):
    grad_output, idx, weight = _f32(grad_output), _i32(idx), _f32(weight)
    n, c = grad_output.shape
    k = idx.shape[1]
    gi = np.zeros((m, c), dtype=np.float32)
    lib().oracle_interpolation_backward(n, c, k, _p(grad_output, _f32p), _p(idx, _i32p), _p(weight, _f32p),
                                        _p(gi, _f32p))
    return gi


def subtraction_forward(input1, input2, idx):
    input1, input2, idx = _f32(input1), _f32(input2), _i32(idx)
    n, c = input1.shape
    ns = idx.shape[1]
    out = np.empty((n, ns, c), dtype=np.float32)
    lib().oracle_subtraction_forward(n, ns, c, _p(input1, _f32p), _p(input2, _f32p), _p(idx, _i32p), _p(out, _f32p))
    return out


def subtraction_backward(idx, grad_output, n2=None):
    idx, grad_output = _i32(idx), _f32(grad_output)
    n, ns, c = grad_output.shape
    g1 = np.zeros((n, c), dtype=np.float32)
    g2 = np.zeros((n if n2 is None else n2, c), dtype=np.float32)
    lib().oracle_subtraction_backward(n, ns, c, _p(idx, _i32p), _p(grad_output, _f32p), _p(g1, _f32p), _p(g2, _f32p))
    return g1, g2


def aggregation_forward(inp, position, weight, idx):
    inp, position, weight, idx = _f32(inp), _f32(position), _f32(weight), _i32(idx)
    n, ns, c = position.shape
    w_c = weight.shape[-1]
    out = np.zeros((n, c), dtype=np.float32)
    lib().oracle_aggregation_forward(n, ns, c, w_c, _p(inp, _f32p), _p(position, _f32p), _p(weight, _f32p),
                                     _p(idx, _i32p), _p(out, _f32p))
    return out


def aggregation_backward(inp, position, weight, idx, grad_output):
    inp, position, weight, idx, grad_output = _f32(inp), _f32(position), _f32(weight), _i32(idx), _f32(grad_output)
    n, ns, c = position.shape
    w_c = weight.shape[-1]
    gi = np.zeros_like(inp)
    gp = np.zeros_like(position)
    gw = np.zeros_like(weight)
    lib().oracle_aggregation_backward(n, ns, c, w_c, _p(inp, _f32p), _p(position, _f32p), _p(weight, _f32p),
                                      _p(idx, _i32p), _p(grad_output, _f32p), _p(gi, _f32p), _p(gp, _f32p),
                                      _p(gw, _f32p))
    return gi, gp, gw


def square_distance(src, dst):
    """pointnet2_utils.square_distance (pointnet2_utils.py:20-41): (B,N,3),(B,M,3) -> (B,N,M)."""
    src, dst = _f32(src), _f32(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = np.empty((B, N, M), dtype=np.float32)
    lib().oracle_square_distance(B, N, M, _p(src, _f32p), _p(dst, _f32p), _p(out, _f32p))
    return out


def radius_sq_f32(radius):
    """The threshold `sqrdists > radius ** 2` (pointnet2_utils.py:135) is evaluated with:
    python computes radius**2 in double, torch casts that scalar to the tensor dtype (fp32)."""
    return np.float32(float(radius) ** 2)


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet2_utils.query_ball_point (pointnet2_utils.py:120-144) -> (B,S,nsample) int64."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    out = np.empty((B, S, nsample), dtype=np.int64)
    lib().oracle_ball_query(B, N, S, int(nsample), ctypes.c_float(radius_sq_f32(radius)), _p(xyz, _f32p),
                            _p(new_xyz, _f32p), _p(out, _i64p))
    return out


def three_nn(xyz1, xyz2):
    """three nearest support points of PointNetFeaturePropagation (pointnet2_utils.py:333-335).
    xyz1 (B,N,3) queries, xyz2 (B,S,3) support -> dist (B,N,3) fp32 (expanded form), idx (B,N,3) int64."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    dist = np.empty((B, N, 3), dtype=np.float32)
    idx = np.empty((B, N, 3), dtype=np.int64)
    lib().oracle_three_nn(B, N, S, _p(xyz1, _f32p), _p(xyz2, _f32p), _p(dist, _f32p), _p(idx, _i64p))
    return dist, idx


def three_interpolate(points2, dist, idx):
    """inverse-distance weighted sum (pointnet2_utils.py:337-340): points2 (B,S,C) -> (B,N,C)."""
    points2, dist, idx = _f32(points2), _f32(dist), _i64(idx)
    B, S, C = points2.shape
    N = dist.shape[1]
    out = np.empty((B, N, C), dtype=np.float32)
    lib().oracle_three_interpolate(B, N, S, C, _p(points2, _f32p), _p(dist, _f32p), _p(idx, _i64p), _p(out, _f32p))
    return out


def group_points(xyz, new_xyz, points, idx, xyz_first=True):
    """grouping lines of sample_and_group (pointnet2_utils.py:162-169, xyz_first=True) and of
    PointNetSetAbstractionMsg (pointnet2_utils.py:281-285, xyz_first=False) -> (B,S,K,3+D)."""
    xyz, new_xyz, idx = _f32(xyz), _f32(new_xyz), _i64(idx)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    D = 0 if points is None else points.shape[2]
    pts = None if points is None else _f32(points)
    out = np.empty((B, S, K, 3 + D), dtype=np.float32)
    rc = lib().oracle_group_points(B, N, S, K, D, _p(xyz, _f32p), _p(new_xyz, _f32p),
                                   None if pts is None else _p(pts, _f32p), _p(idx, _i64p), int(bool(xyz_first)),
                                   _p(out, _f32p))
    if rc != 0:
        raise IndexError("group index out of range (empty ball -> index N), as the reference would raise")
    return out


def index_points(points, idx):
    """pointnet2_utils.index_points (pointnet2_utils.py:44-61): points[b, idx[b,...], :]."""
    points = np.asarray(points)
    idx = np.asarray(idx)
    B = points.shape[0]
    bidx = np.arange(B).reshape((B,) + (1,) * (idx.ndim - 1))
    return points[bidx, idx]


def farthest_point_sample(xyz, npoint, mode=0):
    """pointnet2_utils.farthest_point_sample (pointnet2_utils.py:64-98): (B,N,3) -> (B,npoint) int64 local idx."""
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    offset = (np.arange(1, B + 1) * N).astype(np.int32)
    new_offset = (np.arange(1, B + 1) * npoint).astype(np.int32)
    idx = furthestsampling(xyz.reshape(-1, 3), offset, new_offset, mode=mode)
    return idx.reshape(B, npoint).astype(np.int64) - (np.arange(B, dtype=np.int64) * N).reshape(-1, 1)


def sample_and_group(npoint, radius, nsample, xyz, points, xyz_first=True):
    """sample_and_group (pointnet2_utils.py:147-175): returns new_xyz, new_points, fps_idx, group_idx."""
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(_f32(xyz), fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    new_points = group_points(xyz, new_xyz, points, idx, xyz_first=xyz_first)
    return new_xyz, new_points, fps_idx, idx


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """pointops.interpolation (pointops.py:164-180): kNN(k) inverse-distance (sqrt distances) weighting."""
    idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
    dist_recip = (np.float32(1.0) / (dist + np.float32(1e-8))).astype(np.float32)
    norm = dist_recip[:, 0:1].copy()
    for i in range(1, k):
        norm = norm + dist_recip[:, i:i + 1]
    weight = (dist_recip / norm).astype(np.float32)
    feat = _f32(feat)
    out = np.zeros((idx.shape[0], feat.shape[1]), dtype=np.float32)
    for i in range(k):
        out += feat[idx[:, i].astype(np.int64), :] * weight[:, i:i + 1]
    return out, idx, weight


def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True):
    """pointops.queryandgroup (pointops.py:79-100) -> (m, nsample, 3+c) or (m, nsample, c)."""
    xyz = _f32(xyz)
    new_xyz = xyz if new_xyz is None else _f32(new_xyz)
    feat = _f32(feat)
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    li = np.asarray(idx).astype(np.int64)
    grouped_xyz = xyz[li.reshape(-1)].reshape(li.shape[0], nsample, 3) - new_xyz[:, None, :]
    grouped_feat = feat[li.reshape(-1)].reshape(li.shape[0], nsample, feat.shape[1])
    if use_xyz:
        return np.concatenate([grouped_xyz, grouped_feat], axis=-1)
    return grouped_feat


def set_abstraction_first_layer(xyz, new_xyz, points, idx, W, bias, gamma, beta, mean, var, eps=1e-5, xyz_first=True,
                                reduce_max=False):
    """First shared-MLP layer of a set-abstraction level in eval mode, restated from the reference modules:
    grouping (pointnet2_utils.py:162-169 [xyz first] / 281-285 [features first]) -> Conv2d 1x1 (W (C1, 3+D), bias) ->
    BatchNorm2d with running statistics -> ReLU (:229-233 / :289-293) -> optionally max over the K neighbours
    (:236 / :294).  The grouping is the exact fp32 one; the contraction is accumulated in float64 (the value every fp32
    summation order approximates)."""
    grouped = group_points(xyz, new_xyz, points, idx, xyz_first).astype(np.float64)      # (B,S,K,3+D)
    y = grouped @ np.asarray(W, dtype=np.float64).T + np.asarray(bias, dtype=np.float64)
    y = (y - np.asarray(mean, np.float64)) / np.sqrt(np.asarray(var, np.float64) + eps) * np.asarray(gamma, np.float64) \
        + np.asarray(beta, np.float64)
    y = np.maximum(y, 0.0)
    return y.max(axis=2) if reduce_max else y


def set_abstraction_mlp(xyz, new_xyz, points, idx, layers, eps=1e-5, xyz_first=True):
    """A WHOLE set-abstraction level after sampling and ball query, eval mode, restated from the reference modules:
    grouping (pointnet2_utils.py:162-169 / 281-285) -> [Conv2d 1x1 -> BatchNorm2d (running statistics) -> ReLU] for every
    layer of the shared MLP (:229-233 / :286-293) -> max over the K neighbours (:236 / :294).  layers: list of
    (W (C_out, C_in), bias, gamma, beta, mean, var).  float64 accumulation; returns (B,S,C_last)."""
    y = group_points(xyz, new_xyz, points, idx, xyz_first).astype(np.float64)            # (B,S,K,3+D)
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    for W, bias, gamma, beta, mean, var in layers:
        y = y @ f8(W).T + f8(bias)
        y = (y - f8(mean)) / np.sqrt(f8(var) + eps) * f8(gamma) + f8(beta)
        y = np.maximum(y, 0.0)
    return y.max(axis=2)


def pt_attention_layer(p, x_q, x_k, x_v, idx, sd, share_planes=8, eps=1e-5):
    """PointTransformerLayer.forward after the three input projections, eval mode, restated from
    models/modules/cbl_point_transformer/blocks.py:34-43 in float64.  p (n,3), x_q/x_k/x_v (n,c), idx (n,nsample) neighbour
    rows, sd: the layer's state_dict as numpy arrays (reference names: linear_p.0/1/3, linear_w.0/2/3/5)."""
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    p, x_q, x_k, x_v = f8(p), f8(x_q), f8(x_k), f8(x_v)
    li = np.asarray(idx).astype(np.int64)

    def bn(x, name):          # BatchNorm1d over the channel (last) axis, running statistics
        return (x - f8(sd[name + ".running_mean"])) / np.sqrt(f8(sd[name + ".running_var"]) + eps) * f8(sd[name + ".weight"]) \
            + f8(sd[name + ".bias"])

    def lin(x, name):
        return x @ f8(sd[name + ".weight"]).T + f8(sd[name + ".bias"])
    p_r = p[li] - p[:, None, :]                                              # queryandgroup: xyz[idx] - new_xyz (pointops.py:89-91)
    p_r = lin(np.maximum(bn(lin(p_r, "linear_p.0"), "linear_p.1"), 0.0), "linear_p.3")      # (n, nsample, c)
    w = x_k[li] - x_q[:, None, :] + p_r                                      # blocks.py:39 (mid_planes == out_planes)
    w = lin(np.maximum(bn(w, "linear_w.0"), 0.0), "linear_w.2")
    w = lin(np.maximum(bn(w, "linear_w.3"), 0.0), "linear_w.5")              # (n, nsample, c / share_planes)
    w = np.exp(w - w.max(axis=1, keepdims=True))
    w = w / w.sum(axis=1, keepdims=True)                                     # softmax over the neighbours (:41)
    n, ns, c = p_r.shape
    s = share_planes
    return ((x_v[li] + p_r).reshape(n, ns, s, c // s) * w[:, :, None, :]).sum(1).reshape(n, c)
