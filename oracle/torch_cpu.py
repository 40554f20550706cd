"""The reference's CPU path of the headline workload, torch operation for torch operation.

TEST / BENCHMARK INFRASTRUCTURE ONLY -- see oracle/__init__.py: nothing under toothgroupnetwork_amd/ may import this.

The reference checkout does not exist on the GPU box, so its own functions cannot be timed there.
What decides their run time is the sequence of torch calls they make on CPU tensors -- a python loop of masked updates for the
sampling, an N-wide sort per query for the ball query, advanced indexing for the grouping -- and that sequence is restated
here, call for call, from external_libs/pointnet2_utils/pointnet2_utils.py:

    square_distance      :20-41    -2 * matmul, then the two squared norms added in place
    index_points         :44-61    advanced indexing with a broadcast batch index
    fps                  :103-118  `farthest_point_sample_np`, its random start (:109) replaced by index 0 -- the start of the
                                   CUDA kernel (sampling_cuda_kernel.cu:39) and of every golden fixture
    ball_query           :120-144  arange -> mask to N -> sort over N -> first nsample -> pad with the first hit
    group                :160-169  the gather / centre / cat lines of `sample_and_group`

tests/test_oracle_golden.py pins every function to the fixtures the reference's OWN functions produced in the build
container (tests/golden/reference_cpu.npz): equal indices, equal bits.  bench.py times `headline_levels` on the bench host
as `cpu_baseline` (kind "port": same torch kernels, same shapes, same thread pool as the reference would use there).
"""
import time

import numpy as np
import torch


def square_distance(src, dst):
    b, n, _ = src.shape
    m = dst.shape[1]
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d += torch.sum(src ** 2, -1).view(b, n, 1)
    d += torch.sum(dst ** 2, -1).view(b, 1, m)
    return d


def index_points(points, idx):
    b = points.shape[0]
    lead = [b] + [1] * (idx.dim() - 1)
    tile = [1] + list(idx.shape[1:])
    batch = torch.arange(b, dtype=torch.long).view(lead).repeat(tile)
    return points[batch, idx, :]


def fps(xyz, npoint, start=0):
    """xyz (B, N, 3) CPU tensor -> (B, npoint) int64; every sample costs a full pass of masked updates, as in the reference"""
    b, n, _ = xyz.shape
    picked = torch.zeros(b, npoint, dtype=torch.long)
    nearest = torch.ones(b, n) * 1e10
    far = torch.full((b,), int(start), dtype=torch.long)
    rows = torch.arange(b, dtype=torch.long)
    for i in range(npoint):
        picked[:, i] = far
        c = xyz[rows, far, :].view(b, 1, 3)
        d = torch.sum((xyz - c) ** 2, -1)
        closer = d < nearest
        nearest[closer] = d[closer]
        far = torch.max(nearest, -1)[1]
    return picked


def ball_query(radius, nsample, xyz, new_xyz):
    b, n, _ = xyz.shape
    s = new_xyz.shape[1]
    idx = torch.arange(n, dtype=torch.long).view(1, 1, n).repeat([b, s, 1])
    d = square_distance(new_xyz, xyz)
    idx[d > radius ** 2] = n
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, 0].view(b, s, 1).repeat([1, 1, nsample])
    empty = idx == n
    idx[empty] = first[empty]
    return idx


def group(xyz, new_xyz, points, idx):
    b, s, _ = new_xyz.shape
    rel = index_points(xyz, idx) - new_xyz.view(b, s, 1, 3)
    if points is None:
        return rel
    return torch.cat([rel, index_points(points, idx)], dim=-1)


def headline_levels(scan, npoint, radius, nsample, d, seed=0):
    """One scan (N, 6) through the levels of the headline workload the way the reference's batch-1 loop would run them on CPU:
    FPS -> ball query -> group per level; level l > 0 groups synthetic features of width d[l] (as bench.py's GPU path does).
    Returns per-level (fps, ball, group) seconds."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.from_numpy(np.ascontiguousarray(scan[None, :, :3]))
    pts = torch.from_numpy(np.ascontiguousarray(scan[None]))
    out = []
    for li, (s, r, k) in enumerate(zip(npoint, radius, nsample)):
        t0 = time.perf_counter()
        new_xyz = index_points(xyz, fps(xyz, s))
        t1 = time.perf_counter()
        idx = ball_query(r, k, xyz, new_xyz)
        t2 = time.perf_counter()
        grouped = group(xyz, new_xyz, pts, idx)
        t3 = time.perf_counter()
        assert tuple(grouped.shape) == (1, s, k, 3 + d[li])
        out.append((t1 - t0, t2 - t1, t3 - t2))
        xyz = new_xyz.contiguous()
        pts = torch.randn(1, s, d[li + 1], generator=g) if li + 1 < len(d) else None
    return out
