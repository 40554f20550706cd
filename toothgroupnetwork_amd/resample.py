"""Mesh -> fixed-size point set resampling: the FPS entry of ``gen_utils.py`` (reference lines 124-140),
called by ``preprocess_data.py:56`` and the inference pipelines (inference_pipeline_sem.py:28,
inference_pipeline_tgn.py:43,312 -- "#TODO slow processing speed" in the reference).

``fps`` / ``resample_pcd`` keep the reference's numpy-in / numpy-out signatures.  ``fps_batch`` is the
MI355X-shaped variant: many meshes packed into ONE launch (one workgroup per mesh), which is how
``preprocess_data.py``'s serial loop over scans should be driven on a 256-CU GPU.
"""
import numpy as np
import torch

from . import pointops


def fps(xyz, npoint):
    """(N,>=3) array -> (npoint,) int32 indices; raises if N <= npoint like gen_utils.py:136-137."""
    xyz = np.asarray(xyz)
    if xyz.shape[0] <= npoint:
        raise ValueError("new fps error")  # the reference does `raise "new fps error"` (a TypeError in py3)
    dev = torch.device("cuda")
    pts = torch.from_numpy(np.ascontiguousarray(xyz[:, :3], dtype=np.float32)).to(dev)
    offset = torch.tensor([pts.shape[0]], dtype=torch.int32, device=dev)
    new_offset = torch.tensor([int(npoint)], dtype=torch.int32, device=dev)
    idx = pointops.furthestsampling(pts, offset, new_offset)
    return idx.cpu().numpy().reshape(-1)


def fps_batch(xyz_list, npoint):
    """FPS of several meshes in one launch. xyz_list: list of (N_i,>=3) arrays -> list of (npoint,) int32
    LOCAL indices.  Equivalent to [fps(x, npoint) for x in xyz_list]."""
    if len(xyz_list) == 0:
        return []
    for x in xyz_list:
        if x.shape[0] <= npoint:
            raise ValueError("new fps error")
    dev = torch.device("cuda")
    counts = np.array([x.shape[0] for x in xyz_list], dtype=np.int64)
    packed = np.concatenate([np.ascontiguousarray(x[:, :3], dtype=np.float32) for x in xyz_list], axis=0)
    offset_np = np.cumsum(counts).astype(np.int32)
    pts = torch.from_numpy(packed).to(dev)
    offset = torch.from_numpy(offset_np).to(dev)
    new_offset = torch.arange(1, len(xyz_list) + 1, dtype=torch.int32, device=dev) * int(npoint)
    idx = pointops.furthestsampling(pts, offset, new_offset).cpu().numpy().reshape(len(xyz_list), npoint)
    starts = np.concatenate([[0], offset_np[:-1]]).astype(np.int32)
    return [idx[i] - starts[i] for i in range(len(xyz_list))]


def resample_pcd(pcd_ls, n, method):
    """Drop or duplicate points so that pcd has exactly n points (gen_utils.py:124-133)."""
    if method == "uniformly":
        idx = np.random.permutation(pcd_ls[0].shape[0])
    elif method == "fps":
        idx = fps(pcd_ls[0][:, :3], n)
    else:
        raise ValueError(f"unknown resample method {method!r}")
    return [p[idx[:n]] for p in pcd_ls]
