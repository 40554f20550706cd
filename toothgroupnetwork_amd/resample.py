"""Mesh -> fixed-size point set resampling: the FPS entry of ``gen_utils.py`` (reference lines 124-140),
called by ``preprocess_data.py:56`` and the inference pipelines (inference_pipeline_sem.py:28,
inference_pipeline_tgn.py:43,312 -- "#TODO slow processing speed" in the reference).

``fps`` / ``resample_pcd`` keep the reference's numpy-in / numpy-out signatures.  ``fps_batch`` is the
MI355X-shaped variant: many meshes packed into ONE launch (one workgroup per mesh), which is how
``preprocess_data.py``'s serial loop over scans should be driven on a 256-CU GPU.
"""
import threading

import numpy as np
import torch

from . import pointops

_staging = threading.local()


def fps(xyz, npoint, prefix=False):
    """(N,>=3) array -> (npoint,) int32 indices; raises if N <= npoint like gen_utils.py:136-137.
    prefix=True: leave the FPS-of-an-FPS-result certificate behind (pointops.fps_with_coords) -- a network that samples the
    returned points again (the inference pipelines feed them straight to the model) then gets its first level for free."""
    xyz = np.asarray(xyz)
    if xyz.shape[0] <= npoint:
        raise ValueError("new fps error")  # the reference does `raise "new fps error"` (a TypeError in py3)
    dev = torch.device("cuda")
    pts = torch.from_numpy(np.ascontiguousarray(xyz[:, :3], dtype=np.float32)).to(dev)
    offset = torch.tensor([pts.shape[0]], dtype=torch.int32, device=dev)
    new_offset = torch.tensor([int(npoint)], dtype=torch.int32, device=dev)
    idx = pointops.fps_with_coords(pts, offset, new_offset, prefix=True)[0] if prefix else pointops.furthestsampling(pts, offset, new_offset)
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
    offset_np = np.cumsum(counts).astype(np.int32)
    # the scans are packed straight into a page-locked staging buffer this thread keeps (no 80 MB temporary to page in per
    # launch, and the copy to the device runs at the link's rate); the .cpu() below orders the buffer's reuse
    total = int(counts.sum())
    stage = getattr(_staging, "buf", None)
    if stage is None or stage.shape[0] < total:
        stage = _staging.buf = torch.empty((max(total, 1 << 20), 3), dtype=torch.float32, pin_memory=True)
    host = stage[:total].numpy()
    pos = 0
    for x, n in zip(xyz_list, counts):
        np.copyto(host[pos:pos + n], x[:, :3], casting="unsafe")
        pos += int(n)
    pts = stage[:total].to(dev, non_blocking=True)
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
