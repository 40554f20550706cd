#!/usr/bin/env python3
"""Round-3 golden fixtures (tests/golden/reference_cpu_r3.npz): WHOLE reference networks and the label transfer, produced by
running the REFERENCE's own Python on CPU in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py

  (a) models/modules/pointnet_pp.py:6-70 `get_model()` in eval mode (BASELINE config 2's network, the widths the reference
      instantiates), forward on two 3000-point scans.  Only `farthest_point_sample` (CUDA-only, pointnet2_utils.py:87-98) is
      served -- by the reference's own CPU FPS `farthest_point_sample_np` with the start forced to 0, as in make_golden.py.
  (b) models/modules/cbl_point_transformer/cbl_point_transformer_module.py:93-216 `PointTransformerSeg.forward` in eval mode
      (tgnet_fps stage sizes: BASELINE configs 3 / 4's network), one 3000-point scan (class + offset heads) and a batch of
      two 1400-point scans.  Only the two CUDA kernels it reaches are served by the oracle: `pointops.furthestsampling` and
      `pointops.knnquery`; `queryandgroup` and `interpolation` are the reference's own torch code on top of them
      (pointops.py:79-100,164-180); `.cuda()` / `torch.cuda.*Tensor` are pointed at the CPU types for the duration.
  (c) inference_pipeline_sem.py:37-39: `KDTree(sampled[:, :3], leaf_size=2).query(vertices, k=1)` label transfer (sklearn,
      float64), with the gap between the nearest and the second-nearest neighbour so that the test can tell unique answers
      from float-level ties.

Each network runs twice: in float32 (what the reference computes) and in float64 with every index-producing step kept on
the float32 values (same FPS / ball-query / kNN / three-NN indices), which gives the exact value of the same network
function and therefore the rounding noise of the reference's own fp32 arithmetic.  The tests compare the drop-in with the
float64 result elementwise and report its distance next to the reference-fp32 one's.

Weights: tests/golden/seeded.py draws every tensor from a generator seeded by its name, so the GPU tests rebuild the same
35 MB / 31 MB of parameters from the name list stored here instead of a weight file."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("TGN_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_golden import load_reference, ref_fps  # noqa: E402
from oracle import cpu as O  # noqa: E402
from seeded import seeded_fill  # noqa: E402
from toothgroupnetwork_amd import synth  # noqa: E402


class cpu_as_cuda:
    """`.cuda()` is the identity and the legacy `torch.cuda.*Tensor` constructors build CPU tensors (of `ftype` for
    FloatTensor, so that the float64 pass is not truncated by pointops.py:176)."""

    def __init__(self, ftype=torch.FloatTensor):
        self.ftype = ftype

    def __enter__(self):
        self.keep = (torch.Tensor.cuda, torch.cuda.IntTensor, torch.cuda.FloatTensor)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.cuda.IntTensor, torch.cuda.FloatTensor = torch.IntTensor, self.ftype

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.cuda.IntTensor, torch.cuda.FloatTensor = self.keep


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


def pointnet_pp_net(R, out):
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.pointnet_pp as M
    assert M.PointNetSetAbstractionMsg.__module__ == "external_libs.pointnet2_utils.pointnet2_utils"
    U = sys.modules["external_libs.pointnet2_utils.pointnet2_utils"]
    assert U.__file__.startswith(REFERENCE), U.__file__
    B, N = 2, 3000
    scans = synth.scan_batch(B, N, "arch", seed=301)                       # (B, N, 6)
    feats = torch.from_numpy(np.ascontiguousarray(scans.transpose(0, 2, 1)))
    net = M.get_model().eval()
    out["pnpp_params"] = np.array(seeded_fill(net, 31))
    keep_fps, keep_sqd = U.farthest_point_sample, U.square_distance
    U.farthest_point_sample = lambda x, n: torch.from_numpy(ref_fps(U, x.detach().float().numpy(), n))
    try:
        with torch.no_grad():
            y32 = net([feats])
            # float64 pass: distances (hence ball-query / three-NN indices) stay the fp32 ones
            U.square_distance = lambda a, b: keep_sqd(a.float(), b.float()).double()
            y64 = net.double()([feats.double()])
    finally:
        U.farthest_point_sample, U.square_distance = keep_fps, keep_sqd
    names = ["l0_points", "l3_points", "l0_xyz", "l3_xyz", "offset", "dist", "cls"]
    for n_, a, b in zip(names, y32, y64):
        print(f"  pointnet_pp.get_model {n_:10s} {tuple(a.shape)}  |fp32 - fp64| / (1 + |fp64|) = {rel_err(a.numpy(), b.numpy()):.2e}")
    out["pnpp_scans"] = scans
    sub = {"l0_points": (slice(None), slice(0, None, 8)), "l3_points": (slice(None), slice(0, None, 8))}
    for n_, a, b in zip(names, y32, y64):
        if n_ == "l0_xyz":
            continue
        s = sub.get(n_, (slice(None),))
        out[f"pnpp_{n_}_64"] = b.numpy()[s].astype(np.float32)
        if n_ in ("cls", "offset", "l3_xyz"):
            out[f"pnpp_{n_}_32"] = a.numpy()[s]


def point_transformer_net(out):
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.cbl_point_transformer.cbl_point_transformer_module as M
    import models.modules.cbl_point_transformer.blocks as RB
    RP = RB.pointops
    assert RP.__file__.startswith(REFERENCE), RP.__file__

    def fps(xyz, offset, new_offset):
        return torch.from_numpy(O.furthestsampling(xyz.float().numpy(), offset.numpy(), new_offset.numpy()).astype(np.int32))

    def knn(nsample, xyz, new_xyz, offset, new_offset):
        idx, dist = O.knnquery(int(nsample), xyz.float().numpy(), new_xyz.float().numpy(), offset.numpy(), new_offset.numpy())
        return torch.from_numpy(idx), torch.from_numpy(dist).to(xyz.dtype)

    cfg = dict(c=6, k=17, planes=[32, 64, 128, 256, 512], stride=[1, 4, 4, 4, 4], nsample=[36, 24, 24, 24, 24],
               blocks=[2, 3, 4, 6, 3], block_num=5)
    net = M.get_model(**cfg).eval()
    out["pt_params"] = np.array(seeded_fill(net, 32))
    keep = RP.furthestsampling, RP.knnquery
    RP.furthestsampling, RP.knnquery = fps, knn
    try:
        for tag, B, N, seed in (("one", 1, 3000, 302), ("two", 2, 1400, 303)):
            scans = synth.scan_batch(B, N, "arch", seed=seed)
            feats = torch.from_numpy(np.ascontiguousarray(scans.transpose(0, 2, 1)))
            with torch.no_grad():
                with cpu_as_cuda(torch.FloatTensor):
                    y32 = net.float()([feats])
                with cpu_as_cuda(torch.DoubleTensor):
                    y64 = net.double()([feats.double()])
            out[f"pt_{tag}_scans"] = scans
            for n_, i in (("cls", 0), ("offset", 1), ("x1", 3)):
                if y32[i] is None:
                    assert B > 1 and n_ == "offset"
                    continue
                print(f"  PointTransformerSeg[{tag}] {n_:7s} {tuple(y32[i].shape)}  |fp32 - fp64| / (1 + |fp64|) = "
                      f"{rel_err(y32[i].numpy(), y64[i].numpy()):.2e}")
                out[f"pt_{tag}_{n_}_64"] = y64[i].numpy().astype(np.float32)
                if n_ != "x1" or tag == "one":
                    out[f"pt_{tag}_{n_}_32"] = y32[i].numpy()
    finally:
        RP.furthestsampling, RP.knnquery = keep


def label_transfer(out):
    from sklearn.neighbors import KDTree
    full = synth.arch_cloud(40000, seed=304, with_normals=False).astype(np.float64)
    full[5000:5200] = full[100:300]                                            # duplicated vertices, as raw scans have
    samp_idx = O.furthestsampling(full.astype(np.float32), [full.shape[0]], [6000]).astype(np.int64)
    sampled = full[samp_idx]
    labels = (np.arange(6000) * 7 % 17).astype(np.int64)
    tree = KDTree(sampled[:, :3], leaf_size=2)                                 # inference_pipeline_sem.py:37
    dist, near = tree.query(full[:, :3], k=2, return_distance=True)            # :38 asks for k=1; the 2nd gives the gap
    near1 = tree.query(full[:, :3], k=1, return_distance=False)
    assert np.array_equal(dist[:, 0] == dist[:, 1], dist[:, 0] == dist[:, 1])
    same = near1.reshape(-1) == near[:, 0]
    assert same.all() or (dist[~same, 0] == dist[~same, 1]).all()
    out["lt_full"], out["lt_sampled_idx"], out["lt_labels"] = full, samp_idx, labels
    out["lt_near"] = near1.reshape(-1).astype(np.int64)
    out["lt_gap"] = (dist[:, 1] - dist[:, 0]).astype(np.float64)
    print(f"  KDTree label transfer: {full.shape[0]} vertices -> {sampled.shape[0]} samples, "
          f"{int((out['lt_gap'] == 0).sum())} exact ties, {int((out['lt_gap'] < 1e-6).sum())} gaps below 1e-6")


def main():
    torch.set_num_threads(8)
    out = {}
    R = load_reference()
    pointnet_pp_net(R, out)
    point_transformer_net(out)
    label_transfer(out)
    path = os.path.join(HERE, "reference_cpu_r3.npz")
    np.savez_compressed(path, **out)
    print(f"wrote tests/golden/reference_cpu_r3.npz ({os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
