"""Point-Transformer fixture of make_golden_r2.py: the reference's own PointTransformerLayer
(models/modules/cbl_point_transformer/blocks.py:14-44) in eval mode on CPU.  Its two pointops.queryandgroup calls need
the CUDA kNN, so the module-level `pointops.queryandgroup` the layer calls is pointed at the CPU oracle's
restatement (exact gathers / subtraction on the oracle's kNN indices); every learned layer, the softmax and the
share_planes aggregation are the reference's torch code.  The oracle's float64 restatement is checked against it."""
import os
import sys

import numpy as np
import torch

from make_golden import check
from oracle import cpu as O
from toothgroupnetwork_amd import synth

REFERENCE = os.environ.get("TGN_REFERENCE", "/root/reference")


def pt_fixtures(out):
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.cbl_point_transformer.blocks as RB
    torch.manual_seed(77)
    n_per, c, ns = 700, 32, 16
    xyz = np.concatenate([synth.arch_cloud(n_per, seed=s, with_normals=False) for s in (50, 51)]).astype(np.float32)
    off = np.array([n_per, 2 * n_per], np.int32)
    x = np.random.default_rng(9).normal(size=(2 * n_per, c)).astype(np.float32)

    def qg(nsample, p, new_p, feat, idx, o, n_o, use_xyz=True):
        r = O.queryandgroup(nsample, p.numpy(), new_p.numpy(), feat.detach().numpy(), None, o.numpy(), n_o.numpy(), use_xyz)
        return torch.from_numpy(r)
    RB.pointops.queryandgroup, keep = qg, RB.pointops.queryandgroup
    try:
        layer = RB.PointTransformerLayer(c, c, 8, ns).eval()
        for m_ in layer.modules():
            if isinstance(m_, torch.nn.BatchNorm1d):
                m_.running_mean.normal_(0, 0.2)
                m_.running_var.uniform_(0.5, 2.0)
                m_.weight.data.uniform_(0.5, 1.5)
                m_.bias.data.normal_(0, 0.1)
        with torch.no_grad():
            y = layer([torch.from_numpy(xyz), torch.from_numpy(x), torch.from_numpy(off)]).numpy()
    finally:
        RB.pointops.queryandgroup = keep
    sd = {k: v.numpy() for k, v in layer.state_dict().items()}
    with torch.no_grad():
        t = torch.from_numpy(x)
        xq, xk, xv = layer.linear_q(t).numpy(), layer.linear_k(t).numpy(), layer.linear_v(t).numpy()
    idx, _ = O.knnquery(ns, xyz, xyz, off, off)
    ora = O.pt_attention_layer(xyz, xq, xk, xv, idx, sd, 8)
    check("pt_attention_layer", ora, y, exact=False, tol=2e-5)
    out["pt_xyz"], out["pt_off"], out["pt_x"], out["pt_y"] = xyz, off, x, y
    torch.save(layer.state_dict(), os.path.join(os.path.dirname(os.path.abspath(__file__)), "pt_layer_weights_r2.pt"))
