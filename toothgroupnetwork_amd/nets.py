"""Whole-network compositions of the drop-in modules, state_dict-compatible with the reference's networks.

The reference's own model files import this package's operators unchanged; these mirrors exist for the
places where the reference checkout is not available -- the GPU box's parity tests and benches -- and as the entry points
of the fused eval paths.  Parameter names and shapes equal the reference classes', so trained weights load into them
(tests/test_host_logic.py compares the key lists with the reference's own classes in the build container):

  PointNetPPSeg        models/modules/pointnet_pp.py:6-70 (`get_model`): three multi-scale set-abstraction levels, three
                       feature-propagation levels, offset / distance / class heads; BASELINE.json config 2's network.
  TsgCentroidNet       models/modules/tsg_centroid_module.py:5-46 and
  TsgSegNet            models/modules/tsg_seg_module.py:4-80 -- the two modules of tsegnet (models/modules/tsegnet.py:15-16, the network behind
                       models/tsegnet_model.py): PointNet++-MSG trunks; the second one ends in the ONLY PointNetSetAbstraction(group_all=True)
                       of the reference (`flatten_sa`, :28).
  PointTransformerSeg  models/modules/cbl_point_transformer/cbl_point_transformer_module.py:28-216 with the configuration
                       the reference ships (default.yaml: five stages, `multi` heads over the decoder stages, latent
                       features concatenated); BASELINE.json configs 3 / 4's network.  Inference outputs only: the
                       contrastive-boundary criterion of training (heads.py:62-253) is the reference's own Python.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import functools

from . import _derived, _lib, point_transformer as PT, pointops
from .pointnet2_utils import PointNetFeaturePropagation, PointNetSetAbstraction, PointNetSetAbstractionMsg, linear_relu


def _one_index_check(forward):
    """The gather kernels latch out-of-range indices in a device word that the checked operators read back after their launch -- a
    stream synchronisation each (the price of raising IndexError like torch's advanced indexing, pointnet2_utils.py:56-60).  A whole
    network forward issues six to twelve of them, and at the small levels the GPU then idles behind the host (~0.4 ms of a 4.9 ms
    PointNet++ forward, profiles/r05_pnpp_forward_sequence.txt): the network mirrors read the word ONCE, when the forward is
    through.  The IndexError is the same; it is raised at the end of the forward instead of inside the offending level."""
    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):
        with _lib.deferred_index_check(type(self).__name__ + " forward"):
            return forward(self, *args, **kwargs)
    return wrapped


class PointNetPPSeg(nn.Module):
    def __init__(self, input_feature_num=6, scale=4, cls_pred=True):
        super().__init__()
        s, self.cls_pred = scale, cls_pred
        self.sa1 = PointNetSetAbstractionMsg(1024, [0.025, 0.05], [32, 64], input_feature_num, [[32 * s, 32 * s]] * 2)
        self.sa2 = PointNetSetAbstractionMsg(512, [0.05, 0.1], [32, 64], 64 * s, [[64 * s, 128 * s]] * 2)
        self.sa3 = PointNetSetAbstractionMsg(256, [0.1, 0.2], [32, 64], 256 * s, [[196 * s, 256 * s]] * 2)
        self.fp3 = PointNetFeaturePropagation(768 * s, [256 * s, 256 * s])
        self.fp2 = PointNetFeaturePropagation(320 * s, [128 * s, 128 * s])
        self.fp1 = PointNetFeaturePropagation(128 * s + input_feature_num, [64 * s, 32 * s])
        for head, width in (("offset", 3), ("dist", 1)):
            setattr(self, f"{head}_conv_1", nn.Conv1d(32 * s, 16, 1))
            setattr(self, f"{head}_bn_1", nn.BatchNorm1d(16))
        self.offset_conv_2, self.dist_conv_2 = nn.Conv1d(16, 3, 1), nn.Conv1d(16, 1, 1)
        if cls_pred:
            self.cls_conv_1, self.cls_bn_1, self.cls_conv_2 = nn.Conv1d(32 * s, 17, 1), nn.BatchNorm1d(17), nn.Conv1d(17, 17, 1)
        nn.init.zeros_(self.offset_conv_2.weight)
        nn.init.zeros_(self.dist_conv_2.weight)
        self.conv1, self.bn1 = nn.Conv1d(32, 16, 1), nn.BatchNorm1d(16)     # declared and unused in the reference too (:38-40)

    def _head(self, name, x):
        conv1, bn1, conv2 = getattr(self, f"{name}_conv_1"), getattr(self, f"{name}_bn_1"), getattr(self, f"{name}_conv_2")
        if PT._frozen(self, x) and x.dtype == torch.float32:
            # eval: on channel-last rows (x arrives as the channel-first view of a channel-last tensor, so the permute is free), the
            # BatchNorm folded into the first 1x1 convolution (memoised): two GEMMs with bias, no layout copies of the (B, C, N) features
            def fold():
                s, t = PT._bn_scale_shift(bn1)
                W1 = (conv1.weight.detach().squeeze(-1).float() * s[:, None]).contiguous()
                b1 = (conv1.bias.detach().float() * s + t).contiguous()
                return W1, b1, conv2.weight.detach().squeeze(-1).float().contiguous(), conv2.bias.detach().float().contiguous()
            W1, b1, W2, b2 = _derived.cached(bn1, "head_eval", _derived.sources(conv1, bn1, conv2), None, fold)
            y = linear_relu(x.permute(0, 2, 1), W1, b1)       # (ReLU in the GEMM's epilogue where the rows are contiguous)
            return F.linear(y, W2, b2).permute(0, 2, 1)
        return conv2(F.relu(bn1(conv1(x))))

    @_one_index_check
    def forward(self, xyz_in):
        """xyz_in: [features (B, C, N)] with xyz in the first three channels -> the reference's output list
        [l0_points, l3_points, l0_xyz, l3_xyz, offset, dist(, cls)] (pointnet_pp.py:60-68)."""
        feats = xyz_in[0]
        xyz = [feats[:, :3, :]]
        pts = [feats]
        for sa in (self.sa1, self.sa2, self.sa3):
            x, p = sa(xyz[-1], pts[-1])
            xyz.append(x)
            pts.append(p)
        up = pts[3]
        for lvl, fp in ((2, self.fp3), (1, self.fp2), (0, self.fp1)):
            up = fp(xyz[lvl], xyz[lvl + 1], pts[lvl], up)
        out = [up, pts[3], xyz[0], xyz[3], self._head("offset", up), self._head("dist", up)]
        if self.cls_pred:
            out.append(self._head("cls", up))
        return out


def _msg_sa(mod, suffix, c_in):
    """The three multi-scale set-abstraction levels both tsegnet modules share (tsg_centroid_module.py:10-12, tsg_seg_module.py:11-13 /
    :24-26), registered on `mod` as sa{1,2,3}{suffix}."""
    levels = ((1024, [0.025, 0.05], c_in, [32, 32]), (512, [0.05, 0.1], 64, [64, 128]), (256, [0.1, 0.2], 256, [196, 256]))
    for i, (S, radii, c, widths) in enumerate(levels, 1):
        setattr(mod, f"sa{i}{suffix}", PointNetSetAbstractionMsg(S, radii, [32, 64], c, [widths, widths]))


def _msg_fp(mod, suffix, c_in):
    """... and their three feature-propagation levels fp{3,2,1}{suffix} (:15-17 / :16-18 / :30-32)."""
    for i, (c, widths) in ((3, (768, [256, 256])), (2, (320, [128, 128])), (1, (128 + c_in, [64, 32]))):
        setattr(mod, f"fp{i}{suffix}", PointNetFeaturePropagation(c, widths))


def _run_trunk(mod, suffix, feats):
    """-> (per-point features (B, 32, N), xyz / features of the three coarse levels)"""
    xyz, pts = [feats[:, :3, :]], [feats]
    for i in (1, 2, 3):
        x, p = getattr(mod, f"sa{i}{suffix}")(xyz[-1], pts[-1])
        xyz.append(x)
        pts.append(p)
    up = pts[3]
    for lvl in (3, 2, 1):
        up = getattr(mod, f"fp{lvl}{suffix}")(xyz[lvl - 1], xyz[lvl], pts[lvl - 1], up)
    return up, xyz, pts


class TsgCentroidNet(nn.Module):
    """tsg_centroid_module.get_model: per coarse point (256 of them) an offset to the nearest tooth centroid and its distance."""

    def __init__(self):
        super().__init__()
        _msg_sa(self, "", 6)
        _msg_fp(self, "", 6)
        for head, width in (("offset", 3), ("dist", 1)):
            setattr(self, f"{head}_conv_1", nn.Conv1d(515, 256, 1))
            setattr(self, f"{head}_bn_1", nn.BatchNorm1d(256))
        self.offset_conv_2, self.dist_conv_2 = nn.Conv1d(256, 3, 1), nn.Conv1d(256, 1, 1)
        nn.init.zeros_(self.offset_conv_2.weight)
        nn.init.zeros_(self.dist_conv_2.weight)

    @_one_index_check
    def forward(self, feats):
        """feats (B, 6, N), xyz first -> [l0_points, l3_points, l0_xyz, l3_xyz, offset (B,3,256), dist (B,1,256)] (:29-46)"""
        up, xyz, pts = _run_trunk(self, "", feats)
        coarse = torch.cat([pts[3], xyz[3]], 1)
        heads = [getattr(self, f"{h}_conv_2")(F.relu(getattr(self, f"{h}_bn_1")(getattr(self, f"{h}_conv_1")(coarse)))) for h in ("offset", "dist")]
        return [up, pts[3], xyz[0], xyz[3]] + heads


class TsgSegNet(nn.Module):
    """tsg_seg_module.get_model: two trunks on a cropped tooth neighbourhood (36 input channels; the second sees the first's
    foreground probabilities), a per-point mask each, and a tooth-id head on the max over ALL 256 coarse points -- the reference's
    one PointNetSetAbstraction(group_all=True), which runs on tgn_sa_all_mlp2_max in eval mode."""

    def __init__(self, input_feature_num=36):
        super().__init__()
        _msg_sa(self, "_1", input_feature_num)
        _msg_fp(self, "_1", input_feature_num)
        self.pd_mask_1, self.pd_mask_1_softmax, self.wt_mask_1 = nn.Conv1d(32, 2, 1), nn.Softmax(dim=1), nn.Conv1d(32, 1, 1)
        _msg_sa(self, "_2", input_feature_num + 2)                       # (module order = the reference's: state_dict keys line up)
        self.flatten_sa = PointNetSetAbstraction(None, None, None, 512 + 3, [256, 512], True)
        _msg_fp(self, "_2", input_feature_num + 2)
        self.pd_mask_2 = nn.Conv1d(32, 1, 1)
        self.fc1, self.bn1, self.fc2 = nn.Linear(512, 256), nn.LayerNorm(256), nn.Linear(256, 17)
        nn.init.zeros_(self.fc2.weight)
        nn.init.zeros_(self.fc2.bias)

    @_one_index_check
    def forward(self, feats):
        """feats (B, 36, N), xyz first -> (pd_1 (B,2,N), weight_1 (B,1,N), pd_2 (B,1,N), id_pred (B,17)) (:45-79)"""
        up1, _, _ = _run_trunk(self, "_1", feats)
        pd_1 = self.pd_mask_1_softmax(self.pd_mask_1(up1))
        weight_1 = self.wt_mask_1(up1)
        up2, xyz, pts = _run_trunk(self, "_2", torch.cat([feats, pd_1], 1))
        _, pooled = self.flatten_sa(xyz[3], pts[3])
        id_pred = self.fc2(F.relu(self.bn1(self.fc1(pooled.view(feats.shape[0], 512)))))
        return pd_1, weight_1, self.pd_mask_2(up2), id_pred


class _LatentMLP(nn.Module):
    """blocks.py:159-194 with ftype 'latent': Linear + BatchNorm + ReLU to base_fdim, held as `.infer`."""

    def __init__(self, fdim, d_out):
        super().__init__()
        self.infer = nn.Sequential(nn.Linear(fdim, d_out), nn.BatchNorm1d(d_out), nn.ReLU(inplace=True))

    def forward(self, x):
        if PT._frozen(self, x) and x.dtype == torch.float32:
            return PT.mlp_eval(self.infer, x)
        return PT.mlp_train(self.infer, x)


class MultiHead(nn.Module):
    """heads.py:13-61 for `multi: {stage: Ua, ftype: latent, combine: concat}`: every decoder stage's features go through
    their own latent MLP, are carried to the finest stage by nearest-neighbour interpolation (pointops.interpolation, k = 1)
    and concatenated in front of one linear classifier."""

    def __init__(self, fdims, k, base_fdim=32):
        super().__init__()
        self.infer_list = nn.ModuleList([_LatentMLP(f, base_fdim) for f in fdims])
        self.cls = nn.Linear(base_fdim * len(fdims), k)

    def forward(self, up_list):
        p0, _, o0 = up_list[0]
        cols = []
        for i, ((p, x, o), mlp) in enumerate(zip(up_list, self.infer_list)):
            y = mlp(x)
            cols.append(y if i == 0 else pointops.interpolation(p, p0, y.contiguous(), o, o0, k=1))
        return self.cls(torch.cat(cols, 1))


class PointTransformerSeg(nn.Module):
    def __init__(self, c=6, k=17, planes=(32, 64, 128, 256, 512), blocks=(2, 3, 4, 6, 3), stride=(1, 4, 4, 4, 4),
                 nsample=(36, 24, 24, 24, 24), share_planes=8):
        super().__init__()
        self.c, self.k = c, k
        in_planes = c
        for i in range(5):
            layers = [PT.TransitionDown(in_planes, planes[i], stride[i], nsample[i])]
            in_planes = planes[i]
            layers += [PT.PointTransformerBlock(in_planes, in_planes, share_planes, nsample[i]) for _ in range(1, blocks[i])]
            setattr(self, f"enc{i + 1}", nn.Sequential(*layers))
        for i in range(4, -1, -1):
            layers = [PT.TransitionUp(in_planes, None if i == 4 else planes[i])]
            in_planes = planes[i]
            layers.append(PT.PointTransformerBlock(in_planes, in_planes, share_planes, nsample[i]))
            setattr(self, f"dec{i + 1}", nn.Sequential(*layers))
        self.mask_head = MultiHead(planes, 2, planes[0])
        self.cls_head = MultiHead(planes, k, planes[0])
        self.offset_head = MultiHead(planes, 3, planes[0])

    @_one_index_check
    def forward(self, inputs):
        """inputs: [features (B, C, N)] -> [cls (B, k, N), offset (1, 3, N) or None, None, x1 (B*N, planes[0])]
        (cbl_point_transformer_module.py:196-216 without the training criterion)."""
        feats = inputs[0]
        B, C, N = feats.shape
        pxo = feats.permute(0, 2, 1)
        x = pxo.reshape(-1, C).contiguous()
        p = pxo[:, :, :3].reshape(-1, 3).contiguous()
        o = pointops.register_offsets(torch.arange(1, B + 1, dtype=torch.int32, device=feats.device) * N,
                                      [N * (i + 1) for i in range(B)])
        down, cur = [], [p, x, o]
        for i in range(1, 6):
            cur = getattr(self, f"enc{i}")(cur)
            down.append(cur)
        up, above = [None] * 5, None
        for i in range(4, -1, -1):
            pi, xi, oi = down[i]
            dec = getattr(self, f"dec{i + 1}")
            head = dec[0]([pi, xi, oi]) if above is None else dec[0]([pi, xi, oi], above)
            xi = dec[1:]([pi, head, oi])[1]
            above = up[i] = [pi, xi, oi]
        cls = self.cls_head(up).view(B, N, self.k).permute(0, 2, 1)
        offset = self.offset_head(up).view(B, N, 3).permute(0, 2, 1) if B == 1 else None
        return [cls, offset, None, up[0][1]]
