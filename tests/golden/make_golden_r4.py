#!/usr/bin/env python3
"""Round-4 golden fixtures (tests/golden/reference_cpu_r4.npz), produced by running the REFERENCE's own Python on CPU in the
build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r4.py [part ...]      (parts: group_all modules64 tsegnet nets24k train)

  (a) `group_all`: external_libs/pointnet2_utils/pointnet2_utils.py:198-239 `PointNetSetAbstraction(..., group_all=True)` in
      eval mode -- the only form of that module any reference model builds -- at the reference's own instantiation
      (models/modules/tsg_seg_module.py:28: `PointNetSetAbstraction(None, None, None, 512+3, [256, 512], True)` over the 256
      points of the last set-abstraction level) and at a ragged narrow shape (100 points, 3+5 channels -> [16, 32]).
  (a') `modules64`: the module-level fixtures of make_golden.py evaluated in float64 on the same indices.
  (a'') `tsegnet`: the two modules of tsegnet (tsg_centroid_module / tsg_seg_module get_model), whole, eval mode.
  (b) `nets24k`: the WHOLE networks of BASELINE configs 2 and 4 at the configs' own size, ONE 24 000-point scan:
      models/modules/pointnet_pp.py:43-70 `get_model.forward` and cbl_point_transformer_module.py:93-216
      `PointTransformerSeg.forward`, eval mode, served exactly as in make_golden_r3.py (only the CUDA-only FPS / kNN come from
      the reference's CPU FPS / the oracle).  Outputs are stored strided to keep the fixture small.
  (c) `train`: BASELINE config 3's step on the first-stage network -- the reference's `PointTransformerSeg` in TRAIN mode
      (BatchNorm batch statistics), forward on one scan, the loss terms `FpsGroupingNetworkModel.get_loss` applies to its
      outputs (models/fps_grouping_network_model.py:8-24 with train_configs/tgnet_fps.py:16-24's weights: models/tgn_loss.py
      `tooth_class_loss`, `batch_center_offset_loss`; the contrastive-boundary term is the reference's Python on top of
      `pointops.knnquery`, pinned separately, and is left out), backward.  Stored: the loss terms and, for every parameter,
      the gradient's norm and a strided sample of its entries.

Every network runs twice, as in round 3: in float32 (what the reference computes) and in float64 on the SAME indices (the
exact value of the same function), so that the tests can tell the reference's own fp32 rounding noise from a discrepancy."""
import os
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("TGN_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_golden import load_reference, ref_fps  # noqa: E402
from make_golden_r3 import cpu_as_cuda, rel_err  # noqa: E402
from oracle import cpu as O  # noqa: E402
from seeded import seeded_fill  # noqa: E402
from toothgroupnetwork_amd import synth  # noqa: E402

PT_CFG = dict(c=6, k=17, planes=[32, 64, 128, 256, 512], stride=[1, 4, 4, 4, 4], nsample=[36, 24, 24, 24, 24],
              blocks=[2, 3, 4, 6, 3], block_num=5)
GRAD_SAMPLES = 64          # entries kept per parameter gradient (evenly strided over the flattened tensor)


def group_all(R, out):
    for tag, B, N, D, mlp, seed in (("tsg", 3, 256, 512, [256, 512], 41), ("ragged", 2, 100, 5, [16, 32], 42)):
        g = torch.Generator().manual_seed(seed)
        xyz = torch.rand(B, 3, N, generator=g) * 2 - 1
        pts = torch.randn(B, D, N, generator=g)
        mod = R.PointNetSetAbstraction(None, None, None, D + 3, mlp, True).eval()
        seeded_fill(mod, seed)
        with torch.no_grad():
            nx32, y32 = mod(xyz, pts)
            nx64, y64 = mod.double()(xyz.double(), pts.double())
        assert tuple(nx32.shape) == (B, 3, 1) and not nx32.any()
        print(f"  PointNetSetAbstraction(group_all)[{tag}] {tuple(y32.shape)}  |fp32 - fp64| / (1 + |fp64|) = "
              f"{rel_err(y32.numpy(), y64.numpy()):.2e}")
        out[f"ga_{tag}_xyz"], out[f"ga_{tag}_pts"] = xyz.numpy(), pts.numpy()
        out[f"ga_{tag}_out_32"], out[f"ga_{tag}_out_64"] = y32.numpy(), y64.numpy().astype(np.float32)


def modules64(R, out):
    """The module-level fixtures of make_golden.py (reference_cpu.npz: `mod_*`, float32 on CPU BLAS) evaluated once more in float64
    on the same indices, so that tests/test_gpu_modules.py can hold the drop-in modules to 1e-5 of the exact value instead of 1e-4
    of another fp32 evaluation."""
    gold = np.load(os.path.join(HERE, "reference_cpu.npz"))
    sd = torch.load(os.path.join(HERE, "module_weights.pt"))
    sa = R.PointNetSetAbstractionMsg(128, [0.1, 0.2], [8, 16], 6, [[16, 24], [16, 32]]).eval()
    ssg = R.PointNetSetAbstraction(64, 0.2, 16, 6 + 3, [16, 32], False).eval()
    fp = R.PointNetFeaturePropagation(56 + 6, [32, 16]).eval()
    for m_, k in ((sa, "sa"), (ssg, "ssg"), (fp, "fp")):
        m_.load_state_dict(sd[k])
        m_.double()
    xyz_cf, pts_cf = torch.from_numpy(gold["mod_xyz_cf"]), torch.from_numpy(gold["mod_pts_cf"])
    keep_fps, keep_sqd = R.farthest_point_sample, R.square_distance
    R.farthest_point_sample = lambda x, n: torch.from_numpy(ref_fps(R, x.detach().float().numpy(), n))
    R.square_distance = lambda a, b: keep_sqd(a.float(), b.float()).double()
    try:
        with torch.no_grad():
            sa_xyz, sa_feat = sa(xyz_cf.double(), pts_cf.double())
            ssg_xyz, ssg_feat = ssg(xyz_cf.double(), pts_cf.double())
            # the fixture feeds the fp32 reference outputs of `sa` into `fp` (make_golden.py:198): same inputs here
            fp_out = fp(xyz_cf.double(), torch.from_numpy(gold["mod_sa_xyz"]).double(), pts_cf.double(),
                        torch.from_numpy(gold["mod_sa_feat"]).double())
    finally:
        R.farthest_point_sample, R.square_distance = keep_fps, keep_sqd
    assert np.array_equal(sa_xyz.float().numpy(), gold["mod_sa_xyz"]) and np.array_equal(ssg_xyz.float().numpy(), gold["mod_ssg_xyz"])
    for n_, v in (("mod_sa_feat", sa_feat), ("mod_ssg_feat", ssg_feat), ("mod_fp_out", fp_out)):
        print(f"  {n_}: reference fp32 vs fp64 {rel_err(gold[n_], v.numpy()):.2e}")
        out[n_ + "_64"] = v.numpy().astype(np.float32)


def pointnet_pp_24k(out):
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.pointnet_pp as M
    U = sys.modules["external_libs.pointnet2_utils.pointnet2_utils"]
    assert U.__file__.startswith(REFERENCE), U.__file__
    scans = synth.scan_batch(1, 24000, "arch", seed=401)
    feats = torch.from_numpy(np.ascontiguousarray(scans.transpose(0, 2, 1)))
    net = M.get_model().eval()
    seeded_fill(net, 31)
    keep_fps, keep_sqd = U.farthest_point_sample, U.square_distance
    U.farthest_point_sample = lambda x, n: torch.from_numpy(ref_fps(U, x.detach().float().numpy(), n))
    t0 = time.time()
    try:
        with torch.no_grad():
            y32 = net([feats])
            U.square_distance = lambda a, b: keep_sqd(a.float(), b.float()).double()
            y64 = net.double()([feats.double()])
    finally:
        U.farthest_point_sample, U.square_distance = keep_fps, keep_sqd
    names = ["l0_points", "l3_points", "l0_xyz", "l3_xyz", "offset", "dist", "cls"]
    print(f"  pointnet_pp.get_model at 24 000 points: {time.time() - t0:.0f} s")
    for n_, a, b in zip(names, y32, y64):
        print(f"    {n_:10s} {tuple(a.shape)}  |fp32 - fp64| / (1 + |fp64|) = {rel_err(a.numpy(), b.numpy()):.2e}")
    out["pnpp24_seed"] = np.array([401])
    sub = {"l0_points": (slice(None), slice(0, None, 8), slice(0, None, 16)), "l3_points": (slice(None), slice(0, None, 8)),
           "offset": (slice(None), slice(None), slice(0, None, 4)), "dist": (slice(None), slice(None), slice(0, None, 4)),
           "cls": (slice(None), slice(None), slice(0, None, 4))}
    for n_, a, b in zip(names, y32, y64):
        if n_ == "l0_xyz":
            continue
        s = sub.get(n_, (slice(None),))
        out[f"pnpp24_{n_}_64"] = b.numpy()[s].astype(np.float32)
        if n_ in ("cls", "offset", "l3_xyz"):
            out[f"pnpp24_{n_}_32"] = a.numpy()[s]


def tsegnet_modules(out):
    """The two modules of tsegnet (models/modules/tsegnet.py:15-16): tsg_centroid_module.get_model on two 3000-point scans and
    tsg_seg_module.get_model -- whose `flatten_sa` is the reference's one PointNetSetAbstraction(group_all=True) -- on two cropped
    neighbourhoods of 3072 points with 36 channels (xyz + 33 synthetic feature channels), eval mode, float32 and float64 on the same
    indices.  Only `farthest_point_sample` (CUDA-only) is served, by the reference's CPU FPS with start 0."""
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.tsg_centroid_module as CM
    import models.modules.tsg_seg_module as SM
    U = sys.modules["external_libs.pointnet2_utils.pointnet2_utils"]
    assert U.__file__.startswith(REFERENCE), U.__file__
    keep_fps, keep_sqd = U.farthest_point_sample, U.square_distance
    U.farthest_point_sample = lambda x, n: torch.from_numpy(ref_fps(U, x.detach().float().numpy(), n))
    try:
        for tag, M, seed, N, C in (("cent", CM, 34, 3000, 6), ("seg", SM, 35, 3072, 36)):
            scans = synth.scan_batch(2, N, "arch", seed=400 + seed)                      # (2, N, 6)
            g = torch.Generator().manual_seed(seed)
            feats = torch.from_numpy(np.ascontiguousarray(scans.transpose(0, 2, 1)))
            if C > 6:
                feats = torch.cat([feats, 0.5 * torch.randn(2, C - 6, N, generator=g)], 1)
            net = M.get_model().eval()
            out[f"tsg_{tag}_params"] = np.array(seeded_fill(net, seed))
            U.square_distance = keep_sqd
            with torch.no_grad():
                y32 = net(feats)
                U.square_distance = lambda a, b: keep_sqd(a.float(), b.float()).double()
                y64 = net.double()(feats.double())
            out[f"tsg_{tag}_feats"] = feats.numpy()
            for i, (a, b) in enumerate(zip(y32, y64)):
                print(f"  tsg_{tag}[{i}] {tuple(a.shape)}  |fp32 - fp64| / (1 + |fp64|) = {rel_err(a.numpy(), b.numpy()):.2e}")
                sl = (slice(None), slice(None), slice(0, None, 4)) if a.dim() == 3 and a.shape[2] > 1000 else (slice(None),)
                out[f"tsg_{tag}_{i}_64"] = b.numpy()[sl].astype(np.float32)
                out[f"tsg_{tag}_{i}_32"] = a.numpy()[sl]
    finally:
        U.farthest_point_sample, U.square_distance = keep_fps, keep_sqd


def _serve_pointops(RP):
    def fps(xyz, offset, new_offset):
        return torch.from_numpy(O.furthestsampling(xyz.detach().float().numpy(), offset.numpy(), new_offset.numpy()).astype(np.int32))

    def knn(nsample, xyz, new_xyz, offset, new_offset):
        idx, dist = O.knnquery(int(nsample), xyz.detach().float().numpy(), new_xyz.detach().float().numpy(), offset.numpy(),
                               new_offset.numpy())
        return torch.from_numpy(idx), torch.from_numpy(dist).to(xyz.dtype)
    keep = RP.furthestsampling, RP.knnquery
    RP.furthestsampling, RP.knnquery = fps, knn
    return keep


def point_transformer_24k(out):
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.cbl_point_transformer.cbl_point_transformer_module as M
    import models.modules.cbl_point_transformer.blocks as RB
    RP = RB.pointops
    assert RP.__file__.startswith(REFERENCE), RP.__file__
    net = M.get_model(**PT_CFG).eval()
    seeded_fill(net, 32)
    keep = _serve_pointops(RP)
    t0 = time.time()
    try:
        scans = synth.scan_batch(1, 24000, "arch", seed=402)
        feats = torch.from_numpy(np.ascontiguousarray(scans.transpose(0, 2, 1)))
        with torch.no_grad():
            with cpu_as_cuda(torch.FloatTensor):
                y32 = net.float()([feats])
            with cpu_as_cuda(torch.DoubleTensor):
                y64 = net.double()([feats.double()])
    finally:
        RP.furthestsampling, RP.knnquery = keep
    print(f"  PointTransformerSeg at 24 000 points: {time.time() - t0:.0f} s")
    out["pt24_seed"] = np.array([402])
    for n_, i in (("cls", 0), ("offset", 1), ("x1", 3)):
        print(f"    {n_:7s} {tuple(y32[i].shape)}  |fp32 - fp64| / (1 + |fp64|) = {rel_err(y32[i].numpy(), y64[i].numpy()):.2e}")
        s = (slice(None), slice(None), slice(0, None, 4)) if n_ != "x1" else (slice(0, None, 4),)
        out[f"pt24_{n_}_64"] = y64[i].numpy()[s].astype(np.float32)
        out[f"pt24_{n_}_32"] = y32[i].numpy()[s]


def train_labels(xyz, teeth=16):
    """Synthetic per-point tooth labels 0..teeth-1 and -1 (gingiva) along the arch parameter, the value range
    generator.py / preprocess_data.py:38-44 produce (0..16 with 0 gingiva, shifted by -1 in generator.py)."""
    ang = np.arctan2(xyz[:, 1], xyz[:, 0])
    lo, hi = np.quantile(ang, 0.02), np.quantile(ang, 0.98)
    t = np.clip(((ang - lo) / (hi - lo) * teeth).astype(np.int64), 0, teeth - 1)
    t[xyz[:, 2] < np.quantile(xyz[:, 2], 0.35)] = -1
    return t


def train_step(out, N=int(os.environ.get("TGN_TRAIN_GOLDEN_POINTS", "24000"))):
    if REFERENCE not in sys.path:
        sys.path.append(REFERENCE)
    import models.modules.cbl_point_transformer.cbl_point_transformer_module as M
    import models.modules.cbl_point_transformer.blocks as RB
    import models.tgn_loss as TL
    RP = RB.pointops
    assert RP.__file__.startswith(REFERENCE) and TL.__file__.startswith(REFERENCE)
    scans = synth.scan_batch(1, N, "arch", seed=403)
    label = train_labels(scans[0, :, :3])
    feats = torch.from_numpy(np.ascontiguousarray(scans.transpose(0, 2, 1)))
    gt = torch.from_numpy(label).view(1, 1, N)
    weights = {"tooth_class_loss_1": 1.0, "offset_1_loss": 0.03, "offset_1_dir_loss": 0.03}      # train_configs/tgnet_fps.py:16-24
    keep = _serve_pointops(RP)
    res = {}
    try:
        for tag, ftype, dt in (("32", torch.FloatTensor, torch.float32), ("64", torch.DoubleTensor, torch.float64)):
            net = M.get_model(**PT_CFG).train()
            names = seeded_fill(net, 33)
            net = net.to(dt)
            t0 = time.time()
            with cpu_as_cuda(ftype):
                sem, offset, _, _ = net([feats.to(dt)])                                      # (1,17,N), (1,3,N)
                # FpsGroupingNetworkModel.get_loss (fps_grouping_network_model.py:8-16): the 17-class first-stage head here plays
                # tgnet's sem_1 with the full label range (class_num = 16 teeth + gingiva; tooth_class_loss shifts by +1)
                ce = TL.tooth_class_loss(sem, gt, 17)
                cen, dirl = TL.batch_center_offset_loss(offset, feats[:, :3, :].to(dt), gt)
                loss = weights["tooth_class_loss_1"] * ce + weights["offset_1_loss"] * cen + weights["offset_1_dir_loss"] * dirl
                loss.backward()
            print(f"  train step fp{tag}: loss {float(loss):.6f} (ce {float(ce):.6f}, centroid {float(cen):.6f}, dir {float(dirl):.6f}) "
                  f"{time.time() - t0:.0f} s")
            res[tag] = dict(loss=float(loss), ce=float(ce), cen=float(cen), dir=float(dirl), sem=sem.detach().numpy(),
                            offset=offset.detach().numpy(),
                            grads={n: p.grad.detach().numpy().astype(np.float64) for n, p in net.named_parameters() if p.grad is not None},
                            none=[n for n, p in net.named_parameters() if p.grad is None])
    finally:
        RP.furthestsampling, RP.knnquery = keep
    out["train_points"] = np.array([N])
    out["train_params"] = np.array(names)
    out["train_label"] = label.astype(np.int16)
    for tag in ("32", "64"):
        out[f"train_terms_{tag}"] = np.array([res[tag][k] for k in ("loss", "ce", "cen", "dir")], np.float64)
        out[f"train_sem_{tag}"] = res[tag]["sem"][:, :, ::16].astype(np.float32)
        out[f"train_offset_{tag}"] = res[tag]["offset"][:, :, ::16].astype(np.float32)
    gnames = sorted(res["64"]["grads"])
    assert gnames == sorted(res["32"]["grads"])
    out["train_grad_names"] = np.array(gnames)
    out["train_grad_none"] = np.array(res["64"]["none"] or [""])
    norms, samples, own = [], [], []
    for n in gnames:
        g64, g32 = res["64"]["grads"][n].reshape(-1), res["32"]["grads"][n].reshape(-1)
        pick = np.linspace(0, g64.size - 1, min(GRAD_SAMPLES, g64.size)).astype(np.int64)
        row64, row32 = np.zeros(GRAD_SAMPLES), np.zeros(GRAD_SAMPLES)
        row64[:pick.size], row32[:pick.size] = g64[pick], g32[pick]
        norms.append([np.linalg.norm(g64), np.linalg.norm(g32), np.linalg.norm(g32 - g64)])
        samples.append(np.stack([row64, row32]))
        own.append(np.linalg.norm(g32 - g64) / max(np.linalg.norm(g64), 1e-30))
    out["train_grad_norms"] = np.array(norms)                 # per parameter: |g64|, |g32|, |g32 - g64|
    out["train_grad_samples"] = np.array(samples)             # (P, 2, GRAD_SAMPLES): float64 run, float32 run
    own = np.array(own)
    print(f"  {len(gnames)} parameter gradients; reference fp32 vs fp64, relative L2 per parameter: median {np.median(own):.2e}, "
          f"max {own.max():.2e} ({gnames[int(own.argmax())]}); without gradient: {res['64']['none']}")


def main():
    torch.set_num_threads(8)
    parts = sys.argv[1:] or ["group_all", "modules64", "tsegnet", "nets24k", "train"]
    path = os.path.join(HERE, "reference_cpu_r4.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    R = load_reference()
    if "group_all" in parts:
        group_all(R, out)
    if "modules64" in parts:
        modules64(R, out)
    if "tsegnet" in parts:
        tsegnet_modules(out)
    if "nets24k" in parts:
        pointnet_pp_24k(out)
        point_transformer_24k(out)
    if "train" in parts:
        train_step(out)
    np.savez_compressed(path, **out)
    print(f"wrote tests/golden/reference_cpu_r4.npz ({os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
