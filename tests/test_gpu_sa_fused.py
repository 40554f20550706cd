"""GPU: the fused set-abstraction kernels (tgn_sa_point_transform on the fp32 matrix cores, tgn_sa_gather_max,
tgn_sa_gather_act, tgn_sa_direct_max) against the CPU oracle's float64 restatement of the reference layer
(oracle/cpu.py::set_abstraction_first_layer, pointnet2_utils.py:162-169/229-236/281-294) and against outputs of the
reference's own modules (tests/golden/make_golden_r2_sa.py) -- at shapes where the fused path is taken on its own.
Tolerance: elementwise, |got - want| <= 1e-5 * (1 + |want|) (the kernels and the reference's BLAS differ in summation order only)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, want, what="", tol=1e-5):
    """ELEMENTWISE: |got - want| <= tol * (1 + |want|) for every entry (north_star: grouped features within 1e-5 fp32) --
    a small output next to a large one is held to its own size, not to the tensor's maximum."""
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    rel = np.abs(got - want) / (1.0 + np.abs(want))
    worst = int(np.argmax(rel)) if rel.size else 0
    assert rel.size == 0 or rel.flat[worst] <= tol, (f"{what}: worst elementwise error {rel.flat[worst]:.3e} > {tol} "
                                                    f"(got {got.flat[worst]!r}, want {want.flat[worst]!r})")


@pytest.mark.parametrize("M,D,C1", [(1000, 128, 512), (4096, 512, 1024), (777, 61, 24), (300, 125, 784), (129, 0, 32),
                                     (5000, 6, 128), (2048, 64, 96), (257, 1024, 100)])
def test_point_transform_is_an_exact_fp32_contraction(dev, M, D, C1):
    """A = [points, xyz] @ Wt on v_mfma_f32_32x32x2_f32: every tile shape class (partial row / column / K tiles,
    widths that are no multiple of 4), asymmetric operands (a transposed fragment would not pass)."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    g = torch.Generator().manual_seed(M + D + C1)
    xyz = torch.randn(1, M, 3, generator=g).to(dev)
    pts = torch.randn(1, M, D, generator=g).to(dev) if D else None
    Wt = (torch.randn(D + 3, C1, generator=g) * torch.linspace(0.5, 2.0, C1)).to(dev)
    A = U.sa_point_transform(xyz, pts, Wt)
    full = torch.cat([pts, xyz], -1) if D else xyz
    ref = (full.double() @ Wt.double())[0]
    mag = (full.double().abs() @ Wt.double().abs())[0]                     # sum |a||w|: what fp32 rounding scales with
    err = (A[0].double() - ref).abs()
    assert float((err / (mag + 1e-30)).max()) < 4e-7 * max(1.0, (D + 3) ** 0.5 / 4), float((err / mag).max())
    # bitwise: an fp32 fma chain in k order (the MFMA's definition) gives the same bits on a small case
    if M <= 1000 and D + 3 <= 64:
        a = full[0].cpu().numpy().astype(np.float32)
        w = Wt.cpu().numpy().astype(np.float32)
        acc = np.zeros((M, C1), np.float32)
        for k in range(D + 3):
            acc = np.float32(np.float64(a[:, k:k + 1]) * np.float64(w[k:k + 1, :]) + np.float64(acc)).astype(np.float32)
        assert np.array_equal(A[0].cpu().numpy(), acc)


def _layer(dev, D, C1, seed, xyz_first):
    torch.manual_seed(seed)
    conv = torch.nn.Conv2d(3 + D, C1, 1).to(dev)
    bn = torch.nn.BatchNorm2d(C1).to(dev).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.4, 2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
    return conv, bn


@pytest.mark.parametrize("N,S,K,D,C1", [(4096, 1024, 32, 128, 512), (1024, 256, 32, 512, 1024), (6000, 1024, 32, 6, 128),
                                         (900, 100, 16, 6, 64), (700, 50, 64, 13, 256), (500, 60, 7, 61, 100),
                                         (800, 90, 48, 0, 32), (640, 33, 36, 200, 784)])
@pytest.mark.parametrize("xyz_first", [True, False])
def test_fused_level_and_first_layer_vs_oracle(dev, oracle, N, S, K, D, C1, xyz_first):
    """sa_level_max (direct or transform + gather-max) and sa_first_layer (transform + gather-act) against the float64
    restatement of grouping -> conv -> BN -> ReLU (-> max) on real ball-query neighbourhoods."""
    from toothgroupnetwork_amd import pointnet2_utils as U, synth
    B = 2
    rng = np.random.default_rng(N + K + D)
    pts6 = synth.scan_batch(B, N, "arch", seed=N % 97)
    xyz = np.ascontiguousarray(pts6[:, :, :3])
    feat = rng.normal(size=(B, N, D)).astype(np.float32) if D else None
    tx, tf = T(xyz, dev), (T(feat, dev) if D else None)
    fidx = U.farthest_point_sample(tx, S)
    new_xyz = U.index_points(tx, fidx)
    idx = U.query_ball_point(0.3, K, tx, new_xyz)
    conv, bn = _layer(dev, D, C1, 7, xyz_first)
    args = (xyz, new_xyz.cpu().numpy(), feat, idx.cpu().numpy(), conv.weight.detach().reshape(C1, -1).cpu().numpy(),
            conv.bias.detach().cpu().numpy(), bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(),
            bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy(), bn.eps, xyz_first)
    with torch.no_grad():
        got = U.sa_level_max(tx, new_xyz, tf, idx, conv, bn, xyz_first)
        close(got.cpu().numpy(), oracle.set_abstraction_first_layer(*args, reduce_max=True), "level (max)")
        if K <= 64 and C1 % 4 == 0:
            got = U.sa_first_layer(tx, new_xyz, tf, idx.to(torch.int32), conv, bn, xyz_first)
            close(got.cpu().numpy(), oracle.set_abstraction_first_layer(*args, reduce_max=False), "first layer")


def _bn_np(bn):
    return (bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(), bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy())


# (N, S, K, D, C1, C2): the reference networks' own levels first -- pointnet_pp.py:13-15 (sa1: direct first layer; sa2; sa3:
# 784 is no multiple of 32), tsg_centroid / tsg_seg (:10-12 / :11-13: 39-wide rows, 196 -> padded to 208) -- then odd shapes:
# K below 32, K between 32 and 64, a single query, widths that are no multiple of the tiles
MLP2_SHAPES = [(6000, 1024, 32, 6, 128, 128), (6000, 1024, 64, 6, 128, 128), (1024, 512, 32, 256, 256, 512),
               (1024, 512, 64, 256, 256, 512), (512, 256, 32, 1024, 784, 1024), (512, 256, 64, 1024, 784, 1024),
               (3000, 1024, 32, 36, 32, 32), (512, 256, 64, 256, 196, 256), (700, 50, 7, 13, 20, 36), (640, 33, 36, 200, 72, 100),
               (300, 1, 48, 0, 16, 4), (900, 77, 17, 61, 100, 260),
               (400, 20, 32, 40, 16, 48)]   # commuted form with ONE K tile (C1 = 16): the K loop's first trip is also its last


@pytest.mark.parametrize("N,S,K,D,C1,C2", MLP2_SHAPES)
@pytest.mark.parametrize("xyz_first", [True, False])
@pytest.mark.parametrize("bf16x3", [128, 256, False])
def test_two_layer_level_vs_oracle(dev, oracle, N, S, K, D, C1, C2, xyz_first, bf16x3, monkeypatch):
    """tgn_sa_mlp2_max_bf16x3 (second layer as six bf16 MFMAs per fp32 product, the default; both workgroup tiles) and tgn_sa_mlp2_max
    (fp32 MFMA) -- the
    whole two-layer level in one kernel, direct or commuted first layer -- against the float64 restatement of
    grouping -> [conv -> BN -> ReLU] x 2 -> max on real ball-query neighbourhoods, elementwise 1e-5."""
    from toothgroupnetwork_amd import pointnet2_utils as U, synth
    from toothgroupnetwork_amd import _lib
    monkeypatch.setattr(U, "SA_BF16X3", bool(bf16x3))
    prev_tile = _lib.set_tuning("sa_tile", int(bf16x3))          # both workgroup tiles of the bf16x3 kernel: 128 x 128 and 256 x 256
    B = 2
    rng = np.random.default_rng(N + K + D + C2)
    pts6 = synth.scan_batch(B, N, "arch", seed=N % 89)
    xyz = np.ascontiguousarray(pts6[:, :, :3])
    feat = rng.normal(size=(B, N, D)).astype(np.float32) if D else None
    tx, tf = T(xyz, dev), (T(feat, dev) if D else None)
    new_xyz = U.index_points(tx, U.farthest_point_sample(tx, S))
    idx = U.query_ball_point(0.3, K, tx, new_xyz)
    conv1, bn1 = _layer(dev, D, C1, 11, xyz_first)
    torch.manual_seed(12)
    conv2 = torch.nn.Conv2d(C1, C2, 1).to(dev)
    bn2 = torch.nn.BatchNorm2d(C2).to(dev).eval()
    with torch.no_grad():
        bn2.running_mean.normal_(0, 0.3)
        bn2.running_var.uniform_(0.4, 2.0)
        bn2.weight.uniform_(0.5, 1.5)
        bn2.bias.normal_(0, 0.2)
    layers = [(c.weight.detach().reshape(c.out_channels, -1).cpu().numpy(), c.bias.detach().cpu().numpy()) + _bn_np(b_)
              for c, b_ in ((conv1, bn1), (conv2, bn2))]
    want = oracle.set_abstraction_mlp(xyz, new_xyz.cpu().numpy(), feat, idx.cpu().numpy(), layers, bn1.eps, xyz_first)
    with torch.no_grad():
        for index in (idx, idx.to(torch.int32)):
            got = U.sa_level_mlp2_max(tx, new_xyz, tf, index, [conv1, conv2], [bn1, bn2], xyz_first)
            close(got.cpu().numpy(), want, f"two-layer level ({index.dtype})")
    _lib.set_tuning("sa_tile", prev_tile)
    err = np.abs(got.cpu().numpy().astype(np.float64) - want) / (1.0 + np.abs(want))
    print(f"\n{f'bf16x3 tile {bf16x3}' if bf16x3 else 'fp32-mfma'} ({N},{S},{K},{D},{C1},{C2}) xyz_first={xyz_first}: max {err.max():.2e} "
          f"rms {np.sqrt((err ** 2).mean()):.2e} (bound 1e-5)")


def test_two_layer_level_reports_a_bad_index(dev):
    from toothgroupnetwork_amd import pointnet2_utils as U
    torch.manual_seed(0)
    xyz = torch.rand(1, 200, 3, device=dev)
    feat = torch.randn(1, 200, 40, device=dev)
    new_xyz = xyz[:, :10].contiguous()
    idx = torch.randint(0, 200, (1, 10, 16), device=dev)
    convs = [torch.nn.Conv2d(43, 32, 1).to(dev), torch.nn.Conv2d(32, 32, 1).to(dev)]
    bns = [torch.nn.BatchNorm2d(32).to(dev).eval(), torch.nn.BatchNorm2d(32).to(dev).eval()]
    with torch.no_grad():
        U.sa_level_mlp2_max(xyz, new_xyz, feat, idx, convs, bns, True)
        idx[0, 3, 5] = 200                                  # what an empty ball yields (pointnet2_utils.py:136-141)
        with pytest.raises(IndexError):
            U.sa_level_mlp2_max(xyz, new_xyz, feat, idx, convs, bns, True)
        idx[0, 3, 5] = -1                                   # negative indices wrap like torch's indexing
        a = U.sa_level_mlp2_max(xyz, new_xyz, feat, idx, convs, bns, True)
        idx[0, 3, 5] = 199
        assert torch.equal(a, U.sa_level_mlp2_max(xyz, new_xyz, feat, idx, convs, bns, True))


def test_fused_modules_match_the_reference_modules(dev, golden_r2, monkeypatch):
    """The drop-in modules in eval mode take the fused path BY THEMSELVES at these shapes (nothing is patched; a spy only
    counts) and reproduce the reference modules' outputs (same weights) within 1e-5 elementwise."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    g = golden_r2
    state = torch.load(os.path.join(GOLDEN, "module_weights_r2.pt"))
    calls = {"level": 0, "first": 0, "mlp2": 0, "group": 0}
    real_level, real_first, real_group, real_mlp2 = U.sa_level_max, U.sa_first_layer, U.group_points, U.sa_level_mlp2_max

    def spy(name, fn):
        def wrapped(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return wrapped
    monkeypatch.setattr(U, "sa_level_max", spy("level", real_level))
    monkeypatch.setattr(U, "sa_first_layer", spy("first", real_first))
    monkeypatch.setattr(U, "group_points", spy("group", real_group))
    monkeypatch.setattr(U, "sa_level_mlp2_max", spy("mlp2", real_mlp2))
    xyz, feat, pts6 = T(g["sa_in_xyz_cf"], dev), T(g["sa_in_feat_cf"], dev), T(g["sa_in_pts6_cf"], dev)
    mods = {
        "ssg_wide": (U.PointNetSetAbstraction(128, 0.25, 32, 3 + 64, [96], False), feat),
        "ssg_narrow": (U.PointNetSetAbstraction(96, 0.3, 16, 3 + 6, [64], False), pts6),
        "msg": (U.PointNetSetAbstractionMsg(128, [0.2, 0.3], [16, 32], 64, [[128], [64, 96]]), feat),
    }
    for name, (mod, f) in mods.items():
        mod = mod.to(dev).eval()
        mod.load_state_dict(state[name])
        with torch.no_grad():
            nx, nf = mod(xyz, f)
        assert np.array_equal(nx.cpu().numpy(), g[f"sa_{name}_xyz"]), name      # same FPS indices -> identical centres
        close(nf.cpu().numpy(), g[f"sa_{name}_feat"], name)
    # ssg_wide, ssg_narrow and the single-layer Msg branch: one-layer level kernels; the two-layer Msg branch: the chained kernel
    assert calls == {"level": 3, "first": 0, "mlp2": 1, "group": 0}, calls


@pytest.mark.parametrize("pipeline", [False, True])
@pytest.mark.parametrize("mlp", [[[64], [96], [160]], [[48, 64], [80, 96], [100, 160]]])
def test_hotpath_fused_levels_vs_oracle(dev, oracle, pipeline, mlp):
    """bench.py --fused: three chained fused levels (level l's output is level l+1's feature input), one- and two-layer
    shared MLPs, against the oracle chain FPS -> ball query -> float64 MLP + max, level by level, on every scan."""
    from toothgroupnetwork_amd import hotpath, synth
    shape = dict(n=3000, npoint=[512, 128, 32], radius=[0.15, 0.3, 0.6], nsample=[32, 32, 16], d=[6, mlp[0][-1], mlp[1][-1]], mlp=mlp)
    B = 3
    scans = synth.scan_batch(B, 3000, "arch", 41)
    pts = T(scans, dev)
    xyz = pts[:, :, :3].contiguous()
    hp = hotpath.HotPath(B, dev, shape=shape, pipeline=pipeline, fused=True)
    for _ in range(3):
        levels = hp.run(xyz, [pts])
    torch.cuda.synchronize()
    cur, feat = scans[:, :, :3].copy(), scans
    for li, lv in enumerate(levels):
        S, K = lv["S"], lv["K"]
        fidx = oracle.farthest_point_sample(cur, S)
        assert np.array_equal(lv["fps_idx"].cpu().numpy(), fidx)
        new_xyz = oracle.index_points(cur, fidx)
        gidx = oracle.query_ball_point(shape["radius"][li], K, cur, new_xyz)
        assert np.array_equal(lv["group_idx"].cpu().numpy(), gidx)
        layers = [(W, b, np.ones(len(b)), np.zeros(len(b)), np.zeros(len(b)), np.full(len(b), 1.0 - 1e-5)) for W, b in lv["layers"]]
        want = oracle.set_abstraction_mlp(cur, new_xyz, feat, gidx, layers, 1e-5, True)   # (identity BatchNorm)
        close(lv["out"].cpu().numpy(), want, f"level {li + 1}")
        cur, feat = new_xyz, lv["out"].cpu().numpy()                  # the GPU's own fp32 output feeds the next level in both chains


@pytest.mark.parametrize("pipeline", [False, True])
def test_hotpath_fused_multi_scale_levels_vs_oracle(dev, oracle, pipeline):
    """bench.py --shape B --fused 1 in small: two (radius, nsample) branches per level, each one chained two-layer kernel writing
    its columns of the level's concatenated output (pointnet2_utils.py:296-298), [features, centred xyz] row order."""
    from toothgroupnetwork_amd import hotpath, synth
    mlp = [[[24, 32], [40, 48]], [[64, 72], [64, 56]]]
    shape = dict(n=3000, npoint=[384, 96], radius=[[0.1, 0.2], [0.25, 0.5]], nsample=[[16, 32], [32, 64]], d=[6, 80], mlp=mlp,
                 xyz_first=False)
    B = 2
    scans = synth.scan_batch(B, 3000, "arch", 43)
    pts = T(scans, dev)
    xyz = pts[:, :, :3].contiguous()
    hp = hotpath.HotPath(B, dev, shape=shape, pipeline=pipeline, fused=True)
    for _ in range(3):
        levels = hp.run(xyz, [pts])
    torch.cuda.synchronize()
    cur, feat = scans[:, :, :3].copy(), scans
    for li, lv in enumerate(levels):
        fidx = oracle.farthest_point_sample(cur, lv["S"])
        new_xyz = oracle.index_points(cur, fidx)
        col = 0
        for br, (r, K) in zip(lv["branches"], hotpath._branches(shape["radius"][li], shape["nsample"][li])):
            gidx = oracle.query_ball_point(r, K, cur, new_xyz)
            assert np.array_equal(br["group_idx"].cpu().numpy(), gidx)
            # br["layers"] holds (C_out, C_in) matrices in [x, y, z, features] column order; the oracle groups [features, xyz] here
            (W1, b1), (W2, b2) = br["layers"]
            W1f = np.concatenate([W1[:, 3:], W1[:, :3]], 1)
            ident = lambda b: (np.ones(len(b)), np.zeros(len(b)), np.zeros(len(b)), np.full(len(b), 1.0 - 1e-5))
            want = oracle.set_abstraction_mlp(cur, new_xyz, feat, gidx, [(W1f, b1) + ident(b1), (W2, b2) + ident(b2)], 1e-5, False)
            c2 = W2.shape[0]
            close(lv["out"][:, :, col:col + c2].cpu().numpy(), want, f"level {li + 1} branch K={K}")
            col += c2
        assert col == lv["out"].shape[2]
        cur, feat = new_xyz, lv["out"].cpu().numpy()


@pytest.mark.parametrize("M,D,C1", [(6000, 256, 256), (4099, 1024, 784), (1000, 61, 100), (300, 13, 208), (129, 0, 16)])
def test_point_transform_bf16x3_vs_float64(dev, M, D, C1):
    """tgn_sa_point_transform_bf16x3 (per-point first layer as six bf16 MFMAs per fp32 product) and the fp32-MFMA form against
    [points, xyz] @ Wt in float64, elementwise 1e-5; odd row counts, widths that need K padding, D = 0."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    g = torch.Generator().manual_seed(M + D)
    xyz = (torch.rand(1, M, 3, generator=g) * 2 - 1).to(dev)
    pts = torch.randn(1, M, D, generator=g).to(dev) if D else None
    Wt = (torch.randn(D + 3, C1, generator=g) / (D + 3) ** 0.5).to(dev)
    rows = xyz[0] if pts is None else torch.cat([pts[0], xyz[0]], 1)
    want = (rows.double() @ Wt.double()).cpu().numpy()
    exact = U.sa_point_transform(xyz, pts, Wt)[0].cpu().numpy()
    split = U.sa_point_transform(xyz, pts, Wt, U.split_point_transform(Wt))[0].cpu().numpy()
    close(exact, want, "fp32-MFMA transform")
    close(split, want, "bf16x3 transform")
    e1 = np.abs(exact - want) / (1 + np.abs(want))
    e2 = np.abs(split - want) / (1 + np.abs(want))
    print(f"\npoint transform ({M},{D},{C1}): fp32-mfma max {e1.max():.2e} rms {np.sqrt((e1 ** 2).mean()):.2e} | bf16x3 max {e2.max():.2e} "
          f"rms {np.sqrt((e2 ** 2).mean()):.2e}")
