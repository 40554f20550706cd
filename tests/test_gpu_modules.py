"""GPU (-m gpu): the nn.Module layer (set abstraction / feature propagation) against outputs of the
REFERENCE's modules computed on CPU with the same weights (tests/golden/make_golden.py), plus the
preprocess resampler and full-size, size-independent properties at BASELINE.json's shapes."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from toothgroupnetwork_amd import synth

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def weights():
    return torch.load(os.path.join(GOLDEN, "module_weights.pt"))


@pytest.fixture(scope="module")
def exact():
    """The module fixtures evaluated by the reference in float64 on the same indices (tests/golden/make_golden_r4.py modules64)."""
    return dict(np.load(os.path.join(GOLDEN, "reference_cpu_r4.npz")))


def close(got, want64, tol=1e-5):
    """elementwise |got - want| <= tol * (1 + |want|) against the exact (float64) value -- the contract's 1e-5"""
    got, want64 = np.asarray(got, np.float64), np.asarray(want64, np.float64)
    assert got.shape == want64.shape
    err = float(np.max(np.abs(got - want64) / (1.0 + np.abs(want64))))
    assert err <= tol, err


def test_set_abstraction_msg_matches_reference_module(dev, golden, weights, exact):
    from toothgroupnetwork_amd import pointnet2_utils as U
    sa = U.PointNetSetAbstractionMsg(128, [0.1, 0.2], [8, 16], 6, [[16, 24], [16, 32]]).to(dev).eval()
    sa.load_state_dict(weights["sa"])
    with torch.no_grad():
        new_xyz, feat = sa(T(golden["mod_xyz_cf"], dev), T(golden["mod_pts_cf"], dev))
    assert np.array_equal(new_xyz.cpu().numpy(), golden["mod_sa_xyz"])       # FPS indices identical -> same centres
    close(feat.cpu().numpy(), exact["mod_sa_feat_64"])


def test_set_abstraction_ssg_and_feature_propagation_match_reference_modules(dev, golden, weights, exact):
    from toothgroupnetwork_amd import pointnet2_utils as U
    ssg = U.PointNetSetAbstraction(64, 0.2, 16, 9, [16, 32], False).to(dev).eval()
    ssg.load_state_dict(weights["ssg"])
    fp = U.PointNetFeaturePropagation(62, [32, 16]).to(dev).eval()
    fp.load_state_dict(weights["fp"])
    xyz, pts = T(golden["mod_xyz_cf"], dev), T(golden["mod_pts_cf"], dev)
    with torch.no_grad():
        nx, nf = ssg(xyz, pts)
        out = fp(xyz, T(golden["mod_sa_xyz"], dev), pts, T(golden["mod_sa_feat"], dev))
    assert np.array_equal(nx.cpu().numpy(), golden["mod_ssg_xyz"])
    close(nf.cpu().numpy(), exact["mod_ssg_feat_64"])
    close(out.cpu().numpy(), exact["mod_fp_out_64"])


def test_fused_first_layer_equals_unfused_path(dev, golden, weights, monkeypatch):
    """eval-mode fused first layer (grouped tensor never built) vs the materialised path, same weights."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    xyz, pts = T(golden["mod_xyz_cf"], dev), T(golden["mod_pts_cf"], dev)
    sa = U.PointNetSetAbstractionMsg(128, [0.1, 0.2], [8, 16], 6, [[16, 24], [16, 32]]).to(dev).eval()
    sa.load_state_dict(weights["sa"])
    ssg = U.PointNetSetAbstraction(64, 0.2, 16, 9, [16, 32], False).to(dev).eval()
    ssg.load_state_dict(weights["ssg"])
    one = U.PointNetSetAbstraction(64, 0.2, 16, 9, [20], False).to(dev).eval()      # single layer: max fused too
    onem = U.PointNetSetAbstractionMsg(64, [0.2], [16], 6, [[20]]).to(dev).eval()
    with torch.no_grad():
        for m_ in (one, onem):
            for bn in [x for x in m_.modules() if isinstance(x, torch.nn.BatchNorm2d)]:
                bn.running_mean.normal_(0, 0.2)
                bn.running_var.uniform_(0.5, 2.0)
        monkeypatch.setattr(U, "_fuse_pays", lambda *a: True)       # these shapes are below the pay-off point: force it
        fused = [m(xyz, pts) for m in (sa, ssg, one, onem)]
        monkeypatch.setattr(U, "FUSED_SA", False)
        plain = [m(xyz, pts) for m in (sa, ssg, one, onem)]
    for (fx, ff), (px, pf) in zip(fused, plain):
        assert torch.equal(fx, px)
        torch.testing.assert_close(ff, pf, rtol=1e-4, atol=1e-4)
    assert fused[2][1].shape == (2, 20, 64)


def test_modules_train_step_runs_and_gradients_flow(dev):
    from toothgroupnetwork_amd import pointnet2_utils as U
    torch.manual_seed(0)
    sa = U.PointNetSetAbstractionMsg(64, [0.2, 0.4], [8, 16], 6, [[8, 16], [8, 16]]).to(dev)
    fp = U.PointNetFeaturePropagation(32 + 6, [16]).to(dev)
    pts = T(synth.scan_batch(2, 1024, "arch", 3).transpose(0, 2, 1), dev).requires_grad_()
    xyz = pts[:, :3, :].detach()
    nx, nf = sa(xyz, pts)
    out = fp(xyz, nx, pts, nf)
    out.square().mean().backward()
    assert pts.grad is not None and torch.isfinite(pts.grad).all() and pts.grad.abs().sum() > 0
    assert all(p.grad is not None for p in sa.parameters())
    # bf16 autocast (BASELINE config 3): index kernels stay fp32, module output dtype follows autocast
    with torch.autocast("cuda", dtype=torch.bfloat16):
        nx2, nf2 = sa(xyz, pts.detach())
    assert torch.equal(nx2, nx) and nf2.dtype in (torch.bfloat16, torch.float32)


def test_resample_fps_and_batched_preprocess(dev, oracle):
    from toothgroupnetwork_amd import resample
    meshes = [synth.arch_cloud(n, seed=s)[:, :6].astype(np.float64) for s, n in enumerate([5000, 7300, 6100])]
    singles = [resample.fps(m, 2000) for m in meshes]
    for m, idx in zip(meshes, singles):
        assert idx.dtype == np.int32 and idx.shape == (2000,)
        assert np.array_equal(idx, oracle.furthestsampling(m[:, :3].astype(np.float32), [m.shape[0]], [2000]))
    batched = resample.fps_batch(meshes, 2000)
    for a, b in zip(singles, batched):
        assert np.array_equal(a, b)
    out = resample.resample_pcd([meshes[0], meshes[0][:, :1]], 2000, "fps")
    assert out[0].shape == (2000, 6) and np.array_equal(out[0], meshes[0][singles[0]])
    with pytest.raises(ValueError):
        resample.fps(meshes[0][:100], 200)


def test_farthest_point_sample_np_random_start_matches_the_reference(dev, golden_r2):
    """farthest_point_sample_np (pointnet2_utils.py:103-118): with the torch RNG seeded like the golden run, the random
    starts and every index equal the reference's torch-CPU loop -- clouds with exact ties, duplicated vertices and an
    exhausted cloud included (tests/golden/make_golden_r2.py)."""
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U
    g = golden_r2
    for name in ("arch", "lattice", "dupverts"):
        xyz = g[f"fpsnp_{name}_xyz"]
        for seed, npoint in ((1, 256), (2, 64)):
            torch.manual_seed(seed)
            idx = U.farthest_point_sample_np(xyz, npoint)
            assert idx.dtype == np.int64 and np.array_equal(idx, g[f"fpsnp_{name}_{seed}_idx"]), (name, seed)
    torch.manual_seed(7)
    assert np.array_equal(U.farthest_point_sample_np(g["fpsnp_few_xyz"], 12), g["fpsnp_few_idx"])
    # the function is torch-CPU semantics by definition: the kernels' tie-order switch does not reach it
    prev = _lib.set_fps_mode(ties="tree")
    try:
        torch.manual_seed(1)
        assert np.array_equal(U.farthest_point_sample_np(g["fpsnp_lattice_xyz"], 256), g["fpsnp_lattice_1_idx"])
        assert _lib.get_fps_mode()[0] == "tree"
    finally:
        _lib.set_fps_mode(*prev)


# ------------------------------------------------------------- full size (BASELINE.json shapes)
def test_full_size_shape_a_properties(dev, oracle):
    """24 000-point scans, npoint=[4096,1024,256], nsample=32, radii [0.05,0.1,0.2]: size-independent properties on
    every scan, plus an exact oracle check of FPS, ball query and grouping at all three levels on one scan."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    scans = synth.scan_batch(3, 24000, "arch", seed=20)
    xyz = T(scans[:, :, :3], dev)
    pts = T(scans, dev)
    npoints, radii, K = [4096, 1024, 256], [0.05, 0.1, 0.2], 32
    cur_xyz, cur_pts = xyz, pts
    for lvl, (S, r) in enumerate(zip(npoints, radii)):
        fidx = U.farthest_point_sample(cur_xyz, S)
        f = fidx.cpu().numpy()
        assert (f[:, 0] == 0).all()
        for b in range(f.shape[0]):
            assert len(np.unique(f[b])) == S                       # distinct points: no index twice
        new_xyz = U.index_points(cur_xyz, fidx)
        # FPS min-distance sequence is non-increasing (defining property of farthest point sampling)
        cx = new_xyz[0].double().cpu().numpy()
        dmin = np.full(cur_xyz.shape[1], np.inf)
        allx = cur_xyz[0].double().cpu().numpy()
        prev = np.inf
        for j in range(1, 200):
            dmin = np.minimum(dmin, ((allx - cx[j - 1]) ** 2).sum(1))
            cur = dmin[f[0, j]]
            assert cur <= prev * (1 + 1e-6) and cur >= dmin.max() * (1 - 1e-5)
            prev = cur
        gidx = U.query_ball_point(r, K, cur_xyz, new_xyz)
        g = gidx.cpu().numpy()
        N = cur_xyz.shape[1]
        assert g.min() >= 0 and g.max() < N
        # rows are ascending until the padding starts, and padding repeats the first hit
        d = np.diff(g, axis=2)
        asc = d > 0
        pad = g[:, :, 1:] == g[:, :, :1]
        assert (asc | pad).all()
        # every returned index is inside the ball (expanded-form distance, fp32 threshold)
        gx = U.index_points(cur_xyz, gidx)
        q = new_xyz.unsqueeze(2)
        dot = torch.addcmul(torch.addcmul(q[..., 0] * gx[..., 0], q[..., 1], gx[..., 1]), q[..., 2], gx[..., 2])
        dist = (gx.double() - q.double()).pow(2).sum(-1)
        assert (dist <= r * r * (1 + 1e-3) + 1e-5).all()
        # the centre itself is always a member (distance ~0) -> first hit <= centre index
        assert (g[:, :, 0] <= f).all()
        grouped = U.group_points(cur_xyz, new_xyz, cur_pts, gidx, xyz_first=True)
        assert grouped.shape == (3, S, K, 3 + cur_pts.shape[2])
        # idempotence / consistency: grouped == gather - centre, recomputed with plain torch indexing
        bi = torch.arange(3, device=dev).view(3, 1, 1)
        assert torch.equal(grouped[..., :3], cur_xyz[bi, gidx] - new_xyz.unsqueeze(2))
        assert torch.equal(grouped[..., 3:], cur_pts[bi, gidx])
        # scan 0 against the CPU oracle at EVERY level (D = 6, 128, 512: the widths bench.py times): sampled
        # indices, ball-query rows and the grouped tensor, bit for bit
        cx0 = cur_xyz[:1].cpu().numpy()
        o_f = oracle.farthest_point_sample(cx0, S)
        assert np.array_equal(f[:1], o_f), lvl
        o_new = oracle.index_points(cx0, o_f)
        assert np.array_equal(new_xyz[:1].cpu().numpy(), o_new), lvl
        o_g = oracle.query_ball_point(r, K, cx0, o_new)
        assert np.array_equal(g[:1], o_g), lvl
        o_grouped = oracle.group_points(cx0, o_new, cur_pts[:1].cpu().numpy(), o_g, True)
        assert np.array_equal(grouped[:1].cpu().numpy(), o_grouped), lvl
        D_next = [128, 512, 0][lvl]
        cur_xyz = new_xyz.contiguous()
        if D_next:
            cur_pts = torch.randn(3, S, D_next, device=dev)


def test_hotpath_pipelined_equals_single_stream(dev):
    """The two-stream, double-buffered schedule of bench.py produces exactly the single-stream results."""
    from toothgroupnetwork_amd import hotpath
    shape = dict(n=6000, npoint=[1024, 256], radius=[0.1, 0.2], nsample=[32, 32], d=[6, 64])
    B = 6
    scans = synth.scan_batch(B, 6000, "arch", 77)
    pts = T(scans, dev)
    xyz = pts[:, :, :3].contiguous()
    feats = [pts, torch.randn(B, 1024, 64, device=dev)]
    ref = hotpath.HotPath(B, dev, shape=shape)
    ref.run(xyz, feats)
    torch.cuda.synchronize()
    pipe = hotpath.HotPath(B, dev, shape=shape, pipeline=True)
    for step in range(5):
        levels = pipe.run(xyz, feats)
        torch.cuda.synchronize()
        for a, b in zip(levels, ref.levels):
            for key in ("fps_idx", "new_xyz", "group_idx", "grouped"):
                assert torch.equal(a[key], b[key]), (step, key)
    levels = [pipe.run(xyz, feats) for _ in range(4)][-1]     # back-to-back, no host sync in between
    torch.cuda.synchronize()
    for a, b in zip(levels, ref.levels):
        assert torch.equal(a["grouped"], b["grouped"])


def test_full_size_knn_and_three_nn_properties(dev):
    from toothgroupnetwork_amd import pointnet2_utils as U, pointops as P
    xyz_np = synth.arch_cloud(24000, 31, False)
    xyz = T(xyz_np, dev)
    off = torch.tensor([24000], dtype=torch.int32, device=dev)
    idx, dist = P.knnquery(36, xyz, xyz, off, off)           # PT enc1 shape (24000, 24000, 36)
    assert (idx[:, 0] == torch.arange(24000, device=dev)).float().mean() > 0.999   # self is the nearest
    assert (dist[:, 1:] >= dist[:, :-1]).all()               # ascending
    e = xyz.unsqueeze(1) - xyz[idx.long()]
    d2 = (e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]
    assert torch.equal(torch.sqrt(d2), dist)                  # returned distance belongs to returned index
    # k-th distance is a true k-th smallest on a sample of queries
    sub = torch.arange(0, 24000, 480, device=dev)
    full = ((xyz[sub].unsqueeze(1) - xyz.unsqueeze(0)) ** 2)
    full = (full[..., 0] + full[..., 1]) + full[..., 2]
    kth = full.sort(1)[0][:, 35]
    assert torch.equal(torch.sqrt(kth), dist[sub, 35])
    sup = xyz[::24].contiguous().unsqueeze(0)                 # 1000 support points
    d, i = U.three_nn(xyz.unsqueeze(0), sup)
    assert (d[..., 1:] >= d[..., :-1]).all() and i.max() < 1000


def test_hotpath_multi_scale_grouping_matches_the_operators(dev):
    """Shape-B style plan (two radii per level, [features, centred xyz] order) against the drop-in operators."""
    from toothgroupnetwork_amd import hotpath, pointnet2_utils as U
    shape = dict(n=5000, npoint=[512, 128], radius=[[0.05, 0.1], [0.1, 0.2]], nsample=[[16, 32], [16, 32]], d=[6, 40],
                 xyz_first=False)
    B = 3
    pts = T(synth.scan_batch(B, 5000, "arch", 31), dev)
    xyz = pts[:, :, :3].contiguous()
    feats = [pts, torch.randn(B, 512, 40, device=dev)]
    for pipeline in (False, True):
        hp = hotpath.HotPath(B, dev, shape=shape, pipeline=pipeline)
        for _ in range(3):
            levels = hp.run(xyz, feats)
        torch.cuda.synchronize()
        cur = xyz
        for i, lv in enumerate(levels):
            fidx = U.farthest_point_sample(cur, lv["S"])
            assert torch.equal(lv["fps_idx"].long(), fidx)
            new_xyz = U.index_points(cur, fidx)
            assert torch.equal(lv["new_xyz"], new_xyz)
            for br, (r, k) in zip(lv["branches"], hotpath._branches(shape["radius"][i], shape["nsample"][i])):
                gi = U.query_ball_point(r, k, cur, new_xyz)
                assert torch.equal(br["group_idx"].long(), gi)
                assert torch.equal(br["grouped"], U.group_points(cur, new_xyz, feats[i], gi, xyz_first=False))
            cur = new_xyz


def _oracle_chain(oracle, xyz0, feats0, shape):
    """FPS -> ball query -> group of ONE scan through the CPU oracle, level by level (xyz0 (N,3), feats0[l] (N_l,D_l))."""
    out = []
    cur = xyz0[None]
    for S, r, K, f in zip(shape["npoint"], shape["radius"], shape["nsample"], feats0):
        fidx = oracle.farthest_point_sample(cur, S)
        new_xyz = oracle.index_points(cur, fidx)
        gidx = oracle.query_ball_point(r, K, cur, new_xyz)
        grouped = oracle.group_points(cur, new_xyz, f[None], gidx, shape.get("xyz_first", True))
        out.append(dict(fps_idx=fidx[0], new_xyz=new_xyz[0], group_idx=gidx[0], grouped=grouped[0]))
        cur = new_xyz
    return out


def test_pipelined_schedule_at_full_size_matches_one_stream(dev, oracle):
    """256 scans of 24 000 points (the benchmark configuration), two different resident batches, overlapping pairs of
    steps on the two streams: every result set must carry the one-stream checksums (tools/pipeline_stress.py, shorter),
    and two scans of every batch -- the first and one in the middle -- must equal the CPU ORACLE's chain bit for bit
    at all three levels (so neither schedule is only compared with the other)."""
    from toothgroupnetwork_amd import hotpath
    B = 256
    batches = []
    for s in range(2):
        pts = torch.from_numpy(synth.scan_batch(4, 24000, "arch", 700 + s)).to(dev).repeat(B // 4, 1, 1).contiguous()
        feats = [pts, torch.randn(B, 4096, 128, device=dev), torch.randn(B, 1024, 512, device=dev)]
        batches.append((pts[:, :, :3].contiguous(), feats))
    probes = (0, 133)
    want_oracle = [{b: _oracle_chain(oracle, xyz[b].cpu().numpy(), [f[b].cpu().numpy() for f in feats], hotpath.SHAPE_A)
                    for b in probes} for xyz, feats in batches]

    def check_against_oracle(levels, which, tag):
        for b in probes:
            for li, (l, o) in enumerate(zip(levels, want_oracle[which][b])):
                assert np.array_equal(l["fps_idx"][b].cpu().numpy(), o["fps_idx"]), (tag, b, li, "fps")
                assert np.array_equal(l["new_xyz"][b].cpu().numpy(), o["new_xyz"]), (tag, b, li, "new_xyz")
                assert np.array_equal(l["group_idx"][b].cpu().numpy(), o["group_idx"]), (tag, b, li, "ball")
                assert np.array_equal(l["grouped"][b].cpu().numpy(), o["grouped"]), (tag, b, li, "group")

    ref = hotpath.HotPath(B, dev)
    sums = []
    for which, (xyz, feats) in enumerate(batches):
        lv = ref.run(xyz, feats)
        torch.cuda.synchronize()
        check_against_oracle(lv, which, "one stream")
        sums.append([(int(l["fps_idx"].long().sum()), int(l["group_idx"].long().sum()), float(l["grouped"].double().sum())) for l in lv])
    del ref
    torch.cuda.empty_cache()
    hp = hotpath.HotPath(B, dev, pipeline=True)
    for pair in range(6):
        outs = []
        for step in (2 * pair, 2 * pair + 1):
            which = (step + pair) % 2
            # (odd pairs tell the second step that nothing follows it: its groupings run ungated, on the whole chip)
            outs.append((hp.run(*batches[which], inputs_on_current_stream=False, more=not (pair & 1 and step & 1)), sums[which], which))
        torch.cuda.synchronize()
        for lv, want, which in outs:
            for l, (a, b, c) in zip(lv, want):
                assert int(l["fps_idx"].long().sum()) == a and int(l["group_idx"].long().sum()) == b
                assert float(l["grouped"].double().sum()) == c
            if pair in (0, 5):            # (pair 5 ends with an ungated step)
                check_against_oracle(lv, which, f"pipelined pair {pair}")


def test_shape_b_at_full_size_matches_the_oracle(dev, oracle):
    """Shape B -- what the reference net instantiates (pointnet_pp.py:13-15: npoint [1024, 512, 256], TWO radii per level,
    nsample [32, 64], D = [6, 256, 1024], grouped layout [features, centred xyz]) -- on 24 000-point scans, the phased
    schedule and one stream, against the CPU ORACLE's chain at all three levels and both radii, bit for bit."""
    from toothgroupnetwork_amd import hotpath
    shape, B = hotpath.SHAPE_B, 8
    pts = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 910)).to(dev)
    xyz = pts[:, :, :3].contiguous()
    g = torch.Generator().manual_seed(5)
    feats = [pts, torch.randn(B, 1024, 256, generator=g).to(dev), torch.randn(B, 512, 1024, generator=g).to(dev)]
    probes = (0, 5)
    want = {}
    for b in probes:
        cur, per_level = xyz[b].cpu().numpy()[None], []
        for li, S in enumerate(shape["npoint"]):
            fidx = oracle.farthest_point_sample(cur, S)
            new_xyz = oracle.index_points(cur, fidx)
            brs = []
            for r, K in hotpath._branches(shape["radius"][li], shape["nsample"][li]):
                gidx = oracle.query_ball_point(r, K, cur, new_xyz)
                brs.append((gidx[0], oracle.group_points(cur, new_xyz, feats[li][b].cpu().numpy()[None], gidx, False)[0]))
            per_level.append((fidx[0], new_xyz[0], brs))
            cur = new_xyz
        want[b] = per_level
    for pipeline in (False, True):
        hp = hotpath.HotPath(B, dev, shape=shape, pipeline=pipeline)
        for _ in range(3):
            levels = hp.run(xyz, feats)
        torch.cuda.synchronize()
        for b in probes:
            for li, (lv, (fidx, new_xyz, brs)) in enumerate(zip(levels, want[b])):
                assert np.array_equal(lv["fps_idx"][b].cpu().numpy(), fidx), (pipeline, b, li, "fps")
                assert np.array_equal(lv["new_xyz"][b].cpu().numpy(), new_xyz), (pipeline, b, li, "new_xyz")
                assert len(lv["branches"]) == 2
                for bi, (br, (gidx, grouped)) in enumerate(zip(lv["branches"], brs)):
                    assert np.array_equal(br["group_idx"][b].cpu().numpy(), gidx), (pipeline, b, li, bi, "ball")
                    assert np.array_equal(br["grouped"][b].cpu().numpy(), grouped), (pipeline, b, li, bi, "group")
        del hp
        torch.cuda.empty_cache()


def test_feature_propagation_commuted_first_layer_equals_the_plain_form(dev, monkeypatch):
    """PointNetFeaturePropagation: the first convolution applied to the COARSE features before the interpolation (exact
    algebra, S instead of N rows) against the reference order interpolate -> concat -> convolve (pointnet2_utils.py:333-350):
    outputs and parameter / input gradients agree to rounding, in eval and in training mode."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    torch.manual_seed(3)
    B, N, S, D1, D2 = 2, 3000, 400, 6, 96
    pts = T(synth.scan_batch(B, N, "arch", 12), dev)
    xyz1 = pts[:, :, :3].permute(0, 2, 1).contiguous()
    xyz2 = xyz1[:, :, ::7][:, :, :S].contiguous()
    fp = U.PointNetFeaturePropagation(D1 + D2, [64, 32]).to(dev)
    for mode in ("eval", "train"):
        getattr(fp, mode)()
        res = {}
        for commute in (True, False):
            monkeypatch.setattr(U, "COMMUTE_FP", commute)
            p1 = torch.randn(B, D1, N, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).requires_grad_(True)
            p2 = torch.randn(B, D2, S, device=dev, generator=torch.Generator(device=dev).manual_seed(6)).requires_grad_(True)
            fp.zero_grad()
            fp.mlp_bns[0].running_mean.zero_(); fp.mlp_bns[0].running_var.fill_(1.0)
            y = fp(xyz1, xyz2, p1, p2)
            (y * torch.linspace(0.5, 1.5, y.shape[1], device=dev)[None, :, None]).sum().backward()
            res[commute] = [y.detach(), p1.grad, p2.grad, fp.mlp_convs[0].weight.grad.clone(), fp.mlp_convs[1].weight.grad.clone()]
        ya, yb = res[True][0], res[False][0]
        assert float(((ya - yb).abs() / (1.0 + yb.abs())).max()) <= 2e-5, mode                  # outputs: elementwise
        for a, b in zip(res[True][1:], res[False][1:]):                                          # gradients (BatchNorm's backward
            assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()), mode              # subtracts batch means: conditioning)


def test_linear_relu_private_epilogue_equals_the_public_fallback(dev, monkeypatch):
    """pointnet2_utils.linear_relu uses torch._addmm_activation (a private ATen entry point: the ReLU in the GEMM's epilogue) where it
    exists; without it -- a torch that dropped or renamed it -- the public F.linear + relu_ path must give the same bits"""
    from toothgroupnetwork_amd import pointnet2_utils as U
    g = torch.Generator().manual_seed(5)
    for rows, cin, cout in [(24000, 131, 128), (777, 64, 17), (8, 515, 256)]:
        x = torch.randn(2, rows, cin, generator=g).to(dev)
        W, b = torch.randn(cout, cin, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
        with torch.no_grad():
            fast = U.linear_relu(x, W, b)
            with monkeypatch.context() as m:
                if hasattr(torch, "_addmm_activation"):
                    m.delattr(torch, "_addmm_activation")
                slow = U.linear_relu(x, W, b)
            ref = torch.relu(torch.nn.functional.linear(x, W, b))
        assert fast.shape == slow.shape == ref.shape
        assert torch.equal(slow, ref)
        assert torch.equal(fast, slow), (rows, cin, cout, float((fast - slow).abs().max()))


def test_slice_columns_equals_torch_slicing(dev):
    """tgn_slice_columns (the xyz block of (N, 6) scan rows on a side stream) against torch indexing, incl. a middle block and bad arguments"""
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for rows, stride, first, ncols in [(24000 * 5 + 7, 6, 0, 3), (1000, 6, 3, 3), (77, 9, 2, 5), (0, 6, 0, 3)]:
        x = torch.randn(rows, stride, generator=g).to(dev)
        out = torch.full((rows, ncols), -7.0, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        _lib.check(L.tgn_slice_columns(rows, stride, first, ncols, _lib.ptr(x), _lib.ptr(out), _lib.c_void_p(side.cuda_stream)), "slice")
        side.synchronize()
        assert torch.equal(out, x[:, first:first + ncols])
    assert L.tgn_slice_columns(10, 6, 4, 3, _lib.ptr(x), _lib.ptr(out), _lib.stream()) != 0      # columns [4, 7) of 6
