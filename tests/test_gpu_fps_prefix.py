"""FPS of an FPS result is the identity (include/tgn_pointops.h, tgn_furthestsampling_dense_prefix): the certificate
each kernel writes, the on-device shortcut, and the tensor-identity bookkeeping of the drop-in modules.  Whatever
path is taken, the indices must be the ones the oracle computes by actually sampling."""
import os

import numpy as np
import pytest
import torch

from toothgroupnetwork_amd import synth

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _expected_certificate(seq):
    """first iteration j >= 1 whose winning distance is not in (0, 1e10), from the sampled sequence itself."""
    m = seq.shape[0]
    d = np.full(m, 1e10, np.float32)
    for j in range(1, m):
        diff = (seq - seq[j - 1]).astype(np.float32)
        dd = ((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]).astype(np.float32) + diff[:, 2] * diff[:, 2]).astype(np.float32)
        with np.errstate(invalid="ignore"):
            d = np.where(dd < d, dd, d)          # min that ignores NaN, like v_min_f32
        if not (d[j] > 0 and d[j] < np.float32(1e10)):
            return j
    return m


def _claimed(seq):
    """what a launch writes to prefix_out: the whole result when no iteration left (0, 1e10), nothing (1) otherwise."""
    return seq.shape[0] if _expected_certificate(seq) == seq.shape[0] else 1


def _fps(dev, xyz, S, cert_in=None, flags=None, want_cert=True, ref=None):
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    B, N, _ = xyz.shape
    idx = torch.full((B, S), -7, dtype=torch.int32, device=dev)
    new_xyz = torch.full((B, S, 3), -7.0, device=dev)
    cert = torch.full((B,), -7, dtype=torch.int32, device=dev) if want_cert else None
    nbytes = int(L.tgn_fps_workspace_bytes(B, N))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    _lib.check(L.tgn_furthestsampling_dense_prefix(B, N, S, _lib.ptr(xyz), _lib.ptr(ws), nbytes, _lib.ptr(idx),
                                                   _lib.ptr(new_xyz), _lib.ptr(cert_in), _lib.ptr(ref), _lib.ptr(cert),
                                                   _lib.FPS_LOCAL_INDEX if flags is None else flags, _lib.stream()))
    return idx, new_xyz, cert


@pytest.mark.parametrize("n,s1,s2,s3", [(24000, 4096, 1024, 256), (9000, 1500, 400, 64), (3000, 700, 300, 10),
                                        (100000, 24000, 4096, 1024)])
def test_chain_of_levels_matches_sampling_for_real(dev, oracle, n, s1, s2, s3):
    """every kernel family (large-cloud stream, bucket, resident) as producer and as consumer of a certificate"""
    B = 2
    xyz = np.stack([synth.arch_cloud(n, 11 + b, False) for b in range(B)])
    cur, cert = T(xyz, dev), None
    ref = xyz
    for S in (s1, s2, s3):
        idx, new_xyz, cert_out = _fps(dev, cur, S, cert_in=cert)
        want = oracle.farthest_point_sample(ref, S)
        assert np.array_equal(idx.cpu().numpy(), want)
        ref = oracle.index_points(ref, want)
        assert np.array_equal(new_xyz.cpu().numpy(), ref)
        assert (cert_out.cpu().numpy() == S).all()            # distinct finite points: the whole result carries on
        if cert is not None:
            assert np.array_equal(want, np.tile(np.arange(S), (B, 1)))   # ... and the consumer's answer IS the identity
        cur, cert = new_xyz, cert_out


def test_certificate_value_and_fallback_on_degenerate_clouds(dev, oracle):
    rng = np.random.default_rng(3)
    base = synth.uniform_cloud(40, 1)
    clouds = [np.repeat(base, 3, axis=0),                                   # 40 distinct points, 120 rows: exhausted at 40
              (rng.integers(-2, 3, size=(300, 3)) / 4).astype(np.float32),  # 125 lattice sites
              synth.uniform_cloud(300, 5)]
    nanc = synth.uniform_cloud(300, 6)
    nanc[17] = np.nan                                                       # never updated: picked again and again
    clouds.append(nanc)
    for c in clouds:
        c = c[:120] if c.shape[0] < 300 else c
        x = c[None].astype(np.float32)
        S1 = 100
        idx, new_xyz, cert = _fps(dev, T(x, dev), S1)
        want = oracle.farthest_point_sample(x, S1)
        assert np.array_equal(idx.cpu().numpy(), want)
        seq = oracle.index_points(x, want)[0]
        assert int(cert[0]) == _claimed(seq)
        for S2 in (5, 30, 64, 100):
            idx2, new2, cert2 = _fps(dev, new_xyz, S2, cert_in=cert)
            want2 = oracle.farthest_point_sample(seq[None], S2)
            assert np.array_equal(idx2.cpu().numpy(), want2), (S2, int(cert[0]))   # shortcut or fallback: same answer
            assert np.array_equal(new2.cpu().numpy(), oracle.index_points(seq[None], want2), equal_nan=True)
            assert int(cert2[0]) == _claimed(new2[0].cpu().numpy())


def test_tree_tie_order_ignores_the_certificate(dev, oracle):
    from toothgroupnetwork_amd import _lib
    x = (np.random.default_rng(0).integers(-6, 7, size=(1, 3000, 3)) / 8).astype(np.float32)
    fl = _lib.FPS_LOCAL_INDEX | _lib.FPS_CUDA_COMPAT
    idx, new_xyz, _ = _fps(dev, T(x, dev), 512, flags=fl)
    lying = torch.full((1,), 512, dtype=torch.int32, device=dev)           # even a (wrong) certificate is not used
    idx2, _, _ = _fps(dev, new_xyz, 200, cert_in=lying, flags=fl)
    want = oracle.farthest_point_sample(oracle.index_points(x, oracle.farthest_point_sample(x, 512, mode=3)), 200, mode=3)
    assert np.array_equal(idx2.cpu().numpy(), want)
    assert not np.array_equal(want[0], np.arange(200))                     # and here the identity would have been wrong


def test_modules_hand_the_certificate_on_and_results_do_not_change(dev, monkeypatch):
    from toothgroupnetwork_amd import pointnet2_utils as U
    torch.manual_seed(0)
    B, N = 2, 6000
    pts = T(synth.scan_batch(B, N, "arch", 5), dev)
    xyz_cf, feat_cf = pts[:, :, :3].permute(0, 2, 1).contiguous(), pts.permute(0, 2, 1).contiguous()
    sa1 = U.PointNetSetAbstractionMsg(1024, [0.05, 0.1], [16, 32], 6, [[16, 32], [16, 32]]).to(dev).eval()
    sa2 = U.PointNetSetAbstractionMsg(256, [0.1, 0.2], [16, 32], 64, [[32, 64], [32, 64]]).to(dev).eval()
    sa3 = U.PointNetSetAbstraction(64, 0.4, 16, 128 + 3, [64, 128], False).to(dev).eval()

    def run():
        with torch.no_grad():
            x1, f1 = sa1(xyz_cf, feat_cf)
            x2, f2 = sa2(x1, f1)
            x3, f3 = sa3(x2, f2)
        return [t.clone() for t in (x1, f1, x2, f2, x3, f3)]

    monkeypatch.setattr(U, "FPS_PREFIX", False)
    U.fps_prefix_clear()
    plain = run()
    monkeypatch.setattr(U, "FPS_PREFIX", True)
    U.fps_prefix_stats["offered"] = 0
    fast = run()
    assert U.fps_prefix_stats["offered"] == 2              # levels 2 and 3 were offered a certificate
    for a, b in zip(plain, fast):
        assert torch.equal(a, b)
    # the reference's own idiom, new_xyz = index_points(xyz, farthest_point_sample(xyz, S)), passes it on as well
    U.fps_prefix_clear()
    U.fps_prefix_stats["offered"] = 0
    xyz = pts[:, :, :3].contiguous()
    nx1, _ = U.sample_and_group(800, 0.1, 8, xyz, None)
    nx2, _ = U.sample_and_group(200, 0.2, 8, nx1, None)
    assert U.fps_prefix_stats["offered"] == 1
    assert torch.equal(U.farthest_point_sample(nx1, 200)[0], torch.arange(200, device=dev))
    monkeypatch.setattr(U, "FPS_PREFIX", False)
    px1, _ = U.sample_and_group(800, 0.1, 8, xyz, None)
    px2, _ = U.sample_and_group(200, 0.2, 8, px1, None)
    assert torch.equal(nx2, px2)
    # provenance is by CONTENT: a tensor modified after sampling is offered the certificate, fails the on-device
    # comparison and is sampled for real; an untouched copy passes
    monkeypatch.setattr(U, "FPS_PREFIX", True)
    U.fps_prefix_clear()
    nx1, _ = U.sample_and_group(800, 0.1, 8, xyz, None)
    moved = nx1 + torch.randn_like(nx1) * 0.05
    got = U.farthest_point_sample(moved, 200)
    U.fps_prefix_clear()
    monkeypatch.setattr(U, "FPS_PREFIX", False)
    assert torch.equal(got, U.farthest_point_sample(moved, 200))
    assert not torch.equal(got[0], torch.arange(200, device=dev))


def test_dense_prefix_ref_is_checked_per_cloud(dev, oracle):
    """one cloud of the batch altered: that cloud runs, the other takes the shortcut, both match the oracle"""
    x = np.stack([synth.arch_cloud(5000, 21, False), synth.arch_cloud(5000, 22, False)])
    idx, seq, cert = _fps(dev, T(x, dev), 1000)
    altered = seq.clone()
    altered[1, 10:20] = altered[1, 10:20].flip(0)          # same points, different order: no longer the sequence
    idx2, _, cert2 = _fps(dev, altered, 300, cert_in=cert, ref=seq)
    want = oracle.farthest_point_sample(altered.cpu().numpy(), 300)
    assert np.array_equal(idx2.cpu().numpy(), want)
    assert np.array_equal(want[0], np.arange(300)) and not np.array_equal(want[1], np.arange(300))


def test_hotpath_with_prefix_certificates_equals_plain(dev):
    from toothgroupnetwork_amd import hotpath
    shape = dict(n=6000, npoint=[1024, 256, 64], radius=[0.1, 0.2, 0.4], nsample=[32, 32, 16], d=[6, 64, 32])
    B = 4
    pts = T(synth.scan_batch(B, 6000, "arch", 9), dev)
    xyz = pts[:, :, :3].contiguous()
    feats = [pts, torch.randn(B, 1024, 64, device=dev), torch.randn(B, 256, 32, device=dev)]
    ref = hotpath.HotPath(B, dev, shape=shape)
    ref.run(xyz, feats)
    for pipeline in (False, True):
        hp = hotpath.HotPath(B, dev, shape=shape, pipeline=pipeline, fps_prefix=True)
        for _ in range(3):
            levels = hp.run(xyz, feats)
        torch.cuda.synchronize()
        for a, b in zip(levels, ref.levels):
            for key in ("fps_idx", "new_xyz", "group_idx", "grouped"):
                assert torch.equal(a[key], b[key]), (pipeline, key)
        assert all(int(lv["cert"].min()) == lv["S"] for lv in levels)


def test_packed_transition_down_chain_with_content_checked_certificates(dev, oracle, monkeypatch):
    """the Point-Transformer idiom (blocks.py:62-74): idx = furthestsampling(p, o, n_o); n_p = p[idx]; next level
    samples n_p.  Provenance is checked by content on the device; ragged batch with one degenerate cloud (falls back
    inside the same launch) and a perturbed copy (must not be believed)."""
    from toothgroupnetwork_amd import pointops as P
    monkeypatch.setattr(P, "FPS_PREFIX", True)      # the plain operator takes part only when asked to (TGN_FPS_PREFIX=1)
    P.fps_prefix_clear()
    P.fps_prefix_stats["offered"] = 0
    dup = np.repeat(synth.uniform_cloud(150, 4), 8, axis=0)                  # 150 distinct points in 1200 rows
    clouds = [synth.arch_cloud(9000, 1, False), dup, synth.uniform_cloud(3000, 2)]
    xyz = np.concatenate(clouds).astype(np.float32)
    off = np.cumsum([c.shape[0] for c in clouds]).astype(np.int32)
    p, o = T(xyz, dev), T(off, dev)
    p_np, o_np = xyz, off
    for level in range(4):
        sizes = np.diff(np.concatenate([[0], o_np]))
        n_o_np = np.cumsum(sizes // 4).astype(np.int32)
        n_o = T(n_o_np, dev)
        idx = P.furthestsampling(p, o, n_o)
        want = oracle.furthestsampling(p_np, o_np, n_o_np)
        assert np.array_equal(idx.cpu().numpy(), want), level
        n_p = p[idx.long(), :]                                               # the model's own gather: a new tensor
        p, o, p_np, o_np = n_p, n_o, p_np[want.astype(np.int64)], n_o_np
    assert P.fps_prefix_stats["offered"] == 3
    # same layout, different coordinates: the kernel must notice and sample for real
    q = (p * 1.5).contiguous()
    n_o2 = T(np.cumsum(np.diff(np.concatenate([[0], o_np])) // 2).astype(np.int32), dev)
    got = P.furthestsampling(q, o, n_o2)
    assert np.array_equal(got.cpu().numpy(), oracle.furthestsampling(q.cpu().numpy(), o_np, n_o2.cpu().numpy()))
    # TGN_FPS_PREFIX off: identical indices
    monkeypatch.setattr(P, "FPS_PREFIX", False)
    off_idx = P.furthestsampling(T(xyz, dev), T(off, dev), T(np.cumsum([c.shape[0] // 4 for c in clouds]).astype(np.int32), dev))
    assert np.array_equal(off_idx.cpu().numpy(), oracle.furthestsampling(xyz, off, np.cumsum([c.shape[0] // 4 for c in clouds]).astype(np.int32)))


def test_shortcut_is_opt_in_and_survives_raw_writes_into_the_callers_tensor(dev, oracle, monkeypatch):
    """(1) The plain operators never take the shortcut by themselves (only call sites that chain levels ask for it).
    (2) The book compares against a PRIVATE copy: a write into the caller's new_xyz that no version counter sees
    (`.data`) must fail the content check -- the stale certificate is not believed."""
    from toothgroupnetwork_amd import pointnet2_utils as U, pointops as P
    assert U.FPS_PREFIX is None and P.FPS_PREFIX is None or os.environ.get("TGN_FPS_PREFIX") is not None
    U.fps_prefix_clear()
    U.fps_prefix_stats["offered"] = 0
    xyz = T(synth.scan_batch(2, 5000, "arch", 77)[:, :, :3].copy(), dev)
    nx = U.index_points(xyz, U.farthest_point_sample(xyz, 700))
    U.farthest_point_sample(nx, 100)
    if U.FPS_PREFIX is None:
        assert U.fps_prefix_stats["offered"] == 0
    monkeypatch.setattr(U, "FPS_PREFIX", True)
    U.fps_prefix_clear()
    _, nx = U._fps_dense(xyz, 700, want_coords=True)
    nx.data[1, 10:20] = nx.data[1, 10:20].flip(0)          # in place, version counter untouched
    got = U.farthest_point_sample(nx, 300)
    want = oracle.farthest_point_sample(nx.cpu().numpy(), 300)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(want[0], np.arange(300)) and not np.array_equal(want[1], np.arange(300))
