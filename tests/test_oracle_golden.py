"""CPU: the oracle (oracle/pointops_oracle.c) against the golden fixtures produced by the reference's
own torch functions, and against independent brute-force definitions for the CUDA-only operators."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from toothgroupnetwork_amd import synth


def test_fps_matches_reference_cpu(oracle, golden):
    for k in ("arch", "uniform", "lattice"):
        xyz = golden[f"fps_{k}_xyz"]
        ref = golden[f"fps_{k}_idx"]
        got = oracle.farthest_point_sample(xyz, ref.shape[1])
        assert np.array_equal(got, ref.astype(np.int64)), k


def test_square_distance_matches_reference_cpu(oracle, golden):
    assert np.array_equal(oracle.square_distance(golden["sqd_src"], golden["sqd_dst"]), golden["sqd_out"])


def test_ball_query_matches_reference_cpu(oracle, golden):
    for ri in range(4):
        radius, ns = golden[f"ball_{ri}_cfg"]
        got = oracle.query_ball_point(float(radius), int(ns), golden["ball_xyz"], golden["ball_new_xyz"])
        assert np.array_equal(got, golden[f"ball_{ri}_idx"].astype(np.int64)), radius


def test_sample_and_group_matches_reference_cpu(oracle, golden):
    xyz = golden["ball_xyz"]
    nx, npts, fidx, gidx = oracle.sample_and_group(128, 0.1, 16, xyz, golden["sag_points"], xyz_first=True)
    assert np.array_equal(nx, golden["sag_new_xyz"])
    assert np.array_equal(npts, golden["sag_new_points"])


def test_three_nn_interpolate_matches_reference_cpu(oracle, golden):
    d, i = oracle.three_nn(golden["tnn_xyz1"], golden["tnn_xyz2"])
    assert np.array_equal(d, golden["tnn_dist"])
    assert np.array_equal(i, golden["tnn_idx"].astype(np.int64))
    out = oracle.three_interpolate(golden["tnn_feat2"], d, i)
    np.testing.assert_allclose(out, golden["tnn_interp"], rtol=0, atol=1e-5)


def test_regression_vectors_stable(oracle, regression):
    r = regression
    assert np.array_equal(oracle.furthestsampling(r["p_xyz"], r["p_offset"], r["p_new_offset"]), r["p_fps_idx"])
    assert np.array_equal(oracle.furthestsampling(r["p_xyz"], r["p_offset"], r["p_new_offset"], mode=3),
                          r["p_fps_idx_cudacompat"])
    assert np.array_equal(oracle.furthestsampling(r["p_xyz"], r["p_offset"], r["p_new_offset"], mode=2),
                          r["p_fps_idx_tree"])
    q = r["p_xyz"][r["p_fps_idx"].astype(np.int64)]
    idx, dist = oracle.knnquery(16, r["p_xyz"], q, r["p_offset"], r["p_new_offset"])
    assert np.array_equal(idx, r["p_knn_idx"]) and np.array_equal(dist, r["p_knn_dist"])


# ---- independent definitions -------------------------------------------------------------------
def _fps_numpy(xyz, m):
    """vectorised numpy FPS, torch-CPU arithmetic: ((dx*dx)+(dy*dy))+(dz*dz), first max."""
    n = xyz.shape[0]
    d = np.full(n, 1e10, dtype=np.float32)
    out = np.zeros(m, dtype=np.int64)
    cur = 0
    for j in range(1, m):
        diff = xyz - xyz[cur]
        dd = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        d = np.minimum(d, dd)
        cur = int(np.argmax(d))
        out[j] = cur
    return out


@pytest.mark.parametrize("kind,n,m", [("uniform", 777, 200), ("arch", 1500, 300), ("lattice", 0, 150)])
def test_fps_vs_numpy_definition(oracle, kind, n, m):
    xyz = {"uniform": lambda: synth.uniform_cloud(n, 3), "arch": lambda: synth.arch_cloud(n, 4, False),
           "lattice": lambda: synth.lattice_cloud(7, dup=30, seed=5)}[kind]()
    got = oracle.farthest_point_sample(xyz[None], m)[0]
    assert np.array_equal(got, _fps_numpy(xyz, m))


def test_fps_packed_offsets_and_edge_cases(oracle):
    a, b, c = synth.uniform_cloud(50, 1), synth.uniform_cloud(1, 2), synth.uniform_cloud(300, 3)
    xyz = np.concatenate([a, b, c])
    offset = np.array([50, 51, 351], np.int32)
    new_offset = np.array([20, 21, 121], np.int32)
    idx = oracle.furthestsampling(xyz, offset, new_offset)
    assert np.array_equal(idx[:20], _fps_numpy(a, 20))
    assert idx[20] == 50
    assert np.array_equal(idx[21:], _fps_numpy(c, 100) + 51)
    # more samples than points: remaining picks have distance 0 -> first index of the cloud
    idx2 = oracle.furthestsampling(a[:5], np.array([5], np.int32), np.array([9], np.int32))
    assert sorted(idx2[:5].tolist()) == [0, 1, 2, 3, 4] and (idx2[5:] == 0).all()
    # empty batch
    assert oracle.furthestsampling(np.zeros((0, 3), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32)).size == 0


def test_fps_cuda_compat_differs_only_by_ties_and_rounding(oracle):
    xyz = synth.uniform_cloud(2000, 11)
    off, noff = np.array([2000], np.int32), np.array([64], np.int32)
    a = oracle.furthestsampling(xyz, off, noff, mode=0)
    b = oracle.furthestsampling(xyz, off, noff, mode=3)
    assert a[0] == b[0] == 0
    # both are valid FPS sequences: every pick is (numerically) a farthest point
    for seq in (a, b):
        d = np.full(2000, np.inf)
        for j in range(1, 64):
            d = np.minimum(d, ((xyz - xyz[seq[j - 1]]) ** 2).sum(1))
            assert d[seq[j]] >= d.max() * (1 - 1e-5)


def _knn_bruteforce(k, xyz, q, off, noff):
    idx = np.zeros((q.shape[0], k), np.int32)
    d2 = np.zeros((q.shape[0], k), np.float32)
    starts = np.concatenate([[0], off[:-1]])
    qstarts = np.concatenate([[0], noff[:-1]])
    for b in range(len(off)):
        pts = xyz[starts[b]:off[b]]
        for qi in range(qstarts[b], noff[b]):
            e = q[qi] - pts
            d = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]
            order = np.lexsort((np.arange(len(d)), d))[:k]
            idx[qi, :len(order)] = order + starts[b]
            d2[qi, :len(order)] = d[order]
            idx[qi, len(order):] = starts[b]
            d2[qi, len(order):] = 1e10
    return idx, d2


def test_knn_vs_bruteforce(oracle):
    xyz = np.concatenate([synth.uniform_cloud(400, 1), synth.arch_cloud(10, 2, False)])
    off = np.array([400, 410], np.int32)
    q = np.concatenate([synth.uniform_cloud(30, 3), synth.arch_cloud(7, 4, False)])
    noff = np.array([30, 37], np.int32)
    for k in (1, 3, 16):
        idx, dist = oracle.knnquery(k, xyz, q, off, noff)
        bi, bd = _knn_bruteforce(k, xyz, q, off, noff)
        # random data: distances are distinct, so the heap order is the sorted order
        assert np.array_equal(idx, bi), k
        assert np.array_equal(dist, np.sqrt(bd)), k
    # segment smaller than k -> tail is (start, 1e5)
    idx, dist = oracle.knnquery(16, xyz, q, off, noff)
    assert (idx[30:, 10:] == 400).all() and (dist[30:, 10:] == np.float32(1e5)).all()


def test_knn_ties_distance_multiset(oracle):
    """With exact distance ties the heap's insertion history decides WHICH of the tied points survive and
    in what order (knnquery_cuda_kernel.cu:21-48), so only the sorted distances are history-independent."""
    xyz = synth.lattice_cloud(5, dup=10, seed=1)
    off = np.array([xyz.shape[0]], np.int32)
    idx, dist = oracle.knnquery(8, xyz, xyz, off, off)
    bi, bd = _knn_bruteforce(8, xyz, xyz, off, off)
    assert np.array_equal(dist, np.sqrt(bd))
    e = xyz[:, None, :] - xyz[idx.astype(np.int64)]
    d2 = (e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]
    assert np.array_equal(np.sqrt(d2), dist)          # every returned index has the returned distance
    assert all(len(set(r)) == 8 for r in idx.tolist())  # no index twice


def test_gather_family_vs_numpy(oracle):
    rng = np.random.default_rng(0)
    n, m, ns, c, wc = 50, 20, 6, 8, 4
    feat = rng.normal(size=(n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(m, ns)).astype(np.int32)
    assert np.array_equal(oracle.grouping_forward(feat, idx), feat[idx])
    go = rng.normal(size=(m, ns, c)).astype(np.float32)
    gi = np.zeros((n, c), np.float64)
    np.add.at(gi, idx.reshape(-1), go.reshape(-1, c))
    np.testing.assert_allclose(oracle.grouping_backward(go, idx, n), gi, atol=1e-5)
    idn = rng.integers(0, n, size=(n, ns)).astype(np.int32)
    f2 = rng.normal(size=(n, c)).astype(np.float32)
    assert np.array_equal(oracle.subtraction_forward(feat, f2, idn), feat[:, None, :] - f2[idn])
    pos = rng.normal(size=(n, ns, c)).astype(np.float32)
    w = rng.normal(size=(n, ns, wc)).astype(np.float32)
    ref = ((feat[idn] + pos) * np.tile(w, (1, 1, c // wc))).sum(1)
    np.testing.assert_allclose(oracle.aggregation_forward(feat, pos, w, idn), ref, atol=1e-4)
    k = 3
    ik = rng.integers(0, n, size=(m, k)).astype(np.int32)
    wk = rng.random((m, k)).astype(np.float32)
    np.testing.assert_allclose(oracle.interpolation_forward(feat, ik, wk), (feat[ik] * wk[..., None]).sum(1), atol=1e-5)


def test_ball_query_semantics(oracle):
    xyz = synth.arch_cloud(2000, 3, False)[None]
    q = xyz[:, :50]
    out = oracle.query_ball_point(0.1, 16, xyz, q)
    d = oracle.square_distance(q, xyz)[0]
    r2 = oracle.radius_sq_f32(0.1)
    for s in range(50):
        hits = np.nonzero(~(d[s] > r2))[0]
        exp = list(hits[:16]) + [hits[0]] * max(0, 16 - len(hits))
        assert out[0, s].tolist() == exp
    # no hit at all -> N
    far = np.full((1, 1, 3), 50.0, np.float32)
    assert (oracle.query_ball_point(0.1, 4, xyz, far) == 2000).all()
    with pytest.raises(IndexError):
        oracle.group_points(xyz, far, None, oracle.query_ball_point(0.1, 4, xyz, far))


def test_fps_of_an_fps_result_is_the_identity(oracle):
    """The property behind tgn_furthestsampling_dense_prefix (include/tgn_pointops.h): farthest point sampling of a
    cloud that is itself an FPS sequence returns 0, 1, 2, ... -- with exact ties too (first-index order), as long as
    the producing run never picked a point at distance 0 (exhausted cloud); the tree tie order does not have it."""
    from toothgroupnetwork_amd import synth
    rng = np.random.default_rng(5)
    clouds = {"arch": synth.arch_cloud(5000, 1, False), "uniform": synth.uniform_cloud(4000, 2),
              "lattice": synth.lattice_cloud(8, dup=3, seed=0),
              "quantised": (rng.integers(-6, 7, size=(3000, 3)) / 8).astype(np.float32)}
    for name, c in clouds.items():
        x = c[None].astype(np.float32)
        s1 = min(800, x.shape[1])
        seq = oracle.index_points(x, oracle.farthest_point_sample(x, s1))
        distinct = len(np.unique(c, axis=0))
        s2 = min(256, distinct)
        assert np.array_equal(oracle.farthest_point_sample(seq, s2)[0], np.arange(s2)), name
    # exhausted: once every distinct point is taken the reference keeps returning point 0 -- NOT the identity
    c = np.repeat(synth.uniform_cloud(5, 3), 3, axis=0)[None]
    seq = oracle.index_points(c, oracle.farthest_point_sample(c, 12))
    again = oracle.farthest_point_sample(seq, 12)[0]
    assert np.array_equal(again[:5], np.arange(5)) and not np.array_equal(again, np.arange(12))


def test_inference_pipeline_golden_from_the_oracle_side(oracle, tmp_path):
    """tests/golden/reference_cpu_r3_pipeline.npz (the reference's InferencePipeLine.__call__ run on CPU) restated on the CPU from the
    oracle's pieces -- OBJ reader, normals, FPS -- the pipeline's host arithmetic (normalisation, relabelling) and scipy's float64
    nearest sample: the same label on every vertex.  (The GPU test replays it through the kernels.)"""
    import torch
    sys.path.insert(0, GOLDEN)
    from oracle import meshio as OM
    from pipeline_model import MESH, fixed_model
    from toothgroupnetwork_amd import inference, synth
    gold = np.load(os.path.join(GOLDEN, "reference_cpu_r3_pipeline.npz"))
    path = tmp_path / "scan.obj"
    path.write_text(synth.obj_text(MESH[0], MESH[1], MESH[2], "plain", with_tail=False))
    v, f = OM.read_obj(str(path))
    org = np.concatenate([inference.normalise_for_inference(v), OM.vertex_normals(v, f - 1)], axis=1)
    idx = oracle.furthestsampling(np.ascontiguousarray(org[:, :3], dtype=np.float32), [org.shape[0]], [24000]).reshape(-1)
    sampled = org[idx]
    inp = torch.from_numpy(sampled.astype("float32")[None]).permute(0, 2, 1)
    cls = fixed_model([inp])["cls_pred"].argmax(dim=1).reshape(-1).numpy()
    labels = inference.fdi_from_classes(cls)
    from scipy.spatial import cKDTree
    near = cKDTree(sampled[:, :3]).query(org[:, :3], k=1)[1]                   # float64 nearest sample (scipy, not the reference's sklearn)
    assert np.array_equal(labels[near], gold["sem"].astype(np.int64))


def test_torch_cpu_restatement_matches_reference_cpu(golden):
    """oracle/torch_cpu.py -- the reference's CPU path restated torch call for torch call (it is what bench.py times as
    `cpu_baseline` on the bench host, where the reference checkout does not exist) -- against the fixtures the reference's own
    functions produced: FPS indices, square_distance bits, ball-query indices, the grouped tensor of sample_and_group."""
    import torch

    from oracle import torch_cpu as TC
    for k in ("arch", "uniform", "lattice"):
        got = TC.fps(torch.from_numpy(golden[f"fps_{k}_xyz"]), golden[f"fps_{k}_idx"].shape[1])
        assert np.array_equal(got.numpy(), golden[f"fps_{k}_idx"].astype(np.int64)), k
    assert np.array_equal(TC.square_distance(torch.from_numpy(golden["sqd_src"]), torch.from_numpy(golden["sqd_dst"])).numpy(), golden["sqd_out"])
    xyz, new_xyz = torch.from_numpy(golden["ball_xyz"]), torch.from_numpy(golden["ball_new_xyz"])
    for ri in range(4):
        radius, ns = golden[f"ball_{ri}_cfg"]
        got = TC.ball_query(float(radius), int(ns), xyz, new_xyz)
        assert np.array_equal(got.numpy(), golden[f"ball_{ri}_idx"].astype(np.int64)), radius
    nx = TC.index_points(xyz, TC.fps(xyz, 128))
    assert np.array_equal(nx.numpy(), golden["sag_new_xyz"])
    grouped = TC.group(xyz, nx, torch.from_numpy(golden["sag_points"]), TC.ball_query(0.1, 16, xyz, nx))
    assert np.array_equal(grouped.numpy(), golden["sag_new_points"])
    # the whole headline loop at a small size: shapes and the per-level timing triple
    from toothgroupnetwork_amd import synth
    t = TC.headline_levels(synth.arch_cloud(1500, 3), [256, 64], [0.1, 0.2], [8, 8], [6, 16])
    assert len(t) == 2 and all(len(x) == 3 and min(x) >= 0 for x in t)
