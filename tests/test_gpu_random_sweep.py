"""Seeded random sweeps of the sampling / neighbour-search kernels against the oracle: ragged packed batches with empty,
one-point and over-sampled segments, coordinates quantised so that exact distance ties are common, every launch shape
the dispatcher can pick for the drawn sizes.  Sizes stay small enough for the C oracle to finish in seconds."""
import numpy as np
import pytest
import torch

from toothgroupnetwork_amd import synth

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(rng, n, style):
    if style == 0:      # uniform
        return rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    if style == 1:      # quantised: many exact ties and duplicates
        return (rng.integers(-6, 7, size=(n, 3)) / 8.0).astype(np.float32)
    if style == 2:      # thin sheet with clusters (ball queries see crowded and empty neighbourhoods)
        c = rng.uniform(-1, 1, size=(max(n // 40, 1), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(scale=0.03, size=(n, 3))
        p[:, 2] *= 0.05
        return p.astype(np.float32)
    return np.full((n, 3), 0.25, np.float32)  # all points identical


@pytest.mark.parametrize("seed", range(24))
def test_fps_random_ragged_batches(dev, oracle, seed):
    from toothgroupnetwork_amd import pointops as P
    rng = np.random.default_rng(1000 + seed)
    b = int(rng.integers(1, 7))
    sizes = [int(rng.choice([0, 1, 2, 63, 64, 65, 500, 1500, 5000, 9000, 12000])) for _ in range(b)]
    if seed % 4 == 0:
        sizes[0] = int(rng.integers(8192, 20000))     # bucket-skipping kernel territory
    ms = [int(min(max(rng.integers(0, 3) * n // 4 + rng.integers(0, 3), 0), 3000)) if n else 0 for n in sizes]
    ms = [m if n else 0 for m, n in zip(ms, sizes)]
    xyz = np.concatenate([_cloud(rng, n, int(rng.integers(0, 4))) for n in sizes] + [np.zeros((0, 3), np.float32)])
    off = np.cumsum(sizes).astype(np.int32)
    noff = np.cumsum(ms).astype(np.int32)
    got = P.furthestsampling(T(xyz, dev), T(off, dev), T(noff, dev)).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (int(noff[-1]),)
    assert np.array_equal(got, oracle.furthestsampling(xyz, off, noff)), (sizes, ms)
    got_cc, _ = P.fps_with_coords(T(xyz, dev), T(off, dev), T(noff, dev), cuda_compat=True)
    assert np.array_equal(got_cc.cpu().numpy(), oracle.furthestsampling(xyz, off, noff, mode=3)), (sizes, ms)


@pytest.mark.parametrize("seed", range(24))
def test_ball_query_random(dev, oracle, seed):
    from toothgroupnetwork_amd import pointnet2_utils as U
    rng = np.random.default_rng(2000 + seed)
    B = int(rng.integers(1, 4))
    N = int(rng.choice([1, 7, 64, 300, 2047, 2048, 4096, 6000]))
    S = int(rng.choice([1, 5, 64, 257, 700]))
    K = int(rng.choice([1, 3, 16, 32, 64, 100]))
    style = int(rng.integers(0, 4))
    xyz = np.stack([_cloud(rng, N, style) for _ in range(B)])
    q = np.stack([np.concatenate([xyz[b][rng.integers(0, N, S // 2 + 1)], _cloud(rng, S, 0)])[:S] for b in range(B)])
    for radius in (float(rng.choice([0.0, 0.05, 0.125, 0.25, 0.5])), float(rng.uniform(0.01, 1.5)), 10.0):
        got = U.query_ball_point(radius, K, T(xyz, dev), T(q, dev)).cpu().numpy()
        assert np.array_equal(got, oracle.query_ball_point(radius, K, xyz, q)), (B, N, S, K, style, radius)


@pytest.mark.parametrize("seed", range(24))
def test_knn_random_ragged(dev, oracle, seed):
    from toothgroupnetwork_amd import pointops as P
    P.knn_cache_clear()
    rng = np.random.default_rng(3000 + seed)
    b = int(rng.integers(1, 5))
    sizes = [int(rng.choice([1, 2, 17, 64, 200, 1000, 3000])) for _ in range(b)]
    qs = [int(rng.choice([1, 3, 64, 300])) for _ in range(b)]
    k = int(rng.choice([1, 3, 8, 16, 36, 64]))
    style = int(rng.integers(0, 3))
    xyz = np.concatenate([_cloud(rng, n, style) for n in sizes])
    q = np.concatenate([_cloud(rng, m, style) for m in qs])
    off, noff = np.cumsum(sizes).astype(np.int32), np.cumsum(qs).astype(np.int32)
    idx, dist = P.knnquery(k, T(xyz, dev), T(q, dev), T(off, dev), T(noff, dev))
    oi, od = oracle.knnquery(k, xyz, q, off, noff)
    assert np.array_equal(idx.cpu().numpy(), oi), (sizes, qs, k, style)   # exact ties: the heap's order
    assert np.array_equal(dist.cpu().numpy(), od)


@pytest.mark.parametrize("seed", range(6))
def test_three_nn_random(dev, oracle, seed):
    from toothgroupnetwork_amd import pointnet2_utils as U
    rng = np.random.default_rng(4000 + seed)
    B, N, S = int(rng.integers(1, 3)), int(rng.choice([5, 100, 1500])), int(rng.choice([3, 4, 64, 500]))
    xyz1 = np.stack([_cloud(rng, N, int(rng.integers(0, 3))) for _ in range(B)])
    xyz2 = np.stack([_cloud(rng, S, int(rng.integers(0, 3))) for _ in range(B)])
    d, i = U.three_nn(T(xyz1, dev), T(xyz2, dev))
    od, oi = oracle.three_nn(xyz1, xyz2)
    assert np.array_equal(d.cpu().numpy(), od)
    assert np.array_equal(i.cpu().numpy(), oi)


@pytest.mark.parametrize("seed", range(12))
def test_gather_family_random_shapes(dev, oracle, seed):
    """grouping / subtraction / aggregation / interpolation2 forward and backward at odd channel counts, one-neighbour
    and wide-neighbour cases (the kernels tile rows x channel lanes: partial tiles everywhere)."""
    from toothgroupnetwork_amd import pointops as P
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([1, 2, 33, 250, 1031]))
    m = int(rng.choice([1, 7, 64, 300]))
    ns = int(rng.choice([1, 2, 3, 8, 17, 36]))
    wc = int(rng.choice([1, 2, 4, 8]))
    c = wc * int(rng.choice([1, 3, 5, 16, 33]))
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(dev)
    npy = lambda t: t.detach().cpu().numpy()

    def close(got, want):
        # backward passes accumulate with fp32 atomics in arbitrary order (up to m*ns terms into one row when n is
        # tiny): the tolerance follows the magnitude of the sums
        tol = 2e-5 * max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, atol=tol, rtol=1e-4)
    # grouping: (n, c) rows gathered by (m, ns)
    feat = f(n, c).requires_grad_()
    idx = torch.from_numpy(rng.integers(0, n, (m, ns)).astype(np.int32)).to(dev)
    out = P.grouping(feat, idx)
    assert np.array_equal(npy(out), oracle.grouping_forward(npy(feat), npy(idx)))
    go = f(m, ns, c)
    out.backward(go)
    close(npy(feat.grad), oracle.grouping_backward(npy(go), npy(idx), n))
    # subtraction / aggregation: idx is (n, ns) over the same n rows
    idx2 = torch.from_numpy(rng.integers(0, n, (n, ns)).astype(np.int32)).to(dev)
    a, b = f(n, c).requires_grad_(), f(n, c).requires_grad_()
    out = P.subtraction(a, b, idx2)
    assert np.array_equal(npy(out), oracle.subtraction_forward(npy(a), npy(b), npy(idx2)))
    go = f(n, ns, c)
    out.backward(go)
    g1, g2 = oracle.subtraction_backward(npy(idx2), npy(go))
    close(npy(a.grad), g1)
    close(npy(b.grad), g2)
    x, pos, w = f(n, c).requires_grad_(), f(n, ns, c).requires_grad_(), f(n, ns, wc).requires_grad_()
    out = P.aggregation(x, pos, w, idx2)
    close(npy(out), oracle.aggregation_forward(npy(x), npy(pos), npy(w), npy(idx2)))
    go = f(n, c)
    out.backward(go)
    gi, gp, gw = oracle.aggregation_backward(npy(x), npy(pos), npy(w), npy(idx2), npy(go))
    close(npy(x.grad), gi)
    close(npy(pos.grad), gp)
    close(npy(w.grad), gw)


def _knn_grid(dev, k, xyz, q, off, noff):
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    b, n, m = len(off), xyz.shape[0], q.shape[0]
    X, Q, O, NO = T(xyz, dev), T(q, dev), T(off, dev), T(noff, dev)
    idx = torch.full((m, k), -1, dtype=torch.int32, device=dev)
    d2 = torch.full((m, k), -1.0, device=dev)
    nbytes = int(L.tgn_knnquery_grid_workspace_bytes(b, n, m))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    _lib.check(L.tgn_knnquery_grid(b, n, m, k, _lib.ptr(X), _lib.ptr(Q), _lib.ptr(O), _lib.ptr(NO), _lib.ptr(idx),
                                   _lib.ptr(d2), _lib.ptr(ws), nbytes, _lib.stream()))
    return idx.cpu().numpy(), np.sqrt(d2.cpu().numpy())


@pytest.mark.parametrize("case", ["arch24k", "volume", "quantised", "ragged", "outside", "degenerate"])
@pytest.mark.parametrize("k", [1, 16, 36, 63])
def test_knn_grid_path_equals_the_heap_order(dev, oracle, case, k):
    """tgn_knnquery_grid against the oracle's verbatim heap: surfaces, volume-filling clouds (the cell-size guess is too
    small: wider blocks), exact ties, ragged batches with tiny segments, queries far outside the box, degenerate clouds."""
    # fixed per-case seeds (python's hash() of a str is randomised per process: a failure would not reproduce)
    rng = np.random.default_rng({"arch24k": 101, "volume": 202, "quantised": 303, "ragged": 404, "outside": 505,
                                 "degenerate": 606}[case] + k)
    if case == "arch24k":
        segs = [synth.arch_cloud(24000, 5, False)]
        qs = [segs[0][::5]]
    elif case == "volume":
        segs = [rng.uniform(-1, 1, size=(15000, 3)).astype(np.float32)]
        qs = [segs[0][::6]]
    elif case == "quantised":
        segs = [(rng.integers(-20, 21, size=(9000, 3)) / 16.0).astype(np.float32)]
        qs = [segs[0][::4]]
    elif case == "ragged":
        segs = [synth.arch_cloud(12000, 1, False), synth.uniform_cloud(300, 2), synth.arch_cloud(6000, 3, False),
                synth.uniform_cloud(1, 4), synth.uniform_cloud(40, 5)]
        qs = [s[::7][:400] if len(s) > 7 else s for s in segs]
    elif case == "outside":
        segs = [synth.arch_cloud(8000, 9, False), synth.arch_cloud(5000, 10, False)]
        far = np.array([[30, 0, 0], [0, -50, 2], [1e3, 1e3, 1e3], [0.2, 0.1, 5.0]], np.float32)
        qs = [np.concatenate([s[::50], far, s[:3] + 0.013]) for s in segs]
    else:
        flat = synth.uniform_cloud(5000, 11)
        flat[:, 2] = 0.5                                   # zero extent in z
        same = np.full((4000, 3), 0.25, np.float32)        # a single point, 4000 times
        nanc = synth.arch_cloud(6000, 12, False)
        nanc[100] = np.nan
        segs = [flat, same, nanc]
        qs = [s[::9] for s in segs]
    xyz, q = np.concatenate(segs), np.concatenate(qs)
    off = np.cumsum([len(s) for s in segs]).astype(np.int32)
    noff = np.cumsum([len(s) for s in qs]).astype(np.int32)
    gi, gd = _knn_grid(dev, k, xyz, q, off, noff)
    oi, od = oracle.knnquery(k, xyz, q, off, noff)
    assert np.array_equal(gi, oi), case
    assert np.array_equal(gd, od, equal_nan=True)
