"""Property-based tests (hypothesis) of the CPU oracle -- SURVEY.md section 4 (iii).  The oracle is what every GPU parity test is
measured against, so its invariants are worth shrinking counter-examples for: random clouds of random sizes, including lattices
(exact distance ties) and duplicated points.  No GPU; small sizes; a few seconds in all."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

SET = settings(max_examples=30, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


@st.composite
def clouds(draw, lo=8, hi=400):
    n = draw(st.integers(lo, hi))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    kind = draw(st.sampled_from(["uniform", "lattice", "dups"]))
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        xyz = rng.random((n, 3), dtype=np.float32) * 2 - 1
    elif kind == "lattice":
        xyz = rng.integers(-3, 4, size=(n, 3)).astype(np.float32) * 0.25            # many exact ties and duplicates
    else:
        base = rng.random((max(n // 2, 1), 3), dtype=np.float32)
        xyz = base[rng.integers(0, base.shape[0], size=n)]
    return np.ascontiguousarray(xyz, dtype=np.float32)


@SET
@given(xyz=clouds(), frac=st.floats(0.05, 1.0))
def test_fps_prefix_consistency_and_greedy_definition(oracle, xyz, frac):
    """sampling_cuda_kernel.cu:39-59 / pointnet2_utils.py:103-118: the first sample is point 0; sample j is an arg-max of the running
    minimum distance to the samples before it (first index among exact ties); a shorter run is a prefix of a longer one."""
    n = xyz.shape[0]
    m = max(1, int(n * frac))
    idx = oracle.furthestsampling(xyz, [n], [m]).astype(np.int64)
    assert idx[0] == 0 and idx.min() >= 0 and idx.max() < n
    k = max(1, m // 2)
    assert np.array_equal(oracle.furthestsampling(xyz, [n], [k]), idx[:k])
    dmin = np.full(n, 1e10, dtype=np.float32)
    for j in range(1, m):
        d = xyz - xyz[idx[j - 1]]
        dist = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
        dmin = np.minimum(dmin, dist)
        assert idx[j] == int(np.argmax(dmin)), (j, idx[j], int(np.argmax(dmin)))


@SET
@given(xyz=clouds(lo=16), nq=st.integers(1, 40), k=st.integers(1, 24), seed=st.integers(0, 10 ** 6))
def test_knn_is_the_k_smallest_set_in_ascending_order(oracle, xyz, nq, k, seed):
    """knnquery_cuda_kernel.cu:65-108: ascending distances; the multiset of returned distances is the k smallest; a segment shorter
    than k is padded with (segment start, 1e5) -- pointops.py:30-45 takes the square root."""
    n = xyz.shape[0]
    rng = np.random.default_rng(seed)
    q = np.ascontiguousarray(xyz[rng.integers(0, n, size=nq)] + rng.normal(0, 0.01, size=(nq, 3)).astype(np.float32), dtype=np.float32)
    idx, dist = oracle.knnquery(k, xyz, q, [n], [nq])
    d2 = ((q[:, None, :] - xyz[None, :, :]) ** 2).astype(np.float32)
    d2 = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
    kk = min(k, n)
    assert np.all(np.diff(dist[:, :kk], axis=1) >= 0)
    want = np.sqrt(np.sort(d2, axis=1)[:, :kk])
    np.testing.assert_array_equal(dist[:, :kk], want)
    assert np.array_equal(np.sqrt(np.take_along_axis(d2, idx[:, :kk].astype(np.int64), 1)), dist[:, :kk])
    if k > n:
        assert np.all(idx[:, n:] == 0) and np.all(dist[:, n:] == np.float32(1e5))


@SET
@given(xyz=clouds(lo=16), nq=st.integers(1, 30), k=st.integers(1, 40), radius=st.floats(0.05, 1.5))
def test_ball_query_rows_are_the_first_k_hits_in_index_order(oracle, xyz, nq, k, radius):
    """pointnet2_utils.py:120-144: the first nsample indices, ascending, with square_distance <= radius^2 (fp32 threshold), padded with
    the first hit; an empty ball yields index N everywhere."""
    n = xyz.shape[0]
    new_xyz = np.ascontiguousarray(xyz[None, :nq])
    got = oracle.query_ball_point(radius, k, xyz[None], new_xyz)[0]
    sq = oracle.square_distance(new_xyz, xyz[None])[0]
    thr = oracle.radius_sq_f32(radius)
    for qi in range(new_xyz.shape[1]):
        hits = np.nonzero(~(sq[qi] > thr))[0]
        row = got[qi]
        if hits.size == 0:
            assert np.all(row == n)
            continue
        first = hits[:k]
        assert np.array_equal(row[:first.size], first)
        assert np.all(row[first.size:] == first[0])


@SET
@given(n=st.integers(4, 60), ns=st.integers(1, 9), c=st.integers(1, 12), wdiv=st.sampled_from([1, 2, 3, 4]), seed=st.integers(0, 10 ** 6))
def test_gather_family_backward_is_the_adjoint_of_forward(oracle, n, ns, c, wdiv, seed):
    """grouping / subtraction / aggregation / interpolation (the four *_cuda_kernel.cu pairs): <forward(x), g> == <x, backward(g)> for
    every differentiable operand -- what makes the backward kernels gradients at all."""
    rng = np.random.default_rng(seed)
    wc = max(1, c // wdiv) if c % max(1, c // wdiv) == 0 else c
    R = lambda *s: rng.standard_normal(s).astype(np.float32)   # noqa: E731
    x, y, pos, w = R(n, c), R(n, c), R(n, ns, c), R(n, ns, wc)
    idx = rng.integers(0, n, size=(n, ns)).astype(np.int32)
    g3, g2 = R(n, ns, c), R(n, c)
    dot = lambda a, b: float(np.sum(a.astype(np.float64) * b.astype(np.float64)))   # noqa: E731
    tol = dict(rel=2e-4, abs=2e-4)
    assert dot(oracle.grouping_forward(x, idx), g3) == pytest.approx(dot(x, oracle.grouping_backward(g3, idx, n)), **tol)
    g1, gi2 = oracle.subtraction_backward(idx, g3)
    zero = np.zeros_like(x)
    assert dot(oracle.subtraction_forward(x, zero, idx), g3) == pytest.approx(dot(x, g1), **tol)
    assert dot(oracle.subtraction_forward(zero, y, idx), g3) == pytest.approx(dot(y, gi2), **tol)
    ga, gp, gw = oracle.aggregation_backward(x, pos, w, idx, g2)
    zpos = np.zeros_like(pos)
    assert dot(oracle.aggregation_forward(x, zpos, w, idx), g2) == pytest.approx(dot(x, ga), **tol)
    assert dot(oracle.aggregation_forward(zero, pos, w, idx), g2) == pytest.approx(dot(pos, gp), **tol)
    k = min(3, ns)
    ik, wk = np.ascontiguousarray(idx[:, :k]), rng.random((n, k)).astype(np.float32)
    assert dot(oracle.interpolation_forward(x, ik, wk), g2) == pytest.approx(dot(x, oracle.interpolation_backward(g2, ik, wk, n)), **tol)


@SET
@given(xyz=clouds(lo=12), seed=st.integers(0, 10 ** 6))
def test_three_nn_weights_are_a_partition_of_unity(oracle, xyz, seed):
    """pointnet2_utils.py:333-340: three ascending squared distances; interpolating a constant feature returns the constant."""
    rng = np.random.default_rng(seed)
    s = max(3, xyz.shape[0] // 3)
    xyz2 = np.ascontiguousarray(xyz[None, rng.permutation(xyz.shape[0])[:s]])
    d, i = oracle.three_nn(xyz[None], xyz2)
    assert np.all(np.diff(d, axis=2) >= 0) and i.min() >= 0 and i.max() < s
    const = np.full((1, s, 5), 2.5, dtype=np.float32)
    np.testing.assert_allclose(oracle.three_interpolate(const, d, i), 2.5, rtol=0, atol=2e-6)
