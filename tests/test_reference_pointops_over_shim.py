"""CPU, build container only: the REFERENCE's own ``external_libs/pointops/functions/pointops.py`` executed over this
repo's ``pointops_cuda.py`` shim.

The reference file is loaded where it lies under /root/reference (never copied); ``import pointops_cuda`` inside it
resolves to the shim, whose ctypes calls land in a stand-in for libtgn_pointops.so that implements the ten
``*_cuda_launcher`` entry points of include/tgn_pointops.h section 1 AT THE POINTER LEVEL with the CPU oracle: it sees
exactly what the C ABI would see (ints + raw addresses, in the reference's argument order), so a swapped or mistyped
argument in the shim produces wrong numbers here.  ``torch.cuda.IntTensor`` / ``FloatTensor`` (pointops.py:21-22 etc.)
are pointed at their CPU namesakes because this container has no GPU; everything else is the reference's code:
the host loop over the offset tensor (:18-20), the 0-d tensor passed as ``n_max``, the legacy constructors, the
autograd Functions and their backward passes.
"""
import ctypes
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

REF = "/root/reference/external_libs/pointops/functions/pointops.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")


def _arr(p, n, ctype, dtype):
    addr = p.value if isinstance(p, ctypes.c_void_p) else int(p)
    return np.ctypeslib.as_array((ctype * n).from_address(addr)).view(dtype)


def _f(p, n):
    return _arr(p, n, ctypes.c_float, np.float32)


def _i(p, n):
    return _arr(p, n, ctypes.c_int32, np.int32)


class OracleBackedLibrary:
    """Pointer-level stand-in for section 1 of include/tgn_pointops.h, computed by oracle/cpu.py."""

    def __init__(self, oracle):
        self.O = oracle
        self.calls = []
        self.stream_set = 0

    def tgn_set_default_stream(self, s):
        self.stream_set += 1

    def furthestsampling_cuda_launcher(self, b, n, xyz, offset, new_offset, tmp, idx):
        self.calls.append(("fps", b, n))
        off, noff = _i(offset, b).copy(), _i(new_offset, b).copy()
        assert n == int(np.diff(np.concatenate([[0], off])).max()), "n must be the largest segment (pointops.py:18-20)"
        assert np.all(_f(tmp, int(off[-1])) == np.float32(1e10)), "tmp arrives pre-filled with 1e10 (pointops.py:22)"
        _i(idx, int(noff[-1]))[:] = self.O.furthestsampling(_f(xyz, int(off[-1]) * 3).reshape(-1, 3), off, noff)

    def knnquery_cuda_launcher(self, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
        self.calls.append(("knn", m, nsample))
        b = self._segments
        off, noff = _i(offset, b).copy(), _i(new_offset, b).copy()
        assert int(noff[-1]) == m
        oi, od = self.O.knnquery(nsample, _f(xyz, int(off[-1]) * 3).reshape(-1, 3), _f(new_xyz, m * 3).reshape(-1, 3),
                                 off, noff)
        _i(idx, m * nsample)[:] = oi.reshape(-1)
        _f(dist2, m * nsample)[:] = (od.astype(np.float32) ** 2).reshape(-1)     # the launcher writes SQUARED distances

    def grouping_forward_cuda_launcher(self, m, nsample, c, inp, idx, output):
        self.calls.append(("grouping_fwd", m, nsample, c))
        ii = _i(idx, m * nsample).reshape(m, nsample)
        n = int(ii.max()) + 1
        _f(output, m * nsample * c)[:] = self.O.grouping_forward(_f(inp, n * c).reshape(n, c), ii).reshape(-1)

    def grouping_backward_cuda_launcher(self, m, nsample, c, grad_output, idx, grad_input):
        self.calls.append(("grouping_bwd", m, nsample, c))
        n = self._n
        g = self.O.grouping_backward(_f(grad_output, m * nsample * c).reshape(m, nsample, c),
                                     _i(idx, m * nsample).reshape(m, nsample), n)
        _f(grad_input, n * c)[:] += g.reshape(-1)

    def interpolation_forward_cuda_launcher(self, n, c, k, inp, idx, weight, output):
        self.calls.append(("interp_fwd", n, c, k))
        m = self._m
        _f(output, n * c)[:] += self.O.interpolation_forward(_f(inp, m * c).reshape(m, c), _i(idx, n * k).reshape(n, k),
                                                             _f(weight, n * k).reshape(n, k)).reshape(-1)

    def interpolation_backward_cuda_launcher(self, n, c, k, grad_output, idx, weight, grad_input):
        self.calls.append(("interp_bwd", n, c, k))
        m = self._m
        _f(grad_input, m * c)[:] += self.O.interpolation_backward(_f(grad_output, n * c).reshape(n, c),
                                                                  _i(idx, n * k).reshape(n, k),
                                                                  _f(weight, n * k).reshape(n, k), m).reshape(-1)

    def subtraction_forward_cuda_launcher(self, n, nsample, c, input1, input2, idx, output):
        self.calls.append(("sub_fwd", n, nsample, c))
        _f(output, n * nsample * c)[:] = self.O.subtraction_forward(_f(input1, n * c).reshape(n, c),
                                                                    _f(input2, n * c).reshape(n, c),
                                                                    _i(idx, n * nsample).reshape(n, nsample)).reshape(-1)

    def subtraction_backward_cuda_launcher(self, n, nsample, c, idx, grad_output, grad_input1, grad_input2):
        self.calls.append(("sub_bwd", n, nsample, c))
        g1, g2 = self.O.subtraction_backward(_i(idx, n * nsample).reshape(n, nsample),
                                             _f(grad_output, n * nsample * c).reshape(n, nsample, c))
        _f(grad_input1, n * c)[:] += g1.reshape(-1)
        _f(grad_input2, n * c)[:] += g2.reshape(-1)

    def aggregation_forward_cuda_launcher(self, n, nsample, c, w_c, inp, position, weight, idx, output):
        self.calls.append(("agg_fwd", n, nsample, c, w_c))
        _f(output, n * c)[:] += self.O.aggregation_forward(_f(inp, n * c).reshape(n, c),
                                                           _f(position, n * nsample * c).reshape(n, nsample, c),
                                                           _f(weight, n * nsample * w_c).reshape(n, nsample, w_c),
                                                           _i(idx, n * nsample).reshape(n, nsample)).reshape(-1)

    def aggregation_backward_cuda_launcher(self, n, nsample, c, w_c, inp, position, weight, idx, grad_output, grad_input,
                                           grad_position, grad_weight):
        self.calls.append(("agg_bwd", n, nsample, c, w_c))
        gi, gp, gw = self.O.aggregation_backward(_f(inp, n * c).reshape(n, c),
                                                 _f(position, n * nsample * c).reshape(n, nsample, c),
                                                 _f(weight, n * nsample * w_c).reshape(n, nsample, w_c),
                                                 _i(idx, n * nsample).reshape(n, nsample),
                                                 _f(grad_output, n * c).reshape(n, c))
        _f(grad_input, n * c)[:] += gi.reshape(-1)
        _f(grad_position, n * nsample * c)[:] = gp.reshape(-1)
        _f(grad_weight, n * nsample * w_c)[:] += gw.reshape(-1)


class _FakeStream:
    cuda_stream = 0


@pytest.fixture()
def ref_pointops(monkeypatch, oracle):
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import pointops_cuda as shim
    fake = OracleBackedLibrary(oracle)
    monkeypatch.setattr(shim, "lib", lambda: fake)
    monkeypatch.setattr(shim, "require_cuda", lambda *t: None)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "IntTensor", torch.IntTensor, raising=False)
    monkeypatch.setattr(torch.cuda, "FloatTensor", torch.FloatTensor, raising=False)
    monkeypatch.setitem(sys.modules, "pointops_cuda", shim)
    sys.dont_write_bytecode, keep = True, sys.dont_write_bytecode     # nothing is written into /root/reference
    try:
        spec = importlib.util.spec_from_file_location("_reference_pointops_under_test", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = keep
    assert mod.pointops_cuda is shim, "the reference's `import pointops_cuda` (pointops.py:7) must resolve to the shim"
    return mod, fake


def _cloud(seed, sizes):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1, 1, size=(sum(sizes), 3)).astype(np.float32)
    return xyz, np.cumsum(sizes).astype(np.int32)


def test_reference_sampling_and_knn_run_unchanged_over_the_shim(ref_pointops, oracle):
    R, fake = ref_pointops
    xyz, off = _cloud(1, [700, 300, 500])
    noff = np.cumsum([70, 50, 125]).astype(np.int32)
    t_xyz, t_off, t_noff = torch.from_numpy(xyz), torch.from_numpy(off), torch.from_numpy(noff)
    idx = R.furthestsampling(t_xyz, t_off, t_noff)
    assert idx.dtype == torch.int32 and np.array_equal(idx.numpy(), oracle.furthestsampling(xyz, off, noff))
    assert fake.calls[-1] == ("fps", 3, 700) and fake.stream_set >= 1
    fake._segments = 3
    new_xyz = torch.from_numpy(xyz[idx.numpy().astype(np.int64)].copy())
    kidx, kdist = R.knnquery(8, t_xyz, new_xyz, t_off, t_noff)
    oi, od = oracle.knnquery(8, xyz, new_xyz.numpy(), off, noff)
    assert np.array_equal(kidx.numpy(), oi) and np.allclose(kdist.numpy(), od, rtol=0, atol=1e-6)
    # queryandgroup / interpolation of the reference are torch code on top of knnquery (pointops.py:79-100,164-180)
    feat = torch.from_numpy(np.random.default_rng(2).normal(size=(xyz.shape[0], 5)).astype(np.float32))
    got = R.queryandgroup(8, t_xyz, new_xyz, feat, None, t_off, t_noff, use_xyz=True)
    want = oracle.queryandgroup(8, xyz, new_xyz.numpy(), feat.numpy(), None, off, noff, True)
    assert np.array_equal(got.numpy(), want)
    got = R.interpolation(new_xyz, t_xyz, torch.from_numpy(feat.numpy()[idx.numpy().astype(np.int64)].copy()), t_noff, t_off)
    want = oracle.interpolation(new_xyz.numpy(), xyz, feat.numpy()[idx.numpy().astype(np.int64)], noff, off)[0]
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-5, atol=1e-6)


def test_reference_gather_family_forward_and_backward_over_the_shim(ref_pointops, oracle):
    R, fake = ref_pointops
    rng = np.random.default_rng(3)
    n, ns, c, w_c = 60, 6, 8, 4
    inp = rng.normal(size=(n, c)).astype(np.float32)
    inp2 = rng.normal(size=(n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(n, ns)).astype(np.int32)
    idx[0, 0] = n - 1          # the stand-in sizes `input` from the largest index
    pos = rng.normal(size=(n, ns, c)).astype(np.float32)
    w = rng.normal(size=(n, ns, w_c)).astype(np.float32)
    t = lambda a, g=False: torch.from_numpy(a.copy()).requires_grad_(g)

    x = t(inp, True)
    fake._n = n
    out = R.grouping(x, t(idx))
    assert np.array_equal(out.detach().numpy(), oracle.grouping_forward(inp, idx))
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(t(g))
    np.testing.assert_allclose(x.grad.numpy(), oracle.grouping_backward(g, idx, n), rtol=1e-5, atol=1e-6)

    a, b = t(inp, True), t(inp2, True)
    out = R.subtraction(a, b, t(idx))
    assert np.array_equal(out.detach().numpy(), oracle.subtraction_forward(inp, inp2, idx))
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(t(g))
    g1, g2 = oracle.subtraction_backward(idx, g)
    np.testing.assert_allclose(a.grad.numpy(), g1, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b.grad.numpy(), g2, rtol=1e-5, atol=1e-5)

    x, p, ww = t(inp, True), t(pos, True), t(w, True)
    out = R.aggregation(x, p, ww, t(idx))
    np.testing.assert_allclose(out.detach().numpy(), oracle.aggregation_forward(inp, pos, w, idx), rtol=1e-5, atol=1e-5)
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(t(g))
    gi, gp, gw = oracle.aggregation_backward(inp, pos, w, idx, g)
    np.testing.assert_allclose(x.grad.numpy(), gi, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p.grad.numpy(), gp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ww.grad.numpy(), gw, rtol=1e-4, atol=1e-5)

    # interpolation2 = knnquery + the native weighted gather, forward and backward (pointops.py:183-216)
    xyz, off = _cloud(4, [40, 20])
    new_xyz, noff = _cloud(5, [90, 30])
    feat = rng.normal(size=(60, 7)).astype(np.float32)
    fake._segments, fake._m = 2, 60
    f = t(feat, True)
    out = R.interpolation2(t(xyz), t(new_xyz), f, t(off), t(noff), 3)
    want = oracle.interpolation(xyz, new_xyz, feat, off, noff, 3)[0]
    np.testing.assert_allclose(out.detach().numpy(), want, rtol=1e-5, atol=1e-6)
    # (a materialised gradient: the reference hands grad_output to the launcher without .contiguous(), pointops.py:213)
    g = np.ones((120, 7), np.float32)
    out.backward(t(g))
    assert f.grad.shape == (60, 7) and abs(float(f.grad.sum()) - 120 * 7) < 1e-2   # the weights of a row sum to 1
    kinds = [c_[0] for c_ in fake.calls]
    for k in ("grouping_fwd", "grouping_bwd", "sub_fwd", "sub_bwd", "agg_fwd", "agg_bwd", "knn", "interp_fwd", "interp_bwd"):
        assert k in kinds, k
