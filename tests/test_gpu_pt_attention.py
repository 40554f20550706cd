"""GPU: fused Point-Transformer attention (tgn_pt_attention_forward), its trainable tail (tgn_pt_softmax_aggregate_*),
the fused transition-down step and the U-Net forward built from the mirror modules, against the reference layer's
golden output (tests/golden/make_golden_r2_pt.py), the CPU oracle's float64 restatement of blocks.py:34-43 and torch
autograd."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def close(got, want, what="", tol=1e-5):
    want = np.asarray(want, dtype=np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(np.asarray(got, dtype=np.float64) - want).max())
    assert err <= tol * scale, f"{what}: max abs error {err:.3e} > {tol} * {scale:.3f}"


def _randomise_bn(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for m_ in mod.modules():
        if isinstance(m_, torch.nn.BatchNorm1d):
            m_.running_mean.copy_(torch.randn(m_.num_features, generator=g) * 0.2)
            m_.running_var.copy_(torch.rand(m_.num_features, generator=g) * 1.5 + 0.5)
            m_.weight.data.copy_(torch.rand(m_.num_features, generator=g) + 0.5)
            m_.bias.data.copy_(torch.randn(m_.num_features, generator=g) * 0.1)


def test_fused_layer_reproduces_the_reference_layer(dev, golden_r2):
    """our PointTransformerLayer with the reference layer's weights, eval mode: the single fused kernel against the
    reference's own forward (its torch code over CPU gathers)."""
    from toothgroupnetwork_amd import point_transformer as PT
    g = golden_r2
    layer = PT.PointTransformerLayer(32, 32, 8, 16).to(dev).eval()
    layer.load_state_dict(torch.load(os.path.join(GOLDEN, "pt_layer_weights_r2.pt")))
    with torch.no_grad():
        y = layer([T(g["pt_xyz"], dev), T(g["pt_x"], dev), T(g["pt_off"], dev)])
    close(y.cpu().numpy(), g["pt_y"], "fused layer vs reference layer", tol=2e-5)


@pytest.mark.parametrize("n,c,ns", [(24000, 32, 36), (6000, 64, 24), (1500, 128, 24), (375, 256, 24), (93, 512, 24), (500, 32, 5),
                                     (333, 64, 64)])
def test_fused_attention_vs_oracle(dev, oracle, n, c, ns):
    """every stage width of the tgnet_fps U-Net (enc1 ... enc5: c = 32 ... 512, share_planes 8), enc1 at full size
    (24 000 points, 36 neighbours): the fused kernel against the float64 restatement, eval mode; and the training-path
    composition (fused softmax + aggregation tail) against the same."""
    from toothgroupnetwork_amd import point_transformer as PT, pointops as P, synth
    xyz = synth.arch_cloud(n, seed=n % 89, with_normals=False)
    off = np.array([n], np.int32)
    x = np.random.default_rng(c).normal(size=(n, c)).astype(np.float32)
    torch.manual_seed(c + ns)
    layer = PT.PointTransformerLayer(c, c, 8, ns).to(dev).eval()
    _randomise_bn(layer, c)
    tx, tp, to = T(x, dev), T(xyz, dev), T(off, dev)
    with torch.no_grad():
        y = layer([tp, tx, to])
        xq, xk, xv = layer.linear_q(tx), layer.linear_k(tx), layer.linear_v(tx)
        idx, _ = P.knnquery(ns, tp, tp, to, to)
    sd = {k: v.cpu().numpy() for k, v in layer.state_dict().items()}
    want = oracle.pt_attention_layer(xyz, xq.cpu().numpy(), xk.cpu().numpy(), xv.cpu().numpy(), idx.cpu().numpy(), sd, 8)
    close(y.cpu().numpy(), want, "fused")
    if n <= 6000:
        y2 = layer([tp, tx.clone().requires_grad_(True), to])          # autograd on: the composition with the fused tail
        assert y2.requires_grad
        close(y2.detach().cpu().numpy(), want, "training-path composition")


def test_softmax_aggregate_forward_backward_vs_torch(dev):
    from toothgroupnetwork_amd import point_transformer as PT
    torch.manual_seed(3)
    n, nv, ns, c, s = 300, 340, 11, 48, 8
    g = c // s
    xv = torch.randn(nv, c, device=dev, requires_grad=True)
    pr = torch.randn(n, ns, c, device=dev, requires_grad=True)
    lg = torch.randn(n, ns, g, device=dev, requires_grad=True)
    idx = torch.randint(0, nv, (n, ns), device=dev, dtype=torch.int32)
    go = torch.randn(n, c, device=dev)
    out = PT.pt_softmax_aggregate(xv, pr, lg, idx)
    out.backward(go)
    x2, p2, l2 = (t.detach().clone().requires_grad_(True) for t in (xv, pr, lg))
    w = torch.softmax(l2, dim=1)
    ref = ((x2[idx.long()] + p2).view(n, ns, s, g) * w.unsqueeze(2)).sum(1).view(n, c)      # blocks.py:41-43
    ref.backward(go)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pr.grad, p2.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lg.grad, l2.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(xv.grad, x2.grad, rtol=1e-4, atol=1e-4)      # atomically accumulated


def test_transition_down_fused_equals_the_reference_composition(dev, oracle):
    """stride-4 TransitionDown (blocks.py:62-74), eval: FPS + kNN + fused set-abstraction kernels against the same
    module's torch composition (queryandgroup -> Linear -> BN -> ReLU -> MaxPool) and against the oracle chain."""
    from toothgroupnetwork_amd import point_transformer as PT, synth
    sizes = [3000, 1700]
    xyz = np.concatenate([synth.arch_cloud(m, seed=60 + i, with_normals=False) for i, m in enumerate(sizes)])
    off = np.cumsum(sizes).astype(np.int32)
    x = np.random.default_rng(1).normal(size=(sum(sizes), 32)).astype(np.float32)
    torch.manual_seed(5)
    td = PT.TransitionDown(32, 64, 4, 24).to(dev).eval()
    _randomise_bn(td, 9)
    tp, tx, to = T(xyz, dev), T(x, dev), T(off, dev)
    with torch.no_grad():
        p1, y1, o1 = td([tp, tx, to])
    p2, y2, o2 = td([tp, tx.clone().requires_grad_(True), to])            # autograd on: the torch composition
    assert torch.equal(p1, p2) and torch.equal(o1, o2) and o1.tolist() == [750, 1175]
    close(y1.cpu().numpy(), y2.detach().cpu().numpy(), "fused vs composition")
    n_o = np.array([750, 1175], np.int32)
    fidx = oracle.furthestsampling(xyz, off, n_o)
    assert np.array_equal(p1.cpu().numpy(), xyz[fidx.astype(np.int64)])
    kidx, _ = oracle.knnquery(24, xyz, xyz[fidx.astype(np.int64)], off, n_o)
    W = td.linear.weight.detach().cpu().numpy()
    bn = td.bn
    want = oracle.set_abstraction_first_layer(xyz[None], xyz[fidx.astype(np.int64)][None], x[None], kidx[None].astype(np.int64), W,
                                              np.zeros(64), bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy(),
                                              bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy(), bn.eps, True, reduce_max=True)
    close(y1.cpu().numpy(), want[0], "fused vs oracle")


def test_unet_forward_at_24000_points(dev):
    """BASELINE.json config 4 at full size is pinned against the REFERENCE network (run on CPU in float32 and float64) in
    tests/test_gpu_r4_parity.py::test_point_transformer_whole_network_at_24000_points.  Here only: the fused eval path and the
    differentiable composition of the same operators take the same sampling / neighbour decisions at that size -- their outputs
    differ by fp32 rounding through 23 residual blocks and nothing else (the bound is the reference's own fp32-vs-exact distance
    at this size, 2.3e-3 on these features; tests/golden/make_golden_r4.py prints it)."""
    from toothgroupnetwork_amd import point_transformer as PT, synth
    torch.manual_seed(0)
    net = PT.PointTransformerUNet().to(dev).eval()
    _randomise_bn(net, 1)
    inp = T(synth.scan_batch(1, 24000, "arch", 3).transpose(0, 2, 1).copy(), dev)
    with torch.no_grad():
        y = net(inp)
    assert y.shape == (24000, 32) and torch.isfinite(y).all()
    y2 = net(inp.clone().requires_grad_(True))
    close(y.cpu().numpy(), y2.detach().cpu().numpy(), "fused vs composition through 23 blocks", tol=1e-3)


def test_unet_sampling_on_a_side_stream_changes_nothing(dev):
    """PointTransformerUNet runs its sampling pyramid on a side stream beside the first stage (presample): every level's
    indices and coordinates are the ones of the in-place order bit for bit, and the network output agrees (the head's
    per-cloud mean is an atomic float sum, so the output itself is compared to rounding)."""
    from toothgroupnetwork_amd import point_transformer as PT, pointops as P, synth
    torch.manual_seed(0)
    net = PT.PointTransformerUNet(6, (16, 32, 32, 64, 64), (1, 2, 2, 2, 1)).to(dev).eval()
    _randomise_bn(net, 2)
    for B, n in ((1, 24000), (3, 4000)):
        inp = T(synth.scan_batch(B, n, "arch", 5).transpose(0, 2, 1).copy(), dev)
        p = inp.permute(0, 2, 1)[:, :, :3].reshape(-1, 3).contiguous()
        o = P.register_offsets(torch.arange(1, B + 1, dtype=torch.int32, device=dev) * n, [n * (i + 1) for i in range(B)])
        # the pyramid, sampled where the reference samples it
        P.fps_prefix_clear()
        want, pp, oo = [], p, o
        for e in net.enc:
            if e[0].stride != 1:
                n_o = e[0].sample_offsets(oo)
                idx, n_p = P.fps_with_coords(pp, oo, n_o)
                want.append((idx.clone(), n_p.clone(), n_o.clone()))
                pp, oo = n_p, n_o
        # ... and ahead of time on the side stream
        P.fps_prefix_clear()
        net._presample(p, o)
        torch.cuda.synchronize()
        got = [e[0]._presampled for e in net.enc if e[0].stride != 1]
        assert len(got) == len(want) == 4
        for (idx, n_p, n_o), g in zip(want, got):
            assert torch.equal(idx, g[3]) and torch.equal(n_p, g[4]) and torch.equal(n_o, g[2])
            g[5].synchronize()
        for e in net.enc:
            e[0]._presampled = None
        outs = []
        for pre in (False, True, True):
            net.presample = pre
            P.knn_cache_clear()
            P.fps_prefix_clear()
            with torch.no_grad():
                outs.append(net(inp))
            torch.cuda.synchronize()
        for y in outs[1:]:
            close(outs[0].cpu().numpy(), y.cpu().numpy(), "side-stream sampling vs in-place", tol=1e-5)


def test_transition_down_survives_graph_replays_with_eager_work_in_between(dev):
    """Regression: a captured TransitionDown -- offsets, FPS, kNN, fused set abstraction -- replayed with eager
    allocations and kernels that read its SMALL outputs between the replays.  With the kNN redo counter cleared by
    hipMemsetAsync (a memset node in the graph) the second replay died with a memory access fault just past torch's
    private pool; the counter is now cleared by a kernel."""
    from toothgroupnetwork_amd import point_transformer as PT, pointops as P, synth
    torch.manual_seed(0)
    n = 24000
    pts = T(synth.scan_batch(1, n, "arch", 3)[0], dev)
    p = pts[:, :3].contiguous()
    o = P.register_offsets(torch.tensor([n], dtype=torch.int32, device=dev), [n])
    x = torch.randn(n, 32, device=dev)
    td = PT.TransitionDown(32, 64, 4, 24).to(dev).eval()
    with torch.no_grad():
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(2):
                ref = [t.clone() for t in td([p, x, o])]
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        P.knn_cache_clear()
        P.fps_prefix_clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = td([p, x, o])
        P.knn_cache_clear()
        P.fps_prefix_clear()
        for _ in range(4):
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2])      # eager kernels + temporaries on the small outputs
            assert float(out[0].double().sum()) == float(ref[0].double().sum())
            close(out[1].cpu().numpy(), ref[1].cpu().numpy(), "replayed features", tol=1e-6)


@pytest.mark.parametrize("rows,cin,cout", [(24000, 32, 32), (864000, 32, 4), (6000, 64, 64), (9001, 256, 32), (4097, 3, 3), (50000, 3, 64),
                                            (12345, 35, 64), (70001, 128, 100), (375, 256, 256), (1500, 128, 128), (257, 200, 130),
                                            (64, 32, 32), (3, 5, 7), (1, 40, 33)])
def test_tall_narrow_weight_gradient_kernel(dev, rows, cin, cout):
    """tgn_linear_wgrad (one wave per row slice and 32 x 32 output tile on the fp32 matrix cores straight from global memory, then
    one reduction over the slices) through _LinearSplitK: dW, db and dx against torch's own linear backward in float64."""
    from toothgroupnetwork_amd import point_transformer as PT
    g = torch.Generator(device="cpu").manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g).to(dev).requires_grad_(True)
    lin = torch.nn.Linear(cin, cout).to(dev)
    gy = torch.randn(rows, cout, generator=g).to(dev)
    y = PT._LinearSplitK.apply(x, lin.weight, lin.bias)
    gx, gw, gb = torch.autograd.grad(y, (x, lin.weight, lin.bias), gy)
    x64, w64, gy64 = x.detach().double(), lin.weight.detach().double(), gy.double()
    for got, want, what in ((gw, gy64.t() @ x64, "dW"), (gb, gy64.sum(0), "db"), (gx, gy64 @ w64, "dx")):
        scale = float(want.abs().max())
        assert float((got.double() - want).abs().max()) <= 2e-5 * max(scale, 1.0) * max(1.0, (rows / 24000) ** 0.5), what


@pytest.mark.gpu
def test_fused_eval_block_follows_weight_updates(dev):
    """The eval block memoises its folded operands (toothgroupnetwork_amd/_derived.py): after an optimiser-style in-place update,
    a load_state_dict and a change of the BatchNorm statistics it must give what the unfolded composition gives."""
    from toothgroupnetwork_amd import point_transformer as PT, synth
    torch.manual_seed(5)
    blk = PT.PointTransformerBlock(32, 32, 8, 16).to(dev).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    p = torch.from_numpy(synth.arch_cloud(3000, seed=2, with_normals=False)).to(dev)
    x = torch.randn(3000, 32, device=dev)
    o = torch.tensor([3000], dtype=torch.int32, device=dev)

    def both():
        with torch.no_grad():
            fused = blk([p, x, o])[1]
        for q in blk.parameters():
            q.requires_grad_(True)
        ref = blk([p, x, o])[1].detach()                                       # grad enabled + trainable parameters: composition
        return fused, ref
    a, b = both()
    close(a.cpu().numpy(), b.cpu().numpy(), "fused eval block vs composition", tol=2e-5)
    with torch.no_grad():
        for q in blk.parameters():
            q.mul_(1.05)                                                       # in-place, like an optimiser step
        blk.bn1.running_mean.add_(0.3)
        blk.transformer2.linear_w[0].running_var.mul_(1.2)
    a2, b2 = both()
    assert (a2 - a).abs().max().item() > 1e-3                                  # the outputs did change
    close(a2.cpu().numpy(), b2.cpu().numpy(), "after in-place updates", tol=2e-5)
    sd = {k: (v * 0.9 if v.is_floating_point() else v) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    a3, b3 = both()
    close(a3.cpu().numpy(), b3.cpu().numpy(), "after load_state_dict", tol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,relu", [(2, 3, True), (7, 4, False), (1000, 3, True), (4097, 32, True), (100000, 32, False),
                                         (864000, 4, True), (375, 512, True), (93, 1000, False), (5000, 100, True)])
def test_fused_batchnorm_rows_matches_torch(dev, rows, C, relu):
    """csrc/bnorm.hip (training-mode BatchNorm1d over rows, optionally with the ReLU behind it) against nn.BatchNorm1d + F.relu in
    float64: output, the three gradients, the running statistics and the batch counter."""
    from toothgroupnetwork_amd import point_transformer as PT
    torch.manual_seed(rows + C)
    x = (torch.randn(rows, C, device=dev) * torch.linspace(0.5, 3.0, C, device=dev) + torch.linspace(-2.0, 5.0, C, device=dev)).contiguous()
    dy = torch.randn(rows, C, device=dev)
    bn = torch.nn.BatchNorm1d(C).to(dev).train()
    ref = torch.nn.BatchNorm1d(C).to(dev).double().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    for step in range(2):                                                     # twice: the workspace must come back zeroed
        xg = x.clone().requires_grad_(True)
        y = PT.bn_rows(bn, xg, relu=relu)
        assert y.grad_fn is not None and type(y.grad_fn).__name__ == "_BNRowsBackward"
        y.backward(dy)
        xr = x.double().requires_grad_(True)
        yr = ref(xr)
        yr = torch.relu(yr) if relu else yr
        yr.backward(dy.double())
        tol = dict(atol=2e-5, rtol=2e-5)
        assert torch.allclose(y.double(), yr, **tol), (y.double() - yr).abs().max().item()
        assert torch.allclose(xg.grad.double(), xr.grad, atol=5e-5 * max(1.0, xr.grad.abs().max().item()), rtol=1e-4)
        scale = max(1.0, ref.weight.grad.abs().max().item(), ref.bias.grad.abs().max().item())
        assert torch.allclose(bn.weight.grad.double(), ref.weight.grad, atol=1e-5 * scale * rows ** 0.5, rtol=1e-4)
        assert torch.allclose(bn.bias.grad.double(), ref.bias.grad, atol=1e-5 * scale * rows ** 0.5, rtol=1e-4)
        assert torch.allclose(bn.running_mean.double(), ref.running_mean, atol=1e-6, rtol=1e-6)
        assert torch.allclose(bn.running_var.double(), ref.running_var, atol=1e-6, rtol=1e-5)
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == step + 1
        bn.zero_grad()
        ref.zero_grad()
    assert all(int(w.to(torch.int32).abs().sum()) == 0 for w in bn.__dict__["_tgn_bn_ws"].values())    # left zeroed
    v0 = bn.running_mean._version
    PT.bn_rows(bn, x, relu=relu)
    assert bn.running_mean._version > v0                                      # raw-pointer update made visible to torch
