"""Host-side mirror of the Point-Transformer building blocks of the reference
(``models/modules/cbl_point_transformer/blocks.py:14-135``) on top of this package's operators.

Same constructor signatures, sub-module and parameter names as the reference classes, so state_dicts interchange
(tests/test_host_logic.py checks the key sets against the reference's own classes).  What differs is the forward:

* ``PointTransformerLayer`` (blocks.py:31-44), eval mode without autograd: ONE kernel for everything after the three
  input projections (``tgn_pt_attention_forward``) -- gather of keys / values / relative coordinates, both small MLPs
  with their BatchNorms folded, softmax over the neighbours, share_planes-weighted sum.  Training mode keeps the
  learned layers as torch modules (BatchNorm needs batch statistics) and runs the tail -- softmax + weighted sum -- as
  the fused, differentiable ``pt_softmax_aggregate`` (forward + backward kernels).  The neighbour search is shared
  between the two ``queryandgroup`` calls of the reference (pointops' kNN memo).
* ``TransitionDown`` with stride > 1 (blocks.py:62-74), eval mode: FPS (+coordinates from the kernel) -> kNN -> the fused
  set-abstraction kernels (per-point transform on the fp32 matrix cores + gather-max): the (m, nsample, 3+c) tensor of
  the reference is never built.
* ``TransitionUp`` / ``PointTransformerBlock``: the reference's torch composition over ``pointops.interpolation``.

``PointTransformerUNet`` strings them together the way ``PointTransformerSeg.forward`` does
(cbl_point_transformer_module.py:93-160) for the forward benchmark of BASELINE.json config 4 (tools/pt_forward_bench.py).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _derived, _lib, pointnet2_utils as _U, pointops
from ._lib import check, lib, ptr, stream

_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = torch.amp.custom_bwd(device_type="cuda")


class _SoftmaxAggregate(Function):
    """out[n,ch] = sum_j (x_v[idx[n,j],ch] + p_r[n,j,ch]) * softmax_j(logit)[n,j,ch % g]   (blocks.py:41-43)"""

    @staticmethod
    @_fwd
    def forward(ctx, x_v, p_r, logit, idx):
        n, nsample, c = p_r.shape
        g = logit.shape[2]
        x_v, p_r, logit = x_v.contiguous(), p_r.contiguous(), logit.contiguous()
        sm = torch.empty(n, nsample, g, dtype=torch.float32, device=x_v.device)
        out = torch.empty(n, c, dtype=torch.float32, device=x_v.device)
        check(lib().tgn_pt_softmax_aggregate_forward(n, nsample, c, g, ptr(x_v), ptr(p_r), ptr(logit), ptr(idx), ptr(sm), ptr(out),
                                                     stream()), "pt_softmax_aggregate fwd")
        ctx.save_for_backward(x_v, p_r, sm, idx)
        return out

    @staticmethod
    @_bwd
    def backward(ctx, grad_out):
        x_v, p_r, sm, idx = ctx.saved_tensors
        n, nsample, c = p_r.shape
        g = sm.shape[2]
        grad_out = grad_out.contiguous().float()
        g_xv = torch.zeros_like(x_v)
        g_pr = torch.empty_like(p_r)
        g_lg = torch.empty_like(sm)
        check(lib().tgn_pt_softmax_aggregate_backward(n, nsample, c, g, ptr(x_v), ptr(p_r), ptr(sm), ptr(idx), ptr(grad_out),
                                                      ptr(g_xv), ptr(g_pr), ptr(g_lg), stream()), "pt_softmax_aggregate bwd")
        return g_xv, g_pr, g_lg, None


def pt_softmax_aggregate(x_v, p_r, logit, idx):
    """x_v (n_v, c) value rows, p_r (n, nsample, c) position encodings, logit (n, nsample, c // share_planes) attention
    logits, idx (n, nsample) int32 neighbour rows -> (n, c).  Differentiable w.r.t. x_v, p_r and logit."""
    _lib.require_cuda(x_v, p_r, logit, idx)
    return _SoftmaxAggregate.apply(x_v, p_r, logit, idx.to(torch.int32).contiguous())


class _LinearSplitK(Function):
    """y = x W^T + b for TALL inputs (24 000 ... 864 000 rows) and narrow layers (3 ... 128 wide), the training path's
    linears.  The forward and the input gradient are ordinary GEMMs; the weight gradient dW = dy^T x is a contraction over
    the ROWS with a tiny (out x in) result, which rocBLAS runs as one or two tiles walking the whole K (370 us for a 32 x 32
    result over 24 000 rows -- a third of the training step went there).  Here the rows are cut into slices, the slices
    contracted as a batch (one tile each: the whole chip) and the partial results summed."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)          # (under autocast: the bf16 GEMM autocast picks)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1])
        x2 = x.reshape(-1, x.shape[-1])
        if gy2.dtype != x2.dtype:
            x2 = x2.to(gy2.dtype)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (gy2 @ weight.to(gy2.dtype)).reshape(x.shape).to(x.dtype)
        if ctx.needs_input_grad[1]:
            rows = gy2.shape[0]
            if gy2.dtype == torch.float32 and gy2.is_cuda:
                # one pass over gy and x on the fp32 matrix cores, a (cout, cin) partial per row slice (csrc/linear.hip); the bias
                # gradient comes out of the same pass
                gyc, xc = gy2.contiguous(), x2.contiguous()
                cout, cin = gyc.shape[1], xc.shape[1]
                nbytes = int(lib().tgn_linear_wgrad_workspace_bytes(rows, cin, cout))
                ws = torch.empty(nbytes, dtype=torch.uint8, device=gy2.device)
                gw = torch.empty(cout, cin, dtype=torch.float32, device=gy2.device)
                want_b = ctx.has_bias and ctx.needs_input_grad[2]
                gb = torch.empty(cout, dtype=torch.float32, device=gy2.device) if want_b else None
                check(lib().tgn_linear_wgrad(rows, cin, cout, ptr(xc), ptr(gyc), ptr(gw), ptr(gb), ptr(ws), nbytes, stream()), "linear_wgrad")
                gw = gw.to(weight.dtype)
                if want_b:
                    gb = gb.to(weight.dtype)
            else:                                                  # bf16 under autocast: sliced batch GEMM
                slices = max(1, min(512, rows // 1024))
                per = rows // slices
                main = slices * per
                acc = torch.promote_types(gy2.dtype, torch.float32)          # bf16 / fp16 slices summed in fp32, float64 stays float64
                gw = torch.bmm(gy2[:main].reshape(slices, per, -1).transpose(1, 2), x2[:main].reshape(slices, per, -1)).sum(0, dtype=acc)
                if main < rows:
                    gw = gw + (gy2[main:].t() @ x2[main:]).to(acc)
                gw = gw.to(weight.dtype)
        if gb is None and ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy2.sum(0, dtype=torch.promote_types(gy2.dtype, torch.float32)).to(weight.dtype)
        return gx, gw, gb


SPLITK_MIN_ROWS = 256    # (the BLAS pick for a 256 x 256 gradient over 375 rows is one 256 x 256 tile: 94 us)


class _BNRows(Function):
    """Training-mode nn.BatchNorm1d over the rows of x (rows, C) [+ ReLU]: two launches forward, two backward
    (csrc/bnorm.hip) instead of torch's five-odd per normalisation and direction -- the tgnet_fps step has 124 of them, most on
    small deep-stage tensors.  Running statistics and num_batches_tracked are updated by the kernel like the module would."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu):
        rows, C = x.shape
        y = torch.empty_like(x)
        stat = torch.empty(2, C, dtype=torch.float32, device=x.device)             # batch mean, 1 / sqrt(var + eps)
        mean, invstd = stat[0], stat[1]
        # accumulators + ticket, left zeroed by the kernels: one per (module, device, stream), so that two streams (or DataParallel
        # replicas, which share the module's __dict__ entries) never meet in one workspace
        wss = bn.__dict__.setdefault("_tgn_bn_ws", {})
        wkey = (x.device.index, torch.cuda.current_stream().cuda_stream)
        ws = wss.get(wkey)
        if ws is None:
            ws = wss[wkey] = torch.zeros(int(lib().tgn_bn_rows_workspace_bytes(C)), dtype=torch.uint8, device=x.device)
        track = bn.track_running_stats and bn.running_mean is not None
        check(lib().tgn_bn_rows_forward(rows, C, ptr(x), ptr(weight), ptr(bias), float(bn.eps), float(bn.momentum),
                                        ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                        ptr(bn.num_batches_tracked) if track and bn.num_batches_tracked is not None else None,
                                        int(relu), ptr(y), ptr(mean), ptr(invstd), ptr(ws), stream()), "bn_rows_forward")
        if track:   # written through raw pointers: tell torch (the fold memo of the eval paths keys on these counters)
            for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
                if t is not None:
                    torch.autograd.graph.increment_version(t)
        ctx.save_for_backward(x, y if relu else None, weight, mean, invstd)
        ctx.relu, ctx.ws = bool(relu), ws
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, invstd = ctx.saved_tensors
        rows, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dwb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        dgamma, dbeta = dwb[0], dwb[1]
        check(lib().tgn_bn_rows_backward(rows, C, ptr(x), ptr(y), ptr(dy), ptr(weight), ptr(mean), ptr(invstd), int(ctx.relu),
                                         ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ctx.ws), stream()), "bn_rows_backward")
        return dx, dgamma, dbeta, None, None


BN_ROWS = os.environ.get("TGN_BN_ROWS", "1") != "0"


def _plain_batchnorm(bn):
    """Exactly nn.BatchNorm1d, nothing hooked onto it: a subclass (SyncBatchNorm after convert_sync_batchnorm, whose statistics
    span the ranks) or a module with forward hooks must run its own forward."""
    return (type(bn) is nn.BatchNorm1d and not bn._forward_hooks and not bn._forward_pre_hooks and not bn._backward_hooks
            and not bn._backward_pre_hooks)


def bn_rows(bn, x, relu=False):
    """[relu](bn(x)) for x (rows, C): the fused kernels in training mode on fp32 CUDA rows, the module otherwise."""
    if (BN_ROWS and bn.training and _plain_batchnorm(bn) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2 and x.shape[1] <= 1024
            and bn.affine and bn.momentum is not None and x.is_contiguous() and not torch.is_autocast_enabled()):
        return _BNRows.apply(x, bn.weight, bn.bias, bn, relu)
    y = bn(x)
    return F.relu(y) if relu else y


def _lin(mod, x):
    """nn.Linear `mod` applied to x; tall inputs in training take the split-K weight gradient."""
    if (torch.is_grad_enabled() and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and x.numel() // x.shape[-1] >= SPLITK_MIN_ROWS and max(mod.in_features, mod.out_features) <= 256):
        return _LinearSplitK.apply(x, mod.weight, mod.bias)
    return mod(x)


def _mlp_rows(seq, t):
    """nn.Sequential of Linear / BatchNorm1d / ReLU applied to t (n, nsample, c): BatchNorm1d normalises every channel over
    all n * nsample rows -- what the reference gets by transposing to (n, c, nsample) and back (blocks.py:37, 40) -- on the
    FLATTENED (n * nsample, c) view, without the two transposed copies per BatchNorm."""
    n, ns = t.shape[0], t.shape[1]
    return mlp_train(seq, t.reshape(n * ns, -1)).view(n, ns, -1)


def mlp_train(seq, t):
    """nn.Sequential of Linear / BatchNorm1d / ReLU on rows t (rows, c): tall linears through the split-K weight gradient, every
    BatchNorm (with the ReLU behind it) through the fused row kernels."""
    layers = list(seq)
    i = 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, nn.Linear):
            t = _lin(m, t)
        elif isinstance(m, nn.BatchNorm1d):
            fuse = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
            t = bn_rows(m, t, relu=fuse)
            i += 1 if fuse else 0
        else:
            t = m(t)
        i += 1
    return t


def _bn_scale_shift(bn):
    s = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float()
    return s, (bn.bias.detach() - bn.running_mean * s).float()


def folded_linear(lin, bn):
    """(W', b') with eval-mode `bn` folded into `lin`: bn(lin(x)) == x @ W'.T + b'; memoised on `bn`."""
    def fold():
        s, t = _bn_scale_shift(bn)
        W = (lin.weight.detach().float() * s[:, None]).contiguous()
        b = t if lin.bias is None else lin.bias.detach().float() * s + t
        return W, b.contiguous()
    return _derived.cached(bn, "folded_linear", _derived.sources(lin, bn), None, fold)


def mlp_eval(seq, x):
    """nn.Sequential of Linear / BatchNorm1d / ReLU in eval mode with every BatchNorm that follows a Linear folded into it (one GEMM
    with bias instead of GEMM + normalisation) and the ReLUs in place."""
    layers = list(seq)
    i = 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, nn.Linear) and i + 1 < len(layers) and isinstance(layers[i + 1], nn.BatchNorm1d):
            if i + 2 < len(layers) and isinstance(layers[i + 2], nn.ReLU) and not x.requires_grad:
                x = _U.linear_relu(x, *folded_linear(m, layers[i + 1]))      # the ReLU in the GEMM's epilogue: one launch
                i += 3
                continue
            x = F.linear(x, *folded_linear(m, layers[i + 1]))
            i += 2
        elif isinstance(m, nn.ReLU):
            x = torch.relu_(x) if not x.requires_grad else torch.relu(x)
            i += 1
        else:
            x = m(x)
            i += 1
    return x


def fold_pt_layer(layer):
    """Operands of tgn_pt_attention_forward from a PointTransformerLayer in eval mode (include/tgn_pointops.h); memoised on the
    layer until one of its parameters / running statistics changes (_derived.cached)."""
    lp0, bnp, lp3 = layer.linear_p[0], layer.linear_p[1], layer.linear_p[3]
    bnw0, lw2, bnw3, lw5 = layer.linear_w[0], layer.linear_w[2], layer.linear_w[3], layer.linear_w[5]
    return _derived.cached(layer, "pt_attention", _derived.sources(lp0, bnp, lp3, bnw0, lw2, bnw3, lw5), None,
                           lambda: _fold_pt_layer(lp0, bnp, lp3, bnw0, lw2, bnw3, lw5))


def _fold_pt_layer(lp0, bnp, lp3, bnw0, lw2, bnw3, lw5):
    sp, tp = _bn_scale_shift(bnp)
    a1, t1 = _bn_scale_shift(bnw0)
    s3, t3 = _bn_scale_shift(bnw3)
    f = lambda t: t.detach().float().contiguous()
    return dict(Wp1=f(lp0.weight * sp[:, None]), bp1=f(lp0.bias * sp + tp), Wp2=f(lp3.weight), bp2=f(lp3.bias),
                a1=f(a1), t1=f(t1), Ww1=f(lw2.weight * s3[:, None]), bw1=f(lw2.bias * s3 + t3), Ww2=f(lw5.weight), bw2=f(lw5.bias))


def pt_attention(p, x_q, x_k, x_v, idx, params, post=None):
    """The fused eval-mode layer: p (n,3), x_q/x_k/x_v (n,c), idx (n,nsample) int32 -> (n,c).
    post = (scale, shift): relu(out * scale + shift) on the way out (the block's bn2 + ReLU, folded)."""
    n, c = x_q.shape
    nsample = idx.shape[1]
    g = params["Ww2"].shape[0]
    out = torch.empty(n, c, dtype=torch.float32, device=x_q.device)
    P = params
    check(lib().tgn_pt_attention_forward(n, nsample, c, g, ptr(p), ptr(x_q), ptr(x_k), ptr(x_v), ptr(idx), ptr(P["Wp1"]),
                                         ptr(P["bp1"]), ptr(P["Wp2"]), ptr(P["bp2"]), ptr(P["a1"]), ptr(P["t1"]), ptr(P["Ww1"]),
                                         ptr(P["bw1"]), ptr(P["Ww2"]), ptr(P["bw2"]), ptr(post[0]) if post else None,
                                         ptr(post[1]) if post else None, ptr(out), stream()), "pt_attention")
    return out


def _frozen(module, *tensors):
    return (not module.training) and not (torch.is_grad_enabled() and (
        any(t is not None and t.requires_grad for t in tensors) or any(q.requires_grad for q in module.parameters())))


class PointTransformerLayer(nn.Module):
    def __init__(self, in_planes, out_planes, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes = out_planes
        self.share_planes = share_planes
        self.nsample = nsample
        self.linear_q = nn.Linear(in_planes, mid_planes)
        self.linear_k = nn.Linear(in_planes, mid_planes)
        self.linear_v = nn.Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3), nn.ReLU(inplace=True), nn.Linear(3, out_planes))
        self.linear_w = nn.Sequential(nn.BatchNorm1d(mid_planes), nn.ReLU(inplace=True),
                                      nn.Linear(mid_planes, mid_planes // share_planes),
                                      nn.BatchNorm1d(mid_planes // share_planes), nn.ReLU(inplace=True),
                                      nn.Linear(out_planes // share_planes, out_planes // share_planes))
        self.softmax = nn.Softmax(dim=1)

    def forward(self, pxo, post_bn=None):
        """post_bn: the BatchNorm1d the caller applies (followed by ReLU) to this layer's output (PointTransformerBlock.bn2): done here
        -- folded into the fused kernel's epilogue in eval, by the fused rows kernel in training."""
        p, x, o = pxo  # (n, 3), (n, c), (b)
        x_q, x_k, x_v = _lin(self.linear_q, x), _lin(self.linear_k, x), _lin(self.linear_v, x)
        idx = pointops.knn_indices(self.nsample, p, p, o, o)            # one search for both groupings of blocks.py:34-35 (read-only)
        g = self.out_planes // self.share_planes
        # the fused kernel gives a point to a wave, which walks the layer's c x c/8 weight matrix on its own: right for the wide
        # stages (24 000 points x 32 channels: 0.09 ms), wrong for the deep ones -- 93 points x 512 channels keep 93 lone waves busy
        # for 0.94 ms where the composition below needs 0.15 (profiled: 4.3 of the forward's 13 ms went there)
        deep = self.out_planes * g >= 8192 and p.shape[0] < 4096
        if (_frozen(self, p, x) and not deep and self.nsample <= 64 and self.out_planes % 4 == 0 and g in (4, 8, 16, 32, 64)
                and x_q.dtype == torch.float32):
            post = None
            if post_bn is not None:
                post = _derived.cached(post_bn, "scale_shift", _derived.sources(post_bn), None,
                                       lambda: tuple(t.contiguous() for t in _bn_scale_shift(post_bn)))
            return pt_attention(p.contiguous(), x_q.contiguous(), x_k.contiguous(), x_v.contiguous(), idx, fold_pt_layer(self), post)
        # training: the reference's composition with the softmax + weighted sum as one differentiable kernel pair.
        # Under autocast this section stays in fp32: it is bandwidth-bound over (n, nsample, c) tensors that the gather
        # kernels produce and consume as fp32, its learned layers are 3- to c/8-wide, and letting autocast flip every other
        # operator to bf16 costs a cast pass over those tensors each time (measured: 298 ms per step against 114 in fp32).
        with torch.autocast("cuda", enabled=False):
            x_q, x_k, x_v = x_q.float(), x_k.float(), x_v.float()
            # (idx comes from this package's kNN: no index check, i.e. no host round trip per layer)
            x_kg = pointops._QueryGroup.apply(p, p, x_k.contiguous(), idx, True)                            # (n, nsample, 3+c)
            p_r, x_kg = x_kg[:, :, 0:3], x_kg[:, :, 3:]
            p_r = _mlp_rows(self.linear_p, p_r)
            w = _mlp_rows(self.linear_w, x_kg - x_q.unsqueeze(1) + p_r)
            out = pt_softmax_aggregate(x_v.contiguous(), p_r.contiguous(), w.contiguous(), idx)
        return out if post_bn is None else bn_rows(post_bn, out, relu=True)


class TransitionDown(nn.Module):
    def __init__(self, in_planes, out_planes, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        if stride != 1:
            self.linear = nn.Linear(3 + in_planes, out_planes, bias=False)
            self.pool = nn.MaxPool1d(nsample)
        else:
            self.linear = nn.Linear(in_planes, out_planes, bias=False)
        self.bn = nn.BatchNorm1d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def sample_offsets(self, o):
        """Offsets of the down-sampled clouds (blocks.py:64-68: count // stride per cloud), on the device without a host loop
        and -- from the host copy of `o`, known for a dense batch -- on the host without a device->host copy."""
        counts = torch.diff(o, prepend=o.new_zeros(1)) // self.stride
        n_o = torch.cumsum(counts, 0).to(torch.int32)
        o_h, prev, acc, n_o_h = pointops.offsets_host(o)[0], 0, 0, []
        for v in o_h:
            acc += (v - prev) // self.stride
            n_o_h.append(acc)
            prev = v
        return pointops.register_offsets(n_o, n_o_h)

    def forward(self, pxo):
        p, x, o = pxo  # (n, 3), (n, c), (b)
        if self.stride == 1:
            if _frozen(self, x) and x.dtype == torch.float32:
                return [p, _U.linear_relu(x, *folded_linear(self.linear, self.bn)), o]
            return [p, bn_rows(self.bn, _lin(self.linear, x), relu=True), o]
        pre, self._presampled = getattr(self, "_presampled", None), None
        if pre is not None and pre[0] is p and pre[1] is o:
            # sampled ahead on a side stream (PointTransformerUNet._presample): sampling depends on the coordinates only
            n_o, idx, n_p, ev = pre[2:]
            torch.cuda.current_stream().wait_event(ev)
        else:
            n_o = self.sample_offsets(o)
            idx, n_p = pointops.fps_with_coords(p, o, n_o, prefix=True)                 # blocks.py:69-70: indices and p[idx] from one kernel
        C1 = self.linear.out_features
        if _frozen(self, p, x) and self.nsample <= 64 and C1 % 4 == 0 and x.dtype == torch.float32:
            # the whole down-sampling step fused: (m, nsample, 3+c) is never built (blocks.py:71-73)
            kidx = pointops.knn_indices(self.nsample, p, n_p, o, n_o)
            n, c = x.shape
            m = n_p.shape[0]

            def fold():
                s, t = _bn_scale_shift(self.bn)
                W = self.linear.weight.detach().float()                    # (C1, 3+c), columns [xyz, features] (use_xyz=True)
                Wt = torch.cat([W[:, 3:], W[:, :3]], 1).mul(s[:, None]).t().contiguous()   # rows [features..., x, y, z]
                return Wt, Wt[c:].contiguous(), t.contiguous()
            Wt, Wxyz, t = _derived.cached(self, "down", _derived.sources(self.linear, self.bn), c, fold)
            A = torch.empty(n, C1, dtype=torch.float32, device=x.device)
            L = lib()
            check(L.tgn_sa_point_transform(n, c, C1, ptr(p.contiguous()), ptr(x.contiguous()), ptr(Wt), ptr(A), stream()),
                  "sa_point_transform")
            out = torch.empty(m, C1, dtype=torch.float32, device=x.device)
            check(L.tgn_sa_gather_max(1, n, m, self.nsample, C1, ptr(A), ptr(n_p), ptr(Wxyz), ptr(t),
                                      ptr(kidx), 0, 1, ptr(out), stream()), "sa_gather_max")
            return [n_p, out, n_o]
        x = pointops.queryandgroup(self.nsample, p, n_p, x, None, o, n_o, use_xyz=True)  # (m, nsample, 3+c)
        m = x.shape[0]
        x = bn_rows(self.bn, _lin(self.linear, x.reshape(m * self.nsample, -1)), relu=True)   # BatchNorm over all m * nsample rows (blocks.py:72)
        x = x.view(m, self.nsample, -1).max(1)[0]                                        # MaxPool1d(nsample) (:73) -> (m, c)
        return [n_p, x, n_o]


class TransitionUp(nn.Module):
    def __init__(self, in_planes, out_planes=None):
        super().__init__()
        if out_planes is None:
            self.linear1 = nn.Sequential(nn.Linear(2 * in_planes, in_planes), nn.BatchNorm1d(in_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, in_planes), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(nn.Linear(out_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))

    def forward(self, pxo1, pxo2=None):
        if pxo2 is None:
            _, x, o = pxo1
            # x = mlp[x, mlp[mean of the cloud]] (blocks.py:103-116) without the per-cloud host loop
            seg = torch.repeat_interleave(torch.arange(o.shape[0], device=x.device), torch.diff(o, prepend=o.new_zeros(1)).long(),
                                          output_size=x.shape[0])   # (the size is known: no device->host round trip)
            cnt = torch.diff(o, prepend=o.new_zeros(1)).to(x.dtype).unsqueeze(1)
            mean = torch.zeros(o.shape[0], x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, seg, x) / cnt
            if _frozen(self, x) and x.dtype == torch.float32:
                return mlp_eval(self.linear1, torch.cat((x, mlp_eval(self.linear2, mean)[seg]), 1))
            return mlp_train(self.linear1, torch.cat((x, mlp_train(self.linear2, mean)[seg]), 1))
        p1, x1, o1 = pxo1
        p2, x2, o2 = pxo2
        if _frozen(self, x1, x2) and x1.dtype == torch.float32 and x2.dtype == torch.float32:
            return mlp_eval(self.linear1, x1).add_(pointops.interpolation(p2, p1, mlp_eval(self.linear2, x2).contiguous(), o2, o1))
        return mlp_train(self.linear1, x1) + pointops.interpolation(p2, p1, mlp_train(self.linear2, x2).contiguous(), o2, o1)


class PointTransformerBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = nn.Linear(in_planes, planes, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.transformer2 = PointTransformerLayer(planes, planes, share_planes, nsample)
        self.bn2 = nn.BatchNorm1d(planes)
        self.linear3 = nn.Linear(planes, planes * self.expansion, bias=False)
        self.bn3 = nn.BatchNorm1d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo):
        p, x, o = pxo
        identity = x
        if _frozen(self, x) and x.dtype == torch.float32:
            # eval: bn1 / bn3 folded into their linears (a GEMM with bias each), bn2 one fused normalisation kernel
            x = _U.linear_relu(x, *folded_linear(self.linear1, self.bn1))
            x = self.transformer2([p, x, o], post_bn=self.bn2)
            x = F.linear(x, *folded_linear(self.linear3, self.bn3)).add_(identity)
            return [p, torch.relu_(x), o]
        x = bn_rows(self.bn1, _lin(self.linear1, x), relu=True)
        x = self.transformer2([p, x, o], post_bn=self.bn2)
        x = bn_rows(self.bn3, _lin(self.linear3, x))
        x = x + identity
        return [p, self.relu(x), o]


class PointTransformerUNet(nn.Module):
    """Encoder / decoder of PointTransformerSeg (cbl_point_transformer_module.py:44-60, 93-160) with the tgnet_fps stage
    sizes by default (planes 32..512, blocks [2,3,4,6,3], stride [1,4,4,4,4], nsample [36,24,24,24,24]); input
    (B, C, N) channel-first like the reference, output the per-point features of dec1, (B*N, planes[0])."""

    def __init__(self, c=6, planes=(32, 64, 128, 256, 512), blocks=(2, 3, 4, 6, 3), stride=(1, 4, 4, 4, 4),
                 nsample=(36, 24, 24, 24, 24), share_planes=8):
        super().__init__()
        self.in_planes = c
        enc, dec = [], []
        for i in range(5):
            layers = [TransitionDown(self.in_planes, planes[i], stride[i], nsample[i])]
            self.in_planes = planes[i]
            layers += [PointTransformerBlock(planes[i], planes[i], share_planes, nsample[i]) for _ in range(1, blocks[i])]
            enc.append(nn.Sequential(*layers))
        for i in range(4, -1, -1):
            layers = [TransitionUp(self.in_planes, None if i == 4 else planes[i])]
            self.in_planes = planes[i]
            layers += [PointTransformerBlock(planes[i], planes[i], share_planes, nsample[i])]
            dec.append(nn.Sequential(*layers))
        self.enc, self.dec = nn.ModuleList(enc), nn.ModuleList(dec)   # dec[0] = dec5 ... dec[4] = dec1

    presample = True   # the sampling pyramid on a side stream, beside the first stage (False: sampled where the reference does)

    def _presample(self, p, o):
        """The four sampling launches depend on the coordinates only (24 000 -> 6000 is a 4.8 ms serial chain on ONE CU; the
        levels below it are answered by the FPS-of-an-FPS-result identity): they go onto a side stream at the start of the
        forward and run beside the first stage's kNN / attention / linear layers; each transition-down level waits for its
        event.  Same kernels, same results."""
        cur = torch.cuda.current_stream()
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != p.device:
            side = self._side_stream = torch.cuda.Stream(device=p.device)
        side.wait_stream(cur)
        pointops.fps_prefix_adopt(cur, side)      # an FPS result made on `cur` (the resampling in front of the network) counts here
        with torch.cuda.stream(side):
            pp, oo = p, o
            for e in self.enc:
                td = e[0]
                if td.stride == 1:
                    continue
                n_o = td.sample_offsets(oo)
                idx, n_p = pointops.fps_with_coords(pp, oo, n_o, prefix=True)
                ev = torch.cuda.Event()
                ev.record(side)
                for t in (n_o, idx, n_p):
                    t.record_stream(cur)
                td._presampled = (pp, oo, n_o, idx, n_p, ev)
                pp, oo = n_p, n_o

    def forward(self, inputs):
        B, C, N = inputs.shape
        pxo = inputs.permute(0, 2, 1)
        x = pxo.reshape(-1, C).contiguous()
        p = pxo[:, :, :3].reshape(-1, 3).contiguous()
        o = pointops.register_offsets(torch.arange(1, B + 1, dtype=torch.int32, device=inputs.device) * N,
                                      [N * (i + 1) for i in range(B)])
        if self.presample and p.is_cuda:
            self._presample(p, o)
        stages = []
        cur = [p, x, o]
        for e in self.enc:
            cur = e(cur)
            stages.append(cur)
        p5, x5, o5 = stages[4]
        x5 = self.dec[0][1:]([p5, self.dec[0][0]([p5, x5, o5]), o5])[1]
        up = [p5, x5, o5]
        for k, i in enumerate(range(3, -1, -1)):
            pi, xi, oi = stages[i]
            d = self.dec[k + 1]
            xi = d[1:]([pi, d[0]([pi, xi, oi], up), oi])[1]
            up = [pi, xi, oi]
        return up[1]
