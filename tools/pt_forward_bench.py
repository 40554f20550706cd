#!/usr/bin/env python3
"""BASELINE.json config 4: Point-Transformer (pointops kNN + subtraction attention) forward on 24 000-point scans,
built from the mirror modules of toothgroupnetwork_amd.point_transformer (the reference's PointTransformerSeg encoder /
decoder, tgnet_fps stage sizes).  Times the fused eval path against the unfused composition and the enc1 attention
layer alone against its algorithmic bytes: per point the layer must read x_q, gather nsample rows of x_k and of x_v
(each once) and write one row: 4 * n * c * (2 + 2 * nsample) B + idx 4 * n * nsample B."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from toothgroupnetwork_amd import point_transformer as PT, pointops as P, synth

dev = torch.device("cuda")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)


torch.manual_seed(0)
B = int(os.environ.get("B", "1"))
net = PT.PointTransformerUNet().to(dev).eval()
inp = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 3).transpose(0, 2, 1).copy()).to(dev)
with torch.no_grad():
    t_f = timeit(lambda: net(inp))
inp_g = inp.clone().requires_grad_(True)
t_u = timeit(lambda: net(inp_g))
print(f"PointTransformerUNet forward, {B} x 24000 points: fused eval {t_f:.2f} ms ({B / t_f * 1e3:.1f} scans/s), "
      f"unfused composition (autograd on) {t_u:.2f} ms", flush=True)
# the same forward as a HIP graph (no host round trip is left in the fused eval path; the redo counter of the kNN kernels is
# cleared by a kernel, not by a memset node -- DESIGN.md 4.6)
for presample in (True, False):
    net.presample = presample
    try:
        static_in = inp.clone()
        s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_), torch.no_grad():
            for _ in range(2):
                ref = net(static_in)
        torch.cuda.current_stream().wait_stream(s_); torch.cuda.synchronize()
        P.knn_cache_clear(); P.fps_prefix_clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            static_out = net(static_in)
        P.knn_cache_clear(); P.fps_prefix_clear()
        oks = []
        for _ in range(3):
            g.replay(); torch.cuda.synchronize()
            oks.append(float((static_out - ref).abs().max()))     # (an eager allocation + kernel between replays)
        t_g = timeit(g.replay, reps=10)
        print(f"PointTransformerUNet forward, {B} x 24000 points, HIP graph replay (sampling pyramid {'on a side stream' if presample else 'in line'}): "
              f"{t_g:.2f} ms ({B / t_g * 1e3:.1f} scans/s); max |replay - eager| over 3 replays {max(oks):.2e}", flush=True)
        del g, static_out
    except Exception as e:  # noqa: BLE001
        print(f"graph capture (presample={presample}) failed: {type(e).__name__} {str(e)[:300]}", flush=True)
net.presample = True
# enc1 attention layer alone
n, c, ns = 24000 * B, 32, 36
layer = PT.PointTransformerLayer(c, c, 8, ns).to(dev).eval()
p = inp.permute(0, 2, 1)[:, :, :3].reshape(-1, 3).contiguous()
o = torch.arange(1, B + 1, dtype=torch.int32, device=dev) * 24000
x = torch.randn(n, c, device=dev)
with torch.no_grad():
    xq, xk, xv = layer.linear_q(x), layer.linear_k(x), layer.linear_v(x)
    idx, _ = P.knnquery(ns, p, p, o, o)
    prm = PT.fold_pt_layer(layer)
    t_a = timeit(lambda: PT.pt_attention(p, xq, xk, xv, idx, prm))
    t_k = timeit(lambda: (P.knn_cache_clear(), P.knnquery(ns, p, p, o, o)))
alg = 4 * n * c * (2 + 2 * ns) + 4 * n * ns
print(f"enc1 attention ({n} points, k={ns}, c={c}): fused kernel {t_a:.3f} ms = {alg / t_a / 1e6:.0f} GB/s of {alg / 1e6:.1f} MB "
      f"algorithmic (gathered rows are L2 traffic); kNN {t_k:.3f} ms", flush=True)
