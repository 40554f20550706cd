#!/usr/bin/env python3
"""BASELINE.json config 3: one full training step of the tgnet_fps first-stage network (Point-Transformer U-Net with the
tgnet_fps stage sizes, semantic + offset heads, the reference's loss terms) on ONE 24 000-point scan, batch 1, under
torch.autocast(bfloat16) and in fp32 -- forward (BatchNorm in training mode), losses, backward through this package's
differentiable operators (kNN gathers, fused softmax+aggregation, interpolation, square_distance), Adam step.

The network is built from the mirror modules of toothgroupnetwork_amd.point_transformer (same parameter names as the
reference's blocks.py classes); the heads and losses restate models/modules/grouping_network_module.py:38-58 and
models/tgn_loss.py:6-60,110-129 (tooth_class_loss, batch_center_offset_loss) on synthetic labels -- the reference checkout,
its dataset and its sklearn clustering stage are not on the GPU box.  Index kernels stay fp32 under autocast (the operators
declare custom_fwd(cast_inputs=float32)); the learned layers run in bf16."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from toothgroupnetwork_amd import point_transformer as PT, pointnet2_utils as U, synth


class FirstStage(nn.Module):
    """U-Net + the two heads of the first module (grouping_network_module.py: offset (3) and semantic (17) per point)."""

    def __init__(self, planes=(32, 64, 128, 256, 512), blocks=(2, 3, 4, 6, 3), classes=17):
        super().__init__()
        self.unet = PT.PointTransformerUNet(6, planes, blocks)
        c = planes[0]
        self.offset_head = nn.Sequential(nn.Linear(c, c), nn.BatchNorm1d(c), nn.ReLU(inplace=True), nn.Linear(c, 3))
        self.sem_head = nn.Sequential(nn.Linear(c, c), nn.BatchNorm1d(c), nn.ReLU(inplace=True), nn.Linear(c, classes))

    def forward(self, feat):   # (B, 6, N)
        x = self.unet(feat)
        if self.training:   # heads through the same training helpers as the U-Net's MLPs (fused BatchNorm rows, sliced weight gradients)
            return PT.mlp_train(self.offset_head, x), PT.mlp_train(self.sem_head, x)
        return self.offset_head(x), self.sem_head(x)


def losses_loop(offset, sem, xyz, label):
    """tooth_class_loss (cross entropy, tgn_loss.py:110-129) + batch_center_offset_loss (tgn_loss.py:6-60) the way the
    reference writes them: a python loop over the teeth with boolean-mask indexing (one host round trip per mask)."""
    ce = F.cross_entropy(sem.float(), label)
    cen, dirl, cnt = 0.0, 0.0, 0
    for t in range(1, 17):
        m = label == t
        if int(m.sum()) < 5:
            continue
        pts, off = xyz[m][None], offset[m][None].float()
        c = pts.mean(1, keepdim=True)
        cen = cen + U.square_distance(pts + off, c).sum() / pts.shape[1]
        on = off.norm(dim=2, keepdim=True)
        d = (c - pts) / (c - pts).norm(dim=2, keepdim=True).clamp_min(1e-12)
        keep = on[0, :, 0] > 2e-4
        if bool(keep.any()):
            dot = ((off / on.clamp_min(1e-12))[0][keep] * d[0][keep]).sum(1) - 1.0
            dirl = dirl + (dot * dot).mean()
        cnt += 1
    cnt = max(cnt, 1)
    return ce + 0.03 * cen / cnt + 0.03 * dirl / cnt, ce


def losses(offset, sem, xyz, label, teeth=17):
    """The same two terms without a host round trip: per-tooth sums by index_add over the label column instead of 16 boolean
    masks (static shapes: the whole step can be captured in a HIP graph).  Same value as losses_loop up to summation order
    (tests/test_gpu_train_step.py compares them)."""
    ce = F.cross_entropy(sem.float(), label)
    off = offset.float()
    ones = torch.ones_like(label, dtype=torch.float32)
    n_t = torch.zeros(teeth, device=xyz.device).index_add_(0, label, ones)                       # points per tooth
    c_t = torch.zeros(teeth, 3, device=xyz.device).index_add_(0, label, xyz) / n_t.clamp_min(1.0)[:, None]
    valid = ((n_t >= 5) & (torch.arange(teeth, device=xyz.device) >= 1)).float()                  # the loop's `continue`
    d2 = U.square_distance((xyz + off)[None], c_t[None])[0].gather(1, label[:, None])[:, 0]      # |p + off - c_tooth(p)|^2
    cen_t = torch.zeros(teeth, device=xyz.device).index_add_(0, label, d2) / n_t.clamp_min(1.0)
    to_c = c_t[label] - xyz
    d = to_c / to_c.norm(dim=1, keepdim=True).clamp_min(1e-12)
    on = off.norm(dim=1, keepdim=True)
    keep = (on[:, 0] > 2e-4).float()
    dot = ((off / on.clamp_min(1e-12)) * d).sum(1) - 1.0
    k_t = torch.zeros(teeth, device=xyz.device).index_add_(0, label, keep)
    dir_t = torch.zeros(teeth, device=xyz.device).index_add_(0, label, dot * dot * keep) / k_t.clamp_min(1.0)
    cnt = valid.sum().clamp_min(1.0)
    return ce + 0.03 * (cen_t * valid).sum() / cnt + 0.03 * (dir_t * valid).sum() / cnt, ce


class TwoStage(nn.Module):
    """The two passes of GroupingNetworkModule.forward (grouping_network_module.py:16-101): the first-stage network on the whole
    scan; then, around the centroid of every tooth (the reference clusters the moved foreground points with DBSCAN on the CPU and
    crops with a KDTree, :45-72; here: the label centroids, a distance matrix and topk -- the crops are what matters for the
    operators), the `crop` nearest points, centred (ops_utils.centering_object), as ONE batch of ~14 x 3072 points through a second
    U-Net with offset and mask heads (:82): the small-cloud, many-cloud regime of FPS / kNN (3072 -> 768 -> 192 -> 48 -> 12)."""

    def __init__(self, small=False, crop=3072, teeth=14):
        super().__init__()
        mk = (lambda c: FirstStage((16, 32, 32, 64, 64), (1, 2, 2, 2, 1), classes=c)) if small else (lambda c: FirstStage(classes=c))
        self.first, self.second, self.crop, self.teeth = mk(17), mk(2), crop, teeth

    def forward(self, feat, xyz, label):
        offset, sem = self.first(feat)                                                   # (N,3), (N,17)
        with torch.no_grad():
            T = self.teeth
            ones = torch.ones_like(label, dtype=torch.float32)
            n_t = torch.zeros(17, device=xyz.device).index_add_(0, label, ones)
            c_t = torch.zeros(17, 3, device=xyz.device).index_add_(0, label, xyz) / n_t.clamp_min(1.0)[:, None]
            cent = c_t[1:1 + T]                                                          # T tooth centroids
            d = U.square_distance(cent[None], xyz[None])[0]                              # (T, N)
            idx = d.topk(self.crop, dim=1, largest=False)[1]                             # ops_utils.get_nearest_neighbor_idx
            crops = feat[0][:, idx.reshape(-1)].view(6, T, self.crop).permute(1, 0, 2).contiguous()   # (T, 6, crop)
            crops[:, :3] -= crops[:, :3].mean(2, keepdim=True)                           # centering_object
            mask_gt = (label[idx] == torch.arange(1, 1 + T, device=xyz.device)[:, None]).long().reshape(-1)
        offset2, mask2 = self.second(crops)                                              # (T*crop, 3), (T*crop, 2)
        return offset, sem, offset2, mask2, mask_gt


def run_two_stage(net, opt, feat, xyz, label, steps, amp, graph):
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            offset, sem, offset2, mask2, mask_gt = net(feat, xyz, label)
            loss, _ = losses(offset, sem, xyz, label)
            loss = loss + F.cross_entropy(mask2.float(), mask_gt) + 0.03 * offset2.float().square().sum(1).mean()
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss.detach()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        first = float(step())
        step()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    fn, lossbuf = step, None
    if graph:
        from toothgroupnetwork_amd import pointops as P
        P.knn_cache_clear(); P.fps_prefix_clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lossbuf = step()
        P.knn_cache_clear(); P.fps_prefix_clear()
        fn = g.replay
    ms, last = [], first
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
        last = float(lossbuf if graph else r)
    return dict(ms_per_step=float(np.median(ms[1:])), first_loss=first, last_loss=last)


def make_scan(n, seed, dev):
    pts = synth.scan_batch(1, n, "arch", seed)            # (1, n, 6): xyz + normal
    xyz = torch.from_numpy(pts[0, :, :3]).to(dev)
    ang = torch.atan2(xyz[:, 1] + 0.45, xyz[:, 0])        # position along the arch -> 16 "teeth", crown height -> gingiva
    tooth = (ang.clamp(0, 3.14159) / 3.1416 * 16).long().clamp(0, 15) + 1
    label = torch.where(xyz[:, 2] < xyz[:, 2].median(), torch.zeros_like(tooth), tooth)
    return torch.from_numpy(pts.transpose(0, 2, 1).copy()).to(dev), xyz, label


def run_graph(net, opt, feat, xyz, label, steps, amp):
    """The whole step -- forward, losses, backward, Adam -- captured once in a HIP graph and replayed (no host round trip is left
    in it: static-shape losses, trusted kNN indices, capturable fused Adam)."""
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            offset, sem = net(feat)
            loss, ce = losses(offset, sem, xyz, label)
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss.detach(), ce.detach()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        first = step()          # two eager steps on a side stream (allocator warm-up, as torch's capture recipe asks)
        step()
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    first_loss = float(first[0])
    from toothgroupnetwork_amd import pointops as P
    P.knn_cache_clear(); P.fps_prefix_clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss, ce = step()
    P.knn_cache_clear(); P.fps_prefix_clear()
    out = []
    for it in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        out.append(dict(ms=a.elapsed_time(b), loss=float(loss), ce=float(ce), bad_grads=0, out_dtype="graph"))
    out[0]["loss"] = first_loss
    return out


def run(net, opt, feat, xyz, label, steps, amp):
    out = []
    for it in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            offset, sem = net(feat)
            loss, ce = losses(offset, sem, xyz, label)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        # (checked on the first step only: 569 isfinite + all + item round trips are 15 ms of a step)
        bad = [n for n, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()] if it == 0 else []
        opt.step()
        b.record()
        torch.cuda.synchronize()
        out.append(dict(ms=a.elapsed_time(b), loss=float(loss.detach()), ce=float(ce.detach()), bad_grads=len(bad),
                        out_dtype=str(sem.dtype).replace("torch.", "")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=24000)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--small", action="store_true", help="reduced widths / depths (tests)")
    ap.add_argument("--two-stage", action="store_true", help="also time the full two-stage step (first-stage network + ~14 crops of 3072 "
                    "points through a second U-Net, grouping_network_module.py:16-101), eager and as a HIP graph")
    ap.add_argument("--graph", action="store_true", help="also capture the whole step in a HIP graph and time its replays")
    ap.add_argument("--profile", action="store_true", help="print the top GPU kernels of one bf16-autocast step (torch.profiler)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    feat, xyz, label = make_scan(args.points, 3, dev)
    res = {}
    for amp in (False, True):
        torch.manual_seed(0)
        net = (FirstStage((16, 32, 32, 64, 64), (1, 2, 2, 2, 1)) if args.small else FirstStage()).to(dev).train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
        r = run(net, opt, feat, xyz, label, args.steps, amp)
        res["bf16_autocast" if amp else "fp32"] = dict(ms_per_step=float(np.median([x["ms"] for x in r[1:]])), first_loss=r[0]["loss"],
                                                       last_loss=r[-1]["loss"], bad_grads=sum(x["bad_grads"] for x in r),
                                                       head_dtype=r[0]["out_dtype"], params=sum(p.numel() for p in net.parameters()))
    if args.graph:
        for amp in (False, True):
            for presample in (True, False):
                try:
                    torch.manual_seed(0)
                    net = (FirstStage((16, 32, 32, 64, 64), (1, 2, 2, 2, 1)) if args.small else FirstStage()).to(dev).train()
                    net.unet.presample = presample
                    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
                    r = run_graph(net, opt, feat, xyz, label, args.steps, amp)
                    res[f"graph_{'bf16_autocast' if amp else 'fp32'}_{'side_stream_fps' if presample else 'inline_fps'}"] = dict(
                        ms_per_step=float(np.median([x["ms"] for x in r[1:]])), first_loss=r[0]["loss"], last_loss=r[-1]["loss"])
                except Exception as e:  # noqa: BLE001
                    res[f"graph_{'bf16_autocast' if amp else 'fp32'}_{'side_stream_fps' if presample else 'inline_fps'}"] = f"failed: {type(e).__name__} {str(e)[:200]}"
    if args.two_stage:
        crop = min(3072, args.points // 2)
        for graph in (False, True):
            try:
                torch.manual_seed(0)
                net2 = TwoStage(args.small, crop).to(dev).train()
                net2.first.unet.presample = net2.second.unet.presample = not graph
                opt2 = torch.optim.Adam(net2.parameters(), lr=1e-3, fused=True, capturable=graph)
                res[f"two_stage_fp32_{'graph' if graph else 'eager'}"] = dict(
                    run_two_stage(net2, opt2, feat, xyz, label, args.steps, False, graph), crops=f"14 x {crop}",
                    params=sum(p.numel() for p in net2.parameters()))
            except Exception as e:  # noqa: BLE001
                res[f"two_stage_fp32_{'graph' if graph else 'eager'}"] = f"failed: {type(e).__name__} {str(e)[:300]}"
    if args.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            run(net, opt, feat, xyz, label, 1, True)
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
    print(json.dumps({"workload": f"tgnet_fps first-stage train step, 1 x {args.points} points" + (" (small net)" if args.small else ""), **res}))
    return res


if __name__ == "__main__":
    main()
