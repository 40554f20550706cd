"""GPU: BASELINE.json config 3 -- a whole training step (forward in training mode, the reference's loss terms, backward
through the package's differentiable operators, optimizer step) under bf16 autocast and in fp32 (tools/train_step_bench.py,
reduced network, 4096 points).  Every parameter must receive a finite gradient in both precisions, the bf16 loss must
track the fp32 one, and the loss must go down."""
import importlib.util
import os
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_train_step_bf16_autocast_and_fp32(dev, monkeypatch):
    spec = importlib.util.spec_from_file_location("train_step_bench", os.path.join(REPO, "tools", "train_step_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["train_step_bench.py", "--points", "4096", "--steps", "5", "--small"])
    res = mod.main()
    f, b = res["fp32"], res["bf16_autocast"]
    assert f["bad_grads"] == 0 and b["bad_grads"] == 0
    assert b["head_dtype"] == "bfloat16" and f["head_dtype"] == "float32"
    assert abs(b["first_loss"] - f["first_loss"]) <= 0.05 * abs(f["first_loss"])   # same weights, same scan
    assert f["last_loss"] < f["first_loss"] and b["last_loss"] < b["first_loss"]


def _bench_module():
    spec = importlib.util.spec_from_file_location("train_step_bench", os.path.join(REPO, "tools", "train_step_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_static_losses_equal_the_reference_style_loop(dev):
    """tools/train_step_bench.losses (index_add over the label column, no host round trip) against losses_loop (the per-tooth
    boolean-mask loop of tgn_loss.py:6-60): same value and same gradients up to summation order."""
    import torch
    mod = _bench_module()
    feat, xyz, label = mod.make_scan(6000, 5, dev)
    torch.manual_seed(1)
    offset = (0.05 * torch.randn(6000, 3, device=dev)).requires_grad_(True)
    offset.data[::7] = 0.0                                   # points below the direction term's 2e-4 threshold
    sem = torch.randn(6000, 17, device=dev, requires_grad=True)
    a, _ = mod.losses_loop(offset, sem, xyz, label)
    ga = torch.autograd.grad(a, (offset, sem))
    b, _ = mod.losses(offset, sem, xyz, label)
    gb = torch.autograd.grad(b, (offset, sem))
    assert abs(float(a.detach()) - float(b.detach())) <= 1e-5 * abs(float(a.detach()))
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 1e-6 + 1e-4 * float(x.abs().max())


def test_whole_step_replays_as_a_hip_graph(dev, monkeypatch):
    """forward + losses + backward + Adam captured once and replayed: the loss must go down over the replays and start where the
    eager step starts."""
    mod = _bench_module()
    monkeypatch.setattr(sys, "argv", ["train_step_bench.py", "--points", "4096", "--steps", "6", "--small", "--graph"])
    res = mod.main()
    for key in ("graph_fp32_inline_fps", "graph_bf16_autocast_inline_fps"):
        g = res[key]
        assert isinstance(g, dict), g
        assert g["last_loss"] < g["first_loss"]
    assert abs(res["graph_fp32_inline_fps"]["first_loss"] - res["fp32"]["first_loss"]) <= 0.02 * abs(res["fp32"]["first_loss"])
