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
