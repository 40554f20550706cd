"""GPU: the phased three-stream schedule of HotPath against its one-stream results on odd configurations (batch sizes that
are not multiples of 8, int64 indices, Shape B with its multi-scale [features, xyz] layout, the FPS identity shortcut):
tools/hotpath_check.py runs five back-to-back pipelined steps over alternating inputs per configuration and compares
every output tensor bit for bit."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_phased_schedule_equals_one_stream_on_odd_configurations(dev):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "hotpath_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "TOTAL mismatches 0" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["half", "B"])
def test_phased_plan_is_measured_and_sane_on_other_shapes(dev, name):
    """The phased schedule plans itself from three timed launches (hotpath.plan_schedule): on a 12 000-point shape and on Shape B
    it must produce a plan, give the one-stream results, and not be slower than the one-stream schedule."""
    import time

    import torch

    from toothgroupnetwork_amd import hotpath, synth
    if name == "half":
        shape = dict(n=12000, npoint=[2048, 512, 128], radius=[0.07, 0.14, 0.28], nsample=[32, 32, 32], d=[6, 64, 256])
        B = 128
    else:
        shape, B = hotpath.SHAPE_B, 64
    pts = torch.from_numpy(synth.scan_batch(8, shape["n"], "arch", 21)).to(dev).repeat(B // 8, 1, 1).contiguous()
    xyz = pts[:, :, :3].contiguous()
    g = torch.Generator().manual_seed(1)
    feats = [pts] + [torch.randn(B, S, D, generator=g).to(dev) for S, D in zip(shape["npoint"][:-1], shape["d"][1:])]

    def timed(hp, steps=12):
        for _ in range(3):
            hp.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lv = hp.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, lv

    t_one, ref = timed(hotpath.HotPath(B, dev, shape=shape))
    sums = [[float(br["grouped"].double().sum()) for br in l["branches"]] + [int(l["fps_idx"].long().sum())] for l in ref]
    hp = hotpath.HotPath(B, dev, shape=shape, pipeline=True)
    t_ph, lv = timed(hp)
    assert hp.plan is not None and {"spacer_us", "last_query_early", "fps_l1_ms", "group_ms", "setup_ms"} <= set(hp.plan)
    assert 0 <= hp.plan["spacer_us"] <= 300 and hp.plan["fps_l1_ms"] > 0 and hp.plan["group_ms"] > 0
    assert sums == [[float(br["grouped"].double().sum()) for br in l["branches"]] + [int(l["fps_idx"].long().sum())] for l in lv]
    assert t_ph <= 1.10 * t_one, (t_ph, t_one, hp.plan)
