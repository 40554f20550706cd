"""CPU: register / scratch budgets the design relies on, read from hipcc's resource-usage remarks (no GPU needed).

bench.py's two-stream schedule only pays off if a wave of the grouping kernel fits beside the FPS level-1 workgroup
on every SIMD: FPS allocates 2 waves x 232 VGPRs (granule 8) of the 512 per SIMD lane, which leaves 48 = two
grouping waves of 24 (wide rows) or one of up to 48 (short rows)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import REPO

CSRC = os.path.join(REPO, "toothgroupnetwork_amd", "csrc")


def _usage(src):
    out = subprocess.run(["python", os.path.join(CSRC, "resource_usage.py"), os.path.join(CSRC, src)],
                         capture_output=True, text=True, check=True).stdout
    table = {}
    for line in out.splitlines():
        m = re.match(r"(?:void )?(\S.*?)\s+vgpr=\s*(\d+).*scratch=\s*(\d+).*lds=(\d+)", line)
        if m:
            table[m.group(1).strip()] = tuple(int(v) for v in m.groups()[1:])
    return table


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_fps_and_group_kernels_can_share_a_cu():
    fps = _usage("fps_bucket.hip")
    vgpr, scratch, lds = fps["tgn::fps_bucket_kernel<512, 48, 0, false>"]
    assert scratch == 0, "the 24 000-point FPS kernel must not spill"
    assert vgpr <= 232, f"FPS level-1 kernel uses {vgpr} VGPRs: no room left for two grouping waves (needs <= 232)"
    g = _usage("group.hip")
    # levels 2 and 3 (rows of >= 64 floats): TWO waves per SIMD beside the FPS workgroup
    gv, gs, glds = g["tgn::group_points_v2_kernel<int, 16, true>"]
    assert gs == 0 and gv <= 24, f"grouping kernel uses {gv} VGPRs (> 24: only one wave fits beside the FPS workgroup)"
    assert lds + 2 * glds <= 160 * 1024
    # level 1 (9-float rows): at least ONE wave per SIMD
    nv, ns, nlds = g["tgn::group_points_v2_kernel<int, 16, false>"]
    assert ns == 0 and nv <= 48, f"short-row grouping kernel uses {nv} VGPRs (> 48: does not fit beside the FPS workgroup)"
    for name, (v, s_, _) in g.items():
        assert s_ == 0, f"{name} spills"
    for name, (v, s, _) in fps.items():
        if name.endswith(", 0, false>") and "56" not in name:
            assert s == 0, f"{name} spills"
