"""CPU: register / scratch budgets the design relies on, read from hipcc's resource-usage remarks (no GPU needed).

bench.py's pipelined schedule only pays off if the grouping kernels fit beside the FPS level-1 workgroup on every CU:
FPS allocates 2 waves x 232 VGPRs (granule 8) of the 512 per SIMD lane and 63 KiB of the 160 KiB of LDS, which leaves
one wave of up to 48 VGPRs per SIMD (or two of 24) and ~95 KiB: four single-wave workgroups of the row-piece kernel
(19 KiB each) or one 256-thread workgroup of the pairs kernel (18 KiB)."""
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
    vgpr, scratch, lds = fps["tgn::fps_bucket_kernel<512, 48, 0, false, 4>"]
    assert scratch == 0, "the 24 000-point FPS kernel must not spill"
    assert vgpr <= 232, f"FPS level-1 kernel uses {vgpr} VGPRs: no room left for a grouping wave per SIMD (needs <= 232)"
    assert lds <= 66 * 1024, f"FPS level-1 kernel holds {lds} B of LDS: the row-piece grouping kernel no longer fits 4 waves beside it"
    g = _usage("group.hip")
    # levels 2 and 3 (rows of >= 64 floats): the row-piece kernel, ONE wave per SIMD beside the FPS workgroup, 4 per CU
    # (dynamic LDS: 2 images of 32 + R*C floats + 512 floats of set-up planes, R*C <= 2176 -- group.hip launcher)
    rows_lds = (2 * (32 + 2176) + 64 * 3 + 64 * 5) * 4
    # (store policy 2 = nt is what the launcher picks by default, 16 = sc1)
    for name in ("tgn::group_points_rows_kernel<int, 2, 4>", "tgn::group_points_rows_kernel<long long, 2, 4>",
                 "tgn::group_points_rows_kernel<int, 2, 1>", "tgn::group_points_rows_kernel<int, 16, 4>",
                 "tgn::group_points_rows_kernel<long long, 16, 4>", "tgn::group_points_rows_kernel<int, 16, 1>"):
        gv, gs, _ = g[name]
        assert gs == 0 and gv <= 48, f"{name} uses {gv} VGPRs (> 48: does not fit beside the FPS workgroup)"
    assert lds + 4 * rows_lds <= 160 * 1024
    # level 1 (9-float rows): the pairs kernel, one wave per SIMD = one 256-thread workgroup per CU
    for idx_t, pol in (("int", 2), ("long long", 2), ("int", 16), ("long long", 16)):
        nv, ns, nlds = g[f"tgn::group_points_pairs_kernel<{idx_t}, 6, {pol}>"]
        assert ns == 0 and nv <= 48, f"pairs grouping kernel ({idx_t}) uses {nv} VGPRs (> 48: does not fit beside the FPS workgroup)"
        assert lds + nlds <= 160 * 1024
    # the last level's ball query (a scan) runs there too, in front of the groupings: two of its waves per SIMD
    bq = _usage("ball_query.hip")
    qv, qs, _ = bq["tgn::ball_query_scan_kernel<int>"]
    assert qs == 0 and qv <= 24, f"scan ball query uses {qv} VGPRs (> 24: one wave per SIMD beside the FPS workgroup)"
    # the LDS-light fallback (shapes the two above do not take): two waves per SIMD
    gv, gs, glds = g["tgn::group_points_v2_kernel<int, 16, true>"]
    assert gs == 0 and gv <= 24, f"v2 grouping kernel uses {gv} VGPRs (> 24: only one wave fits beside the FPS workgroup)"
    for name, (v, s_, _) in g.items():
        assert s_ == 0, f"{name} spills"
    for name, (v, s, _) in fps.items():
        if ", 0, false, 4>" in name and "56" not in name:
            assert s == 0, f"{name} spills"


def _isa(src):
    out = os.path.join("/tmp", "tgn_isa_" + src.replace(".hip", ".s"))
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-munsafe-fp-atomics",
                    "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)], check=True, capture_output=True)
    return open(out).read()


def _kernel_body(isa, mangled_prefix):
    start = isa.index("\n" + mangled_prefix)
    return isa[start:isa.index("s_endpgm", start)]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_wide_builtin_loads_are_not_narrowed():
    """hipcc 7.2 has narrowed `raw_buffer_load_b64 / _b96 / _b128` whose components are used one by one to a single dword
     -- silently wrong data.  The kernels that rely on 16-byte builtin loads must still contain them."""
    ball = _isa("ball_query.hip")
    for idx_t in ("i", "x"):
        body = _kernel_body(ball, f"_ZN3tgn29ball_grid_query_bitmap_kernelI{idx_t}EE")
        assert body.count("buffer_load_dwordx4") >= 3, "the record loads of the ball query were narrowed"
    sa = _isa("sa.hip")
    for name in ("_ZN3tgn20sa_gather_max_kernelIiEE", "_ZN3tgn20sa_gather_max_kernelIxEE"):
        assert _kernel_body(sa, name).count("buffer_load_dwordx4") >= 1, name
