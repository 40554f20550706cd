#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path for the headline workload, where the reference checkout exists (the build container):
BASELINE.md section 3's protocol.

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_cpu_baseline.py [--runs 5] [--out profiles/r04_reference_cpu.json]

The code timed is the reference's, imported from /root/reference with `pointops_cuda` stubbed (nothing of this repo's product or
oracle computes anything here):
  FPS         external_libs/pointnet2_utils/pointnet2_utils.py:103-118 `farthest_point_sample_np`, its random start
              (`torch.randint`, :109) patched to 0 = the deterministic start of the CUDA kernel (sampling_cuda_kernel.cu:39)
  ball query  :120-144 `query_ball_point`
  group       the `index_points` / centre / `cat` lines of `sample_and_group` (:162-169)
on Shape A (BASELINE.json config 2: 24 000-point scan, npoint [4096, 1024, 256], nsample 32, radii [0.05, 0.1, 0.2], D [6, 128, 512]),
one scan at a time as the reference's batch-1 loop does, with torch.set_num_threads(1) and (nproc); 1 warm-up + `runs` timed
passes, median.  The GPU box has no reference checkout, so bench.py quotes the committed result of this script
(`cpu_baseline.reference_torch_cpu`) next to the C/OpenMP port it times live."""
import argparse
import json
import os
import platform
import sys
import time
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("TGN_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import synth  # noqa: E402  (synthetic scans only: the same generator bench.py uses)

SHAPE_A = dict(n=24000, npoint=[4096, 1024, 256], radius=[0.05, 0.1, 0.2], nsample=[32, 32, 32], d=[6, 128, 512])


def load_reference():
    import importlib.util
    sys.modules["pointops_cuda"] = types.ModuleType("pointops_cuda")
    path = os.path.join(REFERENCE, "external_libs", "pointnet2_utils", "pointnet2_utils.py")
    # its `from external_libs.pointops.functions import pointops` must resolve to the REFERENCE's file
    for name, rel in (("external_libs", None), ("external_libs.pointops", "pointops/__init__.py"),
                      ("external_libs.pointops.functions", "pointops/functions/__init__.py"),
                      ("external_libs.pointops.functions.pointops", "pointops/functions/pointops.py")):
        if rel is None:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REFERENCE, "external_libs")]
        else:
            p = os.path.join(REFERENCE, "external_libs", rel)
            spec = importlib.util.spec_from_file_location(name, p, submodule_search_locations=[os.path.dirname(p)] if rel.endswith("__init__.py") else None)
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("reference_pointnet2_utils", path)
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    assert R.__file__.startswith(REFERENCE)
    return R


def one_scan(R, scan, feats, shape):
    """The three levels of the headline workload on one scan; returns the seconds spent in (fps, ball, group) per level."""
    xyz = torch.from_numpy(np.ascontiguousarray(scan[None, :, :3]))
    pts = torch.from_numpy(np.ascontiguousarray(scan[None]))
    out = []
    for li, (S, r, K) in enumerate(zip(shape["npoint"], shape["radius"], shape["nsample"])):
        t0 = time.perf_counter()
        fps_idx = torch.from_numpy(R.farthest_point_sample_np(xyz.numpy(), S))               # :103-118
        new_xyz = R.index_points(xyz, fps_idx)                                               # :160
        t1 = time.perf_counter()
        idx = R.query_ball_point(r, K, xyz, new_xyz)                                         # :161
        t2 = time.perf_counter()
        grouped_xyz = R.index_points(xyz, idx)                                               # :162-169
        grouped_xyz_norm = grouped_xyz - new_xyz.view(1, S, 1, 3)
        new_points = torch.cat([grouped_xyz_norm, R.index_points(pts, idx)], dim=-1)
        t3 = time.perf_counter()
        assert tuple(new_points.shape) == (1, S, K, 3 + shape["d"][li])
        out.append((t1 - t0, t2 - t1, t3 - t2))
        xyz = new_xyz
        pts = feats[li] if li < len(feats) else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r04_reference_cpu.json"))
    args = ap.parse_args()
    R = load_reference()
    keep = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get("dtype", torch.long))      # start index 0
    shape = SHAPE_A
    scans = synth.scan_batch(args.runs + 1, shape["n"], "arch", seed=100)
    g = torch.Generator().manual_seed(100)
    feats = [torch.randn(1, S, D, generator=g) for S, D in zip(shape["npoint"][:-1], shape["d"][1:])]
    res = {}
    try:
        for threads in (os.cpu_count() or 1, 1):
            torch.set_num_threads(threads)
            per = []
            for i in range(args.runs + 1):
                t0 = time.perf_counter()
                lv = one_scan(R, scans[i], feats, shape)
                per.append((time.perf_counter() - t0, lv))
                print(f"threads={threads} run {i}: {per[-1][0]:.2f} s/scan  levels (fps, ball, group) {[tuple(round(x, 3) for x in l) for l in lv]}", flush=True)
            per = per[1:]                                                                     # warm-up dropped
            tot = float(np.median([p[0] for p in per]))
            lv = np.median(np.array([p[1] for p in per]), axis=0)
            res[f"threads_{threads}"] = {"seconds_per_mesh": tot, "meshes_per_s": 1.0 / tot,
                                         "per_level_seconds_fps_ball_group": [[round(float(x), 4) for x in row] for row in lv]}
    finally:
        torch.randint = keep
    out = {"what": "the reference's own torch-CPU functions (farthest_point_sample_np with start 0, query_ball_point, index_points / "
                   "centre / cat of sample_and_group; pointnet2_utils.py:103-175) on Shape A, one 24 000-point scan at a time",
           "protocol": f"1 warm-up + {args.runs} timed scans, median; torch.set_num_threads(n)",
           "host": {"machine": platform.machine(), "cpu": _cpu_model(), "nproc": os.cpu_count(), "torch": torch.__version__,
                    "where": "build container (the GPU box has no reference checkout)"},
           "shape": shape, "results": res}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


if __name__ == "__main__":
    main()
