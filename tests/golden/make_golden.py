#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own torch
functions on CPU, and check the C oracle against them while doing so ("pinning" the oracle).

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

How the reference is executed:
  * `pointops_cuda` (CUDA-only, not buildable here) is stubbed in sys.modules so that
    external_libs/pointnet2_utils/pointnet2_utils.py imports;
  * the reference's pure-torch functions run unmodified on CPU tensors: square_distance,
    index_points, query_ball_point, sample_and_group(_all), PointNetSetAbstraction[Msg],
    PointNetFeaturePropagation, farthest_point_sample_np;
  * `farthest_point_sample_np` draws a random start (pointnet2_utils.py:109); torch.randint is
    patched to return 0, the deterministic start of the CUDA kernel (sampling_cuda_kernel.cu:39);
  * `farthest_point_sample` (CUDA only) is replaced by that CPU function for the composite
    fixtures (sample_and_group, SetAbstraction).
Nothing is written under /root/reference (sys.dont_write_bytecode).
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("TGN_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import cpu as O  # noqa: E402
from toothgroupnetwork_amd import synth  # noqa: E402


def load_reference():
    sys.modules["pointops_cuda"] = types.ModuleType("pointops_cuda")
    # import the reference's module under a private name so the repo's own external_libs is not shadowed
    import importlib.util
    pkg_root = os.path.join(REFERENCE, "external_libs")
    for name, path in [
        ("external_libs", None),
        ("external_libs.pointops", os.path.join(pkg_root, "pointops", "__init__.py")),
        ("external_libs.pointops.functions", os.path.join(pkg_root, "pointops", "functions", "__init__.py")),
        ("external_libs.pointops.functions.pointops", os.path.join(pkg_root, "pointops", "functions", "pointops.py")),
    ]:
        if path is None:
            m = types.ModuleType(name)
            m.__path__ = [pkg_root]
        else:
            spec = importlib.util.spec_from_file_location(name, path,
                                                          submodule_search_locations=[os.path.dirname(path)]
                                                          if path.endswith("__init__.py") else None)
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location(
        "reference_pointnet2_utils", os.path.join(pkg_root, "pointnet2_utils", "pointnet2_utils.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    return R


def ref_fps(R, xyz, npoint):
    """reference farthest_point_sample_np with the random start forced to 0."""
    orig = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get("dtype", torch.long))
    try:
        return R.farthest_point_sample_np(np.ascontiguousarray(xyz, dtype=np.float32), npoint)
    finally:
        torch.randint = orig


def check(name, a, b, exact=True, tol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if exact:
        bad = int((a != b).sum())
        assert bad == 0, f"{name}: oracle differs from reference in {bad} of {a.size} entries"
    else:
        err = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
        assert err <= tol, f"{name}: max abs err {err} > {tol}"
    print(f"  oracle == reference : {name} {a.shape}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    R = load_reference()
    out = {}

    # ---- FPS (a3 restated with start 0; pins a1/a2 canonical mode) -------------------------
    clouds = {
        "arch": np.stack([synth.arch_cloud(3000, seed=s, with_normals=False) for s in (0, 1)]),
        "uniform": np.stack([synth.uniform_cloud(3000, seed=s) for s in (2, 3)]),
        "lattice": synth.lattice_cloud(9, dup=71, seed=4)[None],  # 800 points, exact ties + duplicates
    }
    for k, xyz in clouds.items():
        npoint = 512 if k != "lattice" else 300
        ref_idx = ref_fps(R, xyz, npoint)
        ora = O.farthest_point_sample(xyz, npoint)
        check(f"fps_{k}", ora, ref_idx)
        out[f"fps_{k}_xyz"] = xyz
        out[f"fps_{k}_idx"] = ref_idx.astype(np.int32)

    # ---- square_distance (a12) ---------------------------------------------------------------
    src = torch.from_numpy(clouds["arch"][:, :200])
    dst = torch.from_numpy(clouds["arch"][:, 200:1200])
    sd = R.square_distance(src, dst).numpy()
    check("square_distance", O.square_distance(src.numpy(), dst.numpy()), sd)
    out["sqd_src"], out["sqd_dst"], out["sqd_out"] = src.numpy(), dst.numpy(), sd
    # tiny / degenerate shapes go through other BLAS paths: still bit-identical?
    for (n_, m_) in [(1, 7), (3, 1), (17, 33)]:
        a = torch.from_numpy(synth.uniform_cloud(n_, seed=50)[None])
        b = torch.from_numpy(synth.uniform_cloud(m_, seed=51)[None])
        check(f"square_distance_{n_}x{m_}", O.square_distance(a.numpy(), b.numpy()), R.square_distance(a, b).numpy())

    # ---- query_ball_point (a14) --------------------------------------------------------------
    xyz = clouds["arch"]
    new_xyz = O.index_points(xyz, out["fps_arch_idx"][:, :256].astype(np.int64))
    for ri, (radius, ns) in enumerate([(0.05, 16), (0.1, 32), (0.2, 64), (0.3, 8)]):
        ref = R.query_ball_point(radius, ns, torch.from_numpy(xyz), torch.from_numpy(new_xyz)).numpy()
        check(f"ball_query_r{radius}", O.query_ball_point(radius, ns, xyz, new_xyz), ref)
        out[f"ball_{ri}_idx"] = ref.astype(np.int32)
        out[f"ball_{ri}_cfg"] = np.array([radius, ns], dtype=np.float64)
    out["ball_xyz"], out["ball_new_xyz"] = xyz, new_xyz
    # threshold semantics: radius**2 is a python double, compared against an fp32 tensor.
    # Find a radius whose fp32(radius**2) rounds UP and a distance equal to it: torch compares in fp32.
    for radius in np.linspace(0.0301, 0.4, 4000):
        r2d = float(radius) ** 2
        if float(np.float32(r2d)) > r2d:
            break
    p = np.zeros((1, 2, 3), dtype=np.float32)
    p[0, 1, 0] = np.sqrt(np.float32(r2d))
    q = np.zeros((1, 1, 3), dtype=np.float32)
    d01 = R.square_distance(torch.from_numpy(q), torch.from_numpy(p)).numpy()[0, 0, 1]
    ref = R.query_ball_point(float(radius), 2, torch.from_numpy(p), torch.from_numpy(q)).numpy()
    print(f"  threshold probe: radius={radius!r} r2(double)={r2d!r} fp32(r2)={float(np.float32(r2d))!r} d={float(d01)!r} -> {ref.tolist()}")
    check("ball_query_threshold", O.query_ball_point(float(radius), 2, p, q), ref)

    # ---- index_points (a13), sample_and_group (a15), sample_and_group_all --------------------
    pts6 = np.stack([synth.arch_cloud(3000, seed=s) for s in (0, 1)])  # xyz + normals, same xyz as clouds["arch"]
    assert np.array_equal(pts6[:, :, :3], xyz)
    R.farthest_point_sample = lambda x, n: torch.from_numpy(ref_fps(R, x.numpy(), n))
    nx, npts, gxyz, fidx = R.sample_and_group(128, 0.1, 16, torch.from_numpy(xyz), torch.from_numpy(pts6), returnfps=True)
    o_nx, o_np, o_fidx, o_gidx = O.sample_and_group(128, 0.1, 16, xyz, pts6, xyz_first=True)
    check("sample_and_group.fps_idx", o_fidx, fidx.numpy())
    check("sample_and_group.new_xyz", o_nx, nx.numpy())
    check("sample_and_group.new_points", o_np, npts.numpy())
    out["sag_new_xyz"], out["sag_new_points"] = nx.numpy(), npts.numpy()
    out["sag_points"] = pts6
    ax, ap = R.sample_and_group_all(torch.from_numpy(xyz), torch.from_numpy(pts6))
    out["saga_new_points"] = ap.numpy()[:, :, :64]

    # ---- three_nn + three_interpolate lines of PointNetFeaturePropagation (a18) --------------
    xyz1 = torch.from_numpy(xyz[:, :1500])
    xyz2 = torch.from_numpy(new_xyz[:, :200])
    d = R.square_distance(xyz1, xyz2)
    d, i = d.sort(dim=-1)
    d, i = d[:, :, :3], i[:, :, :3]
    od, oi = O.three_nn(xyz1.numpy(), xyz2.numpy())
    check("three_nn.dist", od, d.numpy())
    check("three_nn.idx", oi, i.numpy())
    feat2 = torch.from_numpy(np.random.default_rng(7).normal(size=(2, 200, 24)).astype(np.float32))
    dist_recip = 1.0 / (d + 1e-8)
    norm = torch.sum(dist_recip, dim=2, keepdim=True)
    weight = dist_recip / norm
    interp = torch.sum(R.index_points(feat2, i) * weight.view(2, 1500, 3, 1), dim=2)
    check("three_interpolate", O.three_interpolate(feat2.numpy(), od, oi), interp.numpy(), exact=False, tol=1e-5)
    out["tnn_xyz1"], out["tnn_xyz2"] = xyz1.numpy(), xyz2.numpy()
    out["tnn_dist"], out["tnn_idx"] = d.numpy(), i.numpy().astype(np.int32)
    out["tnn_feat2"], out["tnn_interp"] = feat2.numpy(), interp.numpy()

    # ---- module-level fixtures (eval mode, seeded weights) -----------------------------------
    torch.manual_seed(1234)
    sa = R.PointNetSetAbstractionMsg(128, [0.1, 0.2], [8, 16], 6, [[16, 24], [16, 32]]).eval()
    ssg = R.PointNetSetAbstraction(64, 0.2, 16, 6 + 3, [16, 32], False).eval()
    fp = R.PointNetFeaturePropagation(56 + 6, [32, 16]).eval()
    for mod in (sa, ssg, fp):
        for m_ in mod.modules():
            if isinstance(m_, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m_.running_mean.normal_(0, 0.1)
                m_.running_var.uniform_(0.5, 1.5)
    xyz_cf = torch.from_numpy(np.ascontiguousarray(xyz.transpose(0, 2, 1)))      # (B,3,N)
    pts_cf = torch.from_numpy(np.ascontiguousarray(pts6.transpose(0, 2, 1)))     # (B,6,N)
    with torch.no_grad():
        sa_xyz, sa_feat = sa(xyz_cf, pts_cf)
        ssg_xyz, ssg_feat = ssg(xyz_cf, pts_cf)
        fp_out = fp(xyz_cf, sa_xyz, pts_cf, sa_feat)
    out["mod_xyz_cf"], out["mod_pts_cf"] = xyz_cf.numpy(), pts_cf.numpy()
    out["mod_sa_xyz"], out["mod_sa_feat"] = sa_xyz.numpy(), sa_feat.numpy()
    out["mod_ssg_xyz"], out["mod_ssg_feat"] = ssg_xyz.numpy(), ssg_feat.numpy()
    out["mod_fp_out"] = fp_out.numpy()
    torch.save({"sa": sa.state_dict(), "ssg": ssg.state_dict(), "fp": fp.state_dict()},
               os.path.join(HERE, "module_weights.pt"))

    np.savez_compressed(os.path.join(HERE, "reference_cpu.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "reference_cpu.npz"))
    print(f"wrote tests/golden/reference_cpu.npz ({sz/1e6:.2f} MB, {len(out)} arrays) and module_weights.pt")

    # ---- oracle-generated regression vectors for the CUDA-only ops (PARITY UNPINNED) ---------
    reg = {}
    rng = np.random.default_rng(99)
    pxyz = np.concatenate([synth.arch_cloud(1500, seed=5, with_normals=False),
                           synth.uniform_cloud(700, seed=6),
                           synth.lattice_cloud(6, dup=20, seed=8)], axis=0)
    offset = np.array([1500, 2200, 2200 + 236], dtype=np.int32)
    new_offset = np.array([375, 550, 609], dtype=np.int32)
    fidx = O.furthestsampling(pxyz, offset, new_offset)
    fidx_cc = O.furthestsampling(pxyz, offset, new_offset, mode=3)
    reg["p_fps_idx_tree"] = O.furthestsampling(pxyz, offset, new_offset, mode=2)
    reg["p_xyz"], reg["p_offset"], reg["p_new_offset"] = pxyz, offset, new_offset
    reg["p_fps_idx"], reg["p_fps_idx_cudacompat"] = fidx, fidx_cc
    q = pxyz[fidx.astype(np.int64)]
    kidx, kdist = O.knnquery(16, pxyz, q, offset, new_offset)
    reg["p_knn_idx"], reg["p_knn_dist"] = kidx, kdist
    kidx_self, kdist_self = O.knnquery(8, pxyz, None, offset, offset)
    reg["p_knn_self_idx"], reg["p_knn_self_dist"] = kidx_self, kdist_self
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), **reg)
    print("wrote tests/golden/oracle_regression.npz (oracle-generated; unpinned ops)")


if __name__ == "__main__":
    main()
