"""GPU (-m gpu): HIP kernels, called through the C ABI, against
  (1) the committed golden fixtures produced by the reference's own torch functions,
  (2) the CPU oracle on the same seeded inputs,
  (3) the reference's own *_cuda_kernel.cu compiled for gfx950 (oracle/_ref), when present.
Indices are compared bit-exactly; fp32 features within 1e-5 (they are copies/differences: expect 0)."""
import numpy as np
import pytest
import torch

from toothgroupnetwork_amd import synth

pytestmark = pytest.mark.gpu


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


# ------------------------------------------------------------------------------------------- FPS
def test_fps_dense_matches_golden(dev, golden):
    from toothgroupnetwork_amd import pointnet2_utils as U
    for k in ("arch", "uniform", "lattice"):
        xyz = T(golden[f"fps_{k}_xyz"], dev)
        ref = golden[f"fps_{k}_idx"].astype(np.int64)
        got = U.farthest_point_sample(xyz, ref.shape[1])
        assert got.dtype == torch.int64 and tuple(got.shape) == ref.shape
        assert np.array_equal(got.cpu().numpy(), ref), k


@pytest.mark.parametrize("n,m", [(1, 1), (2, 2), (63, 10), (64, 64), (65, 30), (1000, 256), (1024, 256), (1025, 100),
                                 (2047, 300), (2048, 512), (2049, 300), (4096, 1024), (6000, 1500), (8193, 400),
                                 (12288, 300), (16385, 200), (24000, 2048), (28000, 300)])
@pytest.mark.parametrize("bucket_min", [2048, 1000000])
def test_fps_every_kernel_shape_vs_oracle(dev, oracle, n, m, bucket_min):
    """One cloud per launch configuration (threads x points-per-lane), incl. exact-capacity edges; every shape
    of the bucket-skipping kernel (bucket_min=2048) and of the plain kernel (bucket_min huge)."""
    from toothgroupnetwork_amd import _lib, pointops as P
    xyz = synth.uniform_cloud(n, seed=n)
    off, noff = np.array([n], np.int32), np.array([m], np.int32)
    with _lib.tuning(fps_bucket_min=bucket_min):
        got = P.furthestsampling(T(xyz, dev), T(off, dev), T(noff, dev))
    assert got.dtype == torch.int32
    assert np.array_equal(got.cpu().numpy(), oracle.furthestsampling(xyz, off, noff))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_fps_bucket_kernel_ties_nan_and_modes(dev, oracle, mode):
    """Clouds large enough for the bucket-skipping kernel: exact ties (lattice + duplicated vertices), NaN
    coordinates, ragged packed batch, all four arithmetic / tie-order modes, against the oracle."""
    from toothgroupnetwork_amd import _lib
    flags = (_lib.FPS_FMA if mode & 1 else 0) | (_lib.FPS_TREE_TIES if mode & 2 else 0)
    lat = synth.lattice_cloud(16, dup=300, seed=7)                 # 4396 points, many exact ties
    arch = synth.arch_cloud(9000, 3, False)
    arch[[5, 4000, 8999]] = np.nan
    uni = synth.uniform_cloud(2500, 9)
    xyz_np = np.concatenate([lat, arch, uni])
    off_np = np.cumsum([lat.shape[0], 9000, 2500]).astype(np.int32)
    noff_np = np.cumsum([1500, 1200, 2500]).astype(np.int32)       # last cloud: every point gets sampled
    xyz, off, noff = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev)
    out = torch.empty(int(noff_np[-1]), dtype=torch.int32, device=dev)
    with _lib.tuning(fps_bucket_min=2048):                          # force the bucket kernel for these small clouds
        _lib.check(_lib.lib().tgn_furthestsampling(3, 9000, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), None,
                                                   _lib.ptr(out), None, flags, _lib.stream()))
    assert np.array_equal(out.cpu().numpy(), oracle.furthestsampling(xyz_np, off_np, noff_np, mode=mode))


def test_fps_bucket_kernel_equals_plain_kernel(dev):
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U
    xyz = T(np.stack([synth.arch_cloud(24000, s, False) for s in (40, 41)]), dev)
    a = U.farthest_point_sample(xyz, 4096)
    with _lib.tuning(fps_plain=1):
        b = U.farthest_point_sample(xyz, 4096)
    assert torch.equal(a, b)


def test_fps_low_valu_hint_changes_the_kernel_not_the_result(dev, oracle):
    """TGN_FPS_LOW_VALU (include/tgn_pointops.h): clouds of 2048-4096 points on the bucket-skipping kernel -- a scheduling
    hint of the phased HotPath schedule; indices and coordinates must be those of the default launch and of the oracle."""
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    for n, m in [(4096, 1024), (3000, 700), (2048, 2048)]:
        xyz_np = np.stack([synth.arch_cloud(n, s, False) for s in (50, 51, 52)])
        xyz = T(xyz_np, dev)
        outs = []
        for flags in (_lib.FPS_LOCAL_INDEX, _lib.FPS_LOCAL_INDEX | _lib.FPS_LOW_VALU):
            idx = torch.empty(3, m, dtype=torch.int32, device=dev)
            nx = torch.empty(3, m, 3, dtype=torch.float32, device=dev)
            _lib.check(L.tgn_furthestsampling_dense(3, n, m, _lib.ptr(xyz), None, _lib.ptr(idx), _lib.ptr(nx), flags, _lib.stream()), "fps")
            outs.append((idx.cpu().numpy(), nx.cpu().numpy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]), (n, m)
        assert np.array_equal(outs[1][0], oracle.farthest_point_sample(xyz_np, m)), (n, m)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_fps_lean_kernel_every_shape_ties_nan_ragged(dev, oracle, mode):
    """fps_lean_kernel (csrc/fps.hip: packed-pair arithmetic, value-only arg-max, coordinates in the wave records) at every
    shape it has -- 64x8, 256x4, 256x8, 512x8 -- on one ragged packed batch: lattices with duplicated vertices (exact ties
    inside a lane, across lanes and across waves), NaN coordinates, a cloud sampled to exhaustion, exact-capacity sizes;
    plain and FMA arithmetic, first-index and tree tie order; against the oracle, against the kernel it replaces, and the
    coordinates it emits."""
    from toothgroupnetwork_amd import _lib
    flags = (_lib.FPS_FMA if mode & 1 else 0) | (_lib.FPS_TREE_TIES if mode & 2 else 0)
    clouds = [synth.lattice_cloud(7, dup=90, seed=1),             # 433 points -> 64 x 8
              synth.lattice_cloud(9, dup=250, seed=2),            # 979       -> 256 x 4
              synth.arch_cloud(1024, 5, False),                   # exact capacity of 256 x 4
              synth.lattice_cloud(12, dup=300, seed=3),           # 2028      -> 256 x 8
              synth.arch_cloud(3000, 6, False),                   # 512 x 8, NaN below
              synth.uniform_cloud(300, 7),                        # sampled to exhaustion and beyond
              synth.arch_cloud(4096, 8, False),                   # exact capacity of 512 x 8
              np.repeat(synth.uniform_cloud(37, 9), 8, 0)]        # 296 points, every vertex eight times
    clouds[4][[0, 17, 2999]] = np.nan
    clouds[1][500] = np.nan
    ms = [433, 600, 256, 1500, 700, 340, 1024, 100]
    for n_max_only, sel in ((512, [0, 5, 7]), (1024, [1, 2, 0]), (2048, [3, 1, 5]), (4096, [4, 6, 3, 0, 7])):
        xyz_np = np.concatenate([clouds[i] for i in sel])
        off_np = np.cumsum([clouds[i].shape[0] for i in sel]).astype(np.int32)
        noff_np = np.cumsum([ms[i] for i in sel]).astype(np.int32)
        n_max = max(clouds[i].shape[0] for i in sel)
        assert n_max <= n_max_only
        xyz, off, noff = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev)
        outs = {}
        for lean in (2, 0):
            idx = torch.full((int(noff_np[-1]),), -7, dtype=torch.int32, device=dev)
            nx = torch.full((int(noff_np[-1]), 3), -7.0, device=dev)
            with _lib.tuning(fps_lean=lean, fps_bucket_min=1000000):
                _lib.check(_lib.lib().tgn_furthestsampling(len(sel), n_max, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), None,
                                                           _lib.ptr(idx), _lib.ptr(nx), flags, _lib.stream()))
            outs[lean] = (idx.cpu().numpy(), nx.cpu().numpy())
        want = oracle.furthestsampling(xyz_np, off_np, noff_np, mode=mode)
        assert np.array_equal(outs[2][0], want), (n_max_only, mode)
        assert np.array_equal(outs[0][0], want), (n_max_only, mode)
        assert np.array_equal(outs[2][1], xyz_np[want.astype(np.int64)], equal_nan=True), (n_max_only, mode)
        assert np.array_equal(outs[2][1], outs[0][1], equal_nan=True)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_fps_throughput_form_equals_the_register_resident_kernel_and_the_oracle(dev, oracle, mode):
    """TGN_FPS_THROUGHPUT (include/tgn_pointops.h): clouds of 4 097 - 32 768 points on the owner-wave kernel out of the L2-resident
    workspace, four workgroups per CU -- a scheduling choice for large batches.  Ragged packed batch with exact ties (lattice +
    duplicated vertices), NaN coordinates, a cloud at the 32 768-point limit; all four arithmetic / tie-order modes; the result must be
    the oracle's and the register-resident kernel's, indices and coordinates."""
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    flags = (_lib.FPS_FMA if mode & 1 else 0) | (_lib.FPS_TREE_TIES if mode & 2 else 0)
    lat = synth.lattice_cloud(18, dup=400, seed=7)                 # 6232 points, many exact ties
    arch = synth.arch_cloud(24000, 3, False)
    arch[[5, 4000, 23999]] = np.nan
    big = synth.arch_cloud(32768, 4, False)
    small = synth.uniform_cloud(4500, 9)
    clouds, ms = [lat, arch, big, small], [1500, 4096, 700, 4500]
    xyz_np = np.concatenate(clouds)
    off_np = np.cumsum([c.shape[0] for c in clouds]).astype(np.int32)
    noff_np = np.cumsum(ms).astype(np.int32)
    n_max = max(c.shape[0] for c in clouds)
    xyz, off, noff = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev)
    nbytes = int(L.tgn_fps_throughput_workspace_bytes(len(clouds), n_max))
    assert nbytes >= 20 * len(clouds) * n_max
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    outs = {}
    for name, extra in (("l2", _lib.FPS_THROUGHPUT), ("plain", 0)):
        idx = torch.full((int(noff_np[-1]),), -7, dtype=torch.int32, device=dev)
        nx = torch.full((int(noff_np[-1]), 3), -7.0, device=dev)
        _lib.check(L.tgn_furthestsampling_ws(len(clouds), n_max, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), _lib.ptr(ws), nbytes,
                                             _lib.ptr(idx), _lib.ptr(nx), flags | extra, _lib.stream()))
        outs[name] = (idx.cpu().numpy(), nx.cpu().numpy())
    want = oracle.furthestsampling(xyz_np, off_np, noff_np, mode=mode)
    assert np.array_equal(outs["l2"][0], want), mode
    assert np.array_equal(outs["plain"][0], want), mode
    assert np.array_equal(outs["l2"][1], xyz_np[want.astype(np.int64)], equal_nan=True)
    assert np.array_equal(outs["l2"][1], outs["plain"][1], equal_nan=True)
    assert int(L.tgn_fps_throughput_workspace_bytes(4, 4096)) == 0 and int(L.tgn_fps_throughput_workspace_bytes(4, 32769)) == 0


def test_farthest_point_sample_takes_the_throughput_form_for_large_batches(dev, oracle):
    """pointnet2_utils.farthest_point_sample with three or more clouds per CU: the L2-resident form (same indices as the oracle)"""
    from toothgroupnetwork_amd import pointnet2_utils as U
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    B = 3 * cus
    base = np.stack([synth.arch_cloud(4200, 60 + i, False) for i in range(4)])
    xyz_np = np.tile(base, (B // 4 + 1, 1, 1))[:B]
    got = U.farthest_point_sample(T(xyz_np, dev), 96).cpu().numpy()
    want = oracle.farthest_point_sample(base, 96)
    assert np.array_equal(got, want[np.arange(B) % 4])


def test_fps_packed_ragged_batch_and_modes(dev, oracle, regression):
    from toothgroupnetwork_amd import _lib, pointops as P
    r = regression
    xyz, off, noff = T(r["p_xyz"], dev), T(r["p_offset"], dev), T(r["p_new_offset"], dev)
    idx, new_xyz = P.fps_with_coords(xyz, off, noff)
    assert np.array_equal(idx.cpu().numpy(), r["p_fps_idx"])
    assert np.array_equal(new_xyz.cpu().numpy(), r["p_xyz"][r["p_fps_idx"].astype(np.int64)])
    idx_cc, _ = P.fps_with_coords(xyz, off, noff, cuda_compat=True)
    assert np.array_equal(idx_cc.cpu().numpy(), r["p_fps_idx_cudacompat"])
    # tree tie order alone (the reference source without contraction)
    L = _lib.lib()
    out = torch.empty(int(r["p_new_offset"][-1]), dtype=torch.int32, device=dev)
    n_max = int(np.diff(np.concatenate([[0], r["p_offset"]])).max())
    _lib.check(L.tgn_furthestsampling(3, n_max, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), None, _lib.ptr(out),
                                      None, _lib.FPS_TREE_TIES, _lib.stream()))
    assert np.array_equal(out.cpu().numpy(), r["p_fps_idx_tree"])


def test_fps_edge_cases(dev, oracle):
    from toothgroupnetwork_amd import pointops as P
    # more samples than points, single point clouds, duplicated vertices, empty batch
    xyz = np.concatenate([synth.uniform_cloud(5, 1), synth.uniform_cloud(1, 2), np.repeat(synth.uniform_cloud(3, 3), 4, 0)])
    off = np.array([5, 6, 18], np.int32)
    noff = np.array([9, 12, 22], np.int32)
    got = P.furthestsampling(T(xyz, dev), T(off, dev), T(noff, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.furthestsampling(xyz, off, noff))
    e = P.furthestsampling(torch.zeros(0, 3, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                           torch.zeros(0, dtype=torch.int32, device=dev))
    assert e.numel() == 0
    # NaN coordinates never update a distance: same sequence as the oracle
    xyz = synth.uniform_cloud(300, 7)
    xyz[17] = np.nan
    o, m = np.array([300], np.int32), np.array([40], np.int32)
    assert np.array_equal(P.furthestsampling(T(xyz, dev), T(o, dev), T(m, dev)).cpu().numpy(),
                          oracle.furthestsampling(xyz, o, m))


@pytest.mark.parametrize("mode", [0, 3])
def test_fps_large_cloud_workspace_kernel(dev, oracle, mode):
    """Clouds beyond the register capacity (raw scans): bucket kernel over the cell-sorted workspace, ragged
    batch with ties / NaN, both against the oracle; and the legacy tmp-only entry (streaming kernel)."""
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    cap = L.tgn_fps_resident_capacity()
    a = synth.arch_cloud(cap + 1500, 5, False)
    a[[7, 20000]] = np.nan
    bcl = np.concatenate([synth.lattice_cloud(30, dup=3000, seed=2), synth.uniform_cloud(5000, 3)])   # 35000 pts, ties
    c = synth.uniform_cloud(1000, 4)                                                                 # small cloud in the same batch
    xyz_np = np.concatenate([a, bcl, c])
    off_np = np.cumsum([a.shape[0], bcl.shape[0], c.shape[0]]).astype(np.int32)
    noff_np = np.cumsum([700, 900, 300]).astype(np.int32)
    n_max = int(max(a.shape[0], bcl.shape[0]))
    flags = (_lib.FPS_FMA if mode & 1 else 0) | (_lib.FPS_TREE_TIES if mode & 2 else 0)
    xyz, off, noff = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev)
    ref = oracle.furthestsampling(xyz_np, off_np, noff_np, mode=mode)
    nbytes = int(L.tgn_fps_workspace_bytes(3, n_max))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(int(noff_np[-1]), dtype=torch.int32, device=dev)
    _lib.check(L.tgn_furthestsampling_ws(3, n_max, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), _lib.ptr(ws), nbytes,
                                         _lib.ptr(out), None, flags, _lib.stream()))
    assert np.array_equal(out.cpu().numpy(), ref)
    tmp = torch.empty(xyz_np.shape[0], dtype=torch.float32, device=dev)     # reference-style tmp: streaming kernel
    out2 = torch.empty_like(out)
    _lib.check(L.tgn_furthestsampling(3, n_max, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), _lib.ptr(tmp),
                                      _lib.ptr(out2), None, flags, _lib.stream()))
    assert np.array_equal(out2.cpu().numpy(), ref)


@pytest.mark.parametrize("n,mode", [(70000, 0), (150000, 3), (250000, 0), (262144, 1)])
def test_fps_large_cloud_every_metadata_group_count(dev, oracle, n, mode):
    """The large-cloud kernel keeps 64 buckets per wave and register group: 1, 2, 3 and 4 groups (up to its 262 144-point
    limit), duplicated vertices and NaN coordinates included, next to a small cloud in the same launch."""
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    a = synth.arch_cloud(n, 11, False)
    a[5000:5400] = a[100:500]                       # exact ties between buckets and waves
    a[[3, n // 2, n - 1]] = np.nan
    c = synth.uniform_cloud(3000, 4)
    xyz_np = np.concatenate([a, c])
    off_np = np.cumsum([n, c.shape[0]]).astype(np.int32)
    noff_np = np.cumsum([2500, 400]).astype(np.int32)
    flags = (_lib.FPS_FMA if mode & 1 else 0) | (_lib.FPS_TREE_TIES if mode & 2 else 0)
    xyz, off, noff = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev)
    ref = oracle.furthestsampling(xyz_np, off_np, noff_np, mode=mode)
    nbytes = int(L.tgn_fps_workspace_bytes(2, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty(int(noff_np[-1]), dtype=torch.int32, device=dev)
    nx = torch.empty(int(noff_np[-1]), 3, dtype=torch.float32, device=dev)
    _lib.check(L.tgn_furthestsampling_ws(2, n, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff), _lib.ptr(ws), nbytes,
                                         _lib.ptr(out), _lib.ptr(nx), flags, _lib.stream()))
    got = out.cpu().numpy()
    assert np.array_equal(got, ref)
    assert np.array_equal(nx.cpu().numpy(), xyz_np[ref], equal_nan=True)


def test_fps_oversized_cloud_through_python_api(dev, oracle):
    from toothgroupnetwork_amd import _lib, pointops as P
    n = _lib.lib().tgn_fps_resident_capacity() + 1500
    xyz = synth.arch_cloud(n, 5, False)
    o, m = np.array([n], np.int32), np.array([96], np.int32)
    got = P.furthestsampling(T(xyz, dev), T(o, dev), T(m, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.furthestsampling(xyz, o, m))


def test_fps_vs_reference_kernel_on_this_gpu(dev, regression):
    """The reference's sampling_cuda_kernel.cu (compiled for gfx950, no contraction) == TREE_TIES mode."""
    from oracle import ref_gpu
    if not ref_gpu.available():
        pytest.skip("oracle/_ref not built")
    from toothgroupnetwork_amd import _lib
    cases = [(regression["p_xyz"], regression["p_offset"], regression["p_new_offset"]),
             (synth.lattice_cloud(12, dup=100, seed=3), np.array([1828], np.int32), np.array([700], np.int32)),
             (synth.arch_cloud(24000, 9, False), np.array([24000], np.int32), np.array([1024], np.int32))]
    for xyz_np, off_np, noff_np in cases:
        xyz, off, noff = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev)
        ref = ref_gpu.furthestsampling(xyz, off, noff)
        out = torch.empty_like(ref)
        n_max = int(np.diff(np.concatenate([[0], off_np])).max())
        _lib.check(_lib.lib().tgn_furthestsampling(len(off_np), n_max, _lib.ptr(xyz), _lib.ptr(off), _lib.ptr(noff),
                                                   None, _lib.ptr(out), None, _lib.FPS_TREE_TIES, _lib.stream()))
        assert torch.equal(out, ref)


def test_fps_tree_tie_order_is_selectable_from_python_and_matches_the_reference_kernel(dev, regression):
    """set_fps_mode(ties='tree') / TGN_FPS_TIES=tree make pointops.furthestsampling, pointnet2_utils.farthest_point_sample,
    gen_utils-style resampling and the pointops_cuda shim sample the way the reference's own kernel does
    (sampling_cuda_kernel.cu:5-10,64-123, compiled for gfx950 as oracle/_ref) -- duplicated vertices included."""
    from oracle import ref_gpu
    if not ref_gpu.available():
        pytest.skip("oracle/_ref not built")
    import pointops_cuda
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U, pointops as P, resample
    dup = synth.lattice_cloud(12, dup=100, seed=3)                      # exact distance ties
    scan = synth.arch_cloud(24000, 9, False)
    scan[5000:5400] = scan[100:500]                                     # duplicated vertices, as raw scans have
    prev = _lib.set_fps_mode(ties="tree")
    try:
        assert _lib.get_fps_mode() == ("tree", False) and _lib.lib().tgn_get_fps_mode() == _lib.FPS_TREE_TIES
        differs = False
        for xyz_np, m in ((dup, 700), (scan, 2048)):
            n = xyz_np.shape[0]
            xyz, off, noff = T(xyz_np, dev), T(np.array([n], np.int32), dev), T(np.array([m], np.int32), dev)
            ref = ref_gpu.furthestsampling(xyz, off, noff)
            assert torch.equal(P.furthestsampling(xyz, off, noff), ref)
            assert torch.equal(U.farthest_point_sample(xyz[None], m)[0], ref.long())
            assert np.array_equal(resample.fps(xyz_np.astype(np.float64), m), ref.cpu().numpy())
            idx = torch.zeros(m, dtype=torch.int32, device=dev)
            pointops_cuda.furthestsampling_cuda(1, n, xyz, off, noff, torch.full((n,), 1e10, device=dev), idx)
            assert torch.equal(idx, ref)
            _lib.set_fps_mode(ties="first")
            differs |= not torch.equal(P.furthestsampling(xyz, off, noff), ref)
            _lib.set_fps_mode(ties="tree")
        assert differs, "the tie order must matter on clouds with duplicated vertices"
    finally:
        _lib.set_fps_mode(*prev)
    assert _lib.lib().tgn_get_fps_mode() == _lib.fps_flags()


# ------------------------------------------------------------------------------ square_distance
def test_square_distance_matches_golden_bit_exact(dev, golden):
    from toothgroupnetwork_amd import pointnet2_utils as U
    got = U.square_distance(T(golden["sqd_src"], dev), T(golden["sqd_dst"], dev))
    assert np.array_equal(got.cpu().numpy(), golden["sqd_out"])


def test_square_distance_gradients(dev):
    from toothgroupnetwork_amd import pointnet2_utils as U
    a = torch.randn(2, 17, 3, device=dev, requires_grad=True)
    b = torch.randn(2, 29, 3, device=dev, requires_grad=True)
    g = torch.randn(2, 17, 29, device=dev)
    U.square_distance(a, b).backward(g)
    a2, b2 = a.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ((a2[:, :, None, :] - b2[:, None, :, :]) ** 2).sum(-1).backward(g)
    torch.testing.assert_close(a.grad, a2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.grad, b2.grad, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------- ball query
def test_ball_query_matches_golden(dev, golden):
    from toothgroupnetwork_amd import pointnet2_utils as U
    xyz, new_xyz = T(golden["ball_xyz"], dev), T(golden["ball_new_xyz"], dev)
    for ri in range(4):
        radius, ns = golden[f"ball_{ri}_cfg"]
        got = U.query_ball_point(float(radius), int(ns), xyz, new_xyz)
        assert got.dtype == torch.int64
        assert np.array_equal(got.cpu().numpy(), golden[f"ball_{ri}_idx"].astype(np.int64)), radius


@pytest.mark.parametrize("kind", ["arch", "uniform", "lattice"])
def test_ball_query_vs_oracle(dev, oracle, kind):
    from toothgroupnetwork_amd import pointnet2_utils as U
    if kind == "lattice":
        xyz = np.stack([synth.lattice_cloud(10, dup=50, seed=s) for s in (0, 1)])
        radii = [(0.2223, 16), (0.45, 64)]  # radius on lattice spacings: boundary ties
    else:
        gen = (lambda s: synth.arch_cloud(5000, s, False)) if kind == "arch" else (lambda s: synth.uniform_cloud(5000, s))
        xyz = np.stack([gen(s) for s in (0, 1, 2)])
        radii = [(0.05, 32), (0.1, 32), (0.2, 64), (2.5, 8)]
    q = xyz[:, ::7][:, :300]
    for radius, ns in radii:
        got = U.query_ball_point(radius, ns, T(xyz, dev), T(q, dev)).cpu().numpy()
        assert np.array_equal(got, oracle.query_ball_point(radius, ns, xyz, q)), (kind, radius)


def test_ball_query_edge_cases(dev, oracle):
    from toothgroupnetwork_amd import pointnet2_utils as U
    xyz = synth.uniform_cloud(130, 3)[None]
    far = np.full((1, 3, 3), 40.0, np.float32)
    got = U.query_ball_point(0.1, 5, T(xyz, dev), T(far, dev)).cpu().numpy()
    assert (got == 130).all()  # no hit at all -> N, like the reference
    # S = 0, nsample > N, N = 1
    assert U.query_ball_point(0.1, 5, T(xyz, dev), torch.zeros(1, 0, 3, device=dev)).shape == (1, 0, 5)
    got = U.query_ball_point(10.0, 200, T(xyz, dev), T(xyz[:, :4], dev)).cpu().numpy()
    assert np.array_equal(got, oracle.query_ball_point(10.0, 200, xyz, xyz[:, :4]))
    one = xyz[:, :1]
    assert np.array_equal(U.query_ball_point(0.1, 3, T(one, dev), T(one, dev)).cpu().numpy(), np.zeros((1, 1, 3), np.int64))


def test_ball_query_dense_balls_and_large_clouds(dev, oracle):
    """The bitmap-selection kernel: more hits than its hit list holds (256: the two-pass path), more than 64 hits (two
    rounds of ranking), the largest cloud it takes (32 768 points) and the first one it does not (rank-select kernel)."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    for n, radius, ns in [(16384, 0.2, 32), (16384, 0.2, 64), (16384, 0.12, 128), (32768, 0.06, 32), (32769, 0.06, 32)]:
        xyz = synth.uniform_cloud(n, 7)[None] * 0.5 + 0.5    # unit cube: ~n * 4.19 r^3 hits per query (550 at r = 0.2)
        q = xyz[:, ::97][:, :160]
        got = U.query_ball_point(radius, ns, T(xyz, dev), T(q, dev)).cpu().numpy()
        want = oracle.query_ball_point(radius, ns, xyz, q)
        assert np.array_equal(got, want), (n, radius, ns)


@pytest.mark.parametrize("variant", [2, 1, 0])
def test_ball_query_every_grid_kernel_ragged_chunks_and_bad_queries(dev, oracle, variant):
    """The three grid kernels (2: chunks of 16 queries, the default; 1: round 2's bitmap kernel; 0: rank-select) on shapes that
    exercise the chunking: S not a multiple of 16, fewer queries than a chunk, many clouds (chunks never straddle two), every
    bitmap size (N <= 8192, 16384, 24576, 32768), NaN / Inf queries in the middle of a chunk (index-order scan for those
    queries only), a NaN point in one cloud of the batch (that cloud takes the scan path, its neighbours do not), K = 256."""
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U
    with _lib.tuning(ball_bitmap=variant):
        for B, n, S, radius, ns in [(3, 4096, 37, 0.1, 32), (9, 3000, 5, 0.15, 16), (2, 9000, 100, 0.08, 32), (1, 24000, 333, 0.05, 32),
                                    (2, 20000, 64, 0.06, 64), (1, 32768, 130, 0.05, 32), (11, 2500, 17, 0.3, 256)]:
            xyz = np.stack([synth.arch_cloud(n, 40 + b, False) for b in range(B)])
            q = np.ascontiguousarray(xyz[:, :: max(n // S, 1)][:, :S]).copy()
            assert q.shape[1] == S
            got = U.query_ball_point(radius, ns, T(xyz, dev), T(q, dev)).cpu().numpy()
            assert np.array_equal(got, oracle.query_ball_point(radius, ns, xyz, q)), (variant, B, n, S, "plain")
            if S >= 5:
                q[0, 1, 0] = np.nan
                q[B - 1, S - 2, 2] = np.inf
                q[0, 3] = 7.5                              # far outside the cloud: no hit -> N
                got = U.query_ball_point(radius, ns, T(xyz, dev), T(q, dev)).cpu().numpy()
                assert np.array_equal(got, oracle.query_ball_point(radius, ns, xyz, q)), (variant, B, n, S, "bad queries")
            if B >= 2:
                xyz[1, n // 2, 1] = np.nan
                got = U.query_ball_point(radius, ns, T(xyz, dev), T(q, dev)).cpu().numpy()
                assert np.array_equal(got, oracle.query_ball_point(radius, ns, xyz, q)), (variant, B, n, S, "NaN point")


def test_ball_query_int32_rows_of_the_hot_path_equal_the_int64_rows(dev):
    """the C entry point with idx_is_int64 = 0 (what hotpath.HotPath uses) against the int64 rows of the Python operator, through the
    split build / prebuilt entry points, at the headline's level-1 and level-2 shapes"""
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U
    L = _lib.lib()
    for B, n, S, radius, ns in [(4, 24000, 4096, 0.05, 32), (4, 4096, 1024, 0.1, 32)]:
        xyz = T(np.stack([synth.arch_cloud(n, 70 + b, False) for b in range(B)]), dev)
        q = xyz[:, torch.randperm(n, generator=torch.Generator().manual_seed(1))[:S].to(dev)].contiguous()
        want = U.query_ball_point(radius, ns, xyz, q)
        nbytes = int(L.tgn_ball_query_workspace_bytes(B, n, S))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        got = torch.empty(B, S, ns, dtype=torch.int32, device=dev)
        r2 = float(torch.tensor(radius ** 2, dtype=torch.float32).item())
        st = _lib.stream()
        _lib.check(L.tgn_ball_query_build(B, n, S, ns, r2, _lib.ptr(xyz), _lib.ptr(ws), nbytes, st), "build")
        _lib.check(L.tgn_ball_query_prebuilt(B, n, S, ns, r2, _lib.ptr(xyz), _lib.ptr(q), _lib.ptr(got), 0, _lib.ptr(ws), nbytes, st), "query")
        assert torch.equal(got.long(), want), (n, S)


# ------------------------------------------------------------------- grouping / index_points / SA
def test_sample_and_group_matches_golden(dev, golden):
    from toothgroupnetwork_amd import pointnet2_utils as U
    xyz, pts = T(golden["ball_xyz"], dev), T(golden["sag_points"], dev)
    new_xyz, new_points = U.sample_and_group(128, 0.1, 16, xyz, pts)
    assert np.array_equal(new_xyz.cpu().numpy(), golden["sag_new_xyz"])
    np.testing.assert_allclose(new_points.cpu().numpy(), golden["sag_new_points"], rtol=0, atol=1e-5)
    assert np.array_equal(new_points.cpu().numpy(), golden["sag_new_points"])  # copies and differences: exact
    ax, ap = U.sample_and_group_all(xyz, pts)
    assert np.array_equal(ap.cpu().numpy()[:, :, :64], golden["saga_new_points"])


def test_group_points_orders_and_index_points(dev, oracle):
    from toothgroupnetwork_amd import pointnet2_utils as U
    rng = np.random.default_rng(0)
    B, N, S, K, D = 2, 500, 40, 9, 13
    xyz = rng.normal(size=(B, N, 3)).astype(np.float32)
    pts = rng.normal(size=(B, N, D)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    idx = rng.integers(0, N, size=(B, S, K))
    for xyz_first in (True, False):
        got = U.group_points(T(xyz, dev), T(new_xyz, dev), T(pts, dev), T(idx, dev), xyz_first=xyz_first)
        assert np.array_equal(got.cpu().numpy(), oracle.group_points(xyz, new_xyz, pts, idx, xyz_first))
    got = U.group_points(T(xyz, dev), T(new_xyz, dev), None, T(idx, dev))
    assert np.array_equal(got.cpu().numpy(), oracle.group_points(xyz, new_xyz, None, idx))
    for shape_idx in (idx[:, :, 0], idx):
        got = U.index_points(T(pts, dev), T(shape_idx, dev))
        assert np.array_equal(got.cpu().numpy(), oracle.index_points(pts, shape_idx))
    # int32 indices are accepted too
    got = U.index_points(T(pts, dev), T(idx.astype(np.int32), dev))
    assert np.array_equal(got.cpu().numpy(), oracle.index_points(pts, idx))


@pytest.mark.parametrize("D,K", [(64, 32), (128, 32), (131, 7), (200, 33), (256, 64), (512, 32), (320, 128), (66, 1),
                                 (61, 4), (62, 8), (125, 36), (129, 24), (509, 32), (1024, 64), (6, 32), (13, 16), (1, 64)])
def test_group_points_wide_rows(dev, oracle, D, K, monkeypatch):
    """every row width class of the grouping kernels: rows shorter and longer than a 64-float sub-step, K*(3+D) a
    multiple of 4 (16-B store kernel) or not (4-B kernel), K > 64, both channel orders, both index types, negative
    (wrapping) indices and an out-of-range one."""
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U
    rng = np.random.default_rng(D * 1000 + K)
    B, N, S = 2, 300, 37
    xyz = rng.normal(size=(B, N, 3)).astype(np.float32)
    pts = rng.normal(size=(B, N, D)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    idx = rng.integers(0, N, size=(B, S, K))
    for xyz_first in (True, False):
        want = oracle.group_points(xyz, new_xyz, pts, idx, xyz_first)
        for dt in (np.int64, np.int32):
            got = U.group_points(T(xyz, dev), T(new_xyz, dev), T(pts, dev), T(idx.astype(dt), dev), xyz_first=xyz_first)
            assert np.array_equal(got.cpu().numpy(), want)
    assert _lib.lib().tgn_take_index_error(_lib.stream()) == 0
    # negative indices wrap like torch's advanced indexing (reference pointnet2_utils.py:56-60)
    neg = idx.copy()
    neg[0, 3, 0] -= N
    neg[1, 7, K - 1] -= N
    got = U.group_points(T(xyz, dev), T(new_xyz, dev), T(pts, dev), T(neg, dev))
    assert np.array_equal(got.cpu().numpy(), oracle.group_points(xyz, new_xyz, pts, idx, True))
    # out of range: IndexError like the reference (the empty-ball marker N)
    bad = idx.copy()
    bad[1, 5, K // 2] = N
    with pytest.raises(IndexError):
        U.group_points(T(xyz, dev), T(new_xyz, dev), T(pts, dev), T(bad, dev))
    assert _lib.lib().tgn_take_index_error(_lib.stream()) == 0      # raising consumed the flag
    # TGN_INDEX_CHECK=off: no sync, no exception; the row is filled from point 0 and the flag stays readable
    monkeypatch.setattr(_lib, "INDEX_CHECK", "off")
    got = U.group_points(T(xyz, dev), T(new_xyz, dev), T(pts, dev), T(bad, dev))
    assert _lib.take_index_error() and not _lib.take_index_error()
    bad[1, 5, K // 2] = 0
    assert np.array_equal(got.cpu().numpy(), oracle.group_points(xyz, new_xyz, pts, bad, True))


@pytest.mark.parametrize("impl,policy", [(1, -1), (2, 0), (2, 2), (2, 16), (2, 17), (2, 18), (7, 16), (7, 0), (7, 2), (10, -1), (0, -1)])
@pytest.mark.parametrize("N,S,K,D", [(4096, 1024, 32, 128), (1024, 256, 32, 512), (3000, 500, 32, 6), (777, 99, 64, 253),
                                     (500, 300, 4, 61), (900, 64, 36, 125), (256, 40, 8, 1021), (700, 90, 64, 1024), (640, 33, 16, 700)])
def test_group_points_every_kernel_variant(dev, oracle, impl, policy, N, S, K, D):
    """tgn_group_points_ex: every kernel a launcher reaches (per element, staged 16-B stores, row pieces, pairs; a kernel
    that does not take the shape falls back) with each store policy, bounded and unbounded grids, must write the same
    bytes as the oracle (sample_and_group, pointnet2_utils.py:162-169)."""
    from toothgroupnetwork_amd import _lib
    rng = np.random.default_rng(N + K + D)
    B = 9
    xyz = rng.normal(size=(B, N, 3)).astype(np.float32)
    pts = rng.normal(size=(B, N, D)).astype(np.float32)
    new_xyz = xyz[:, :S].copy()
    idx = rng.integers(0, N, size=(B, S, K)).astype(np.int32)
    want = oracle.group_points(xyz, new_xyz, pts, idx.astype(np.int64), True)
    L = _lib.lib()
    tx, tn, tp, ti = T(xyz, dev), T(new_xyz, dev), T(pts, dev), T(idx, dev)
    for max_blocks in (0, 256, 8):
        out = torch.full((B, S, K, 3 + D), float("nan"), device=dev)
        _lib.check(L.tgn_group_points_ex(B, N, S, K, D, _lib.ptr(tx), _lib.ptr(tn), _lib.ptr(tp), _lib.ptr(ti), 0, 1,
                                         _lib.ptr(out), impl, policy, max_blocks, _lib.stream()))
        assert np.array_equal(out.cpu().numpy(), want), (impl, policy, max_blocks)


def test_group_points_empty_ball_raises_like_the_reference(dev):
    from toothgroupnetwork_amd import _lib, pointnet2_utils as U
    xyz = T(synth.uniform_cloud(64, 1)[None], dev)
    far = torch.full((1, 1, 3), 30.0, device=dev)
    idx = U.query_ball_point(0.1, 4, xyz, far)          # all N: out of range, the reference raises (:136-141 -> :56-60)
    with pytest.raises(IndexError):
        U.group_points(xyz, far, None, idx)
    with pytest.raises(IndexError):
        U.index_points(xyz, idx)
    with pytest.raises(IndexError):
        U.index_points(xyz, idx - 200)                  # below -N
    assert _lib.lib().tgn_take_index_error(_lib.stream()) == 0
    # index_points wraps negative indices
    neg = torch.tensor([[-1, -64, 5]], device=dev)
    assert torch.equal(U.index_points(xyz, neg), xyz[:, [63, 0, 5]])


def test_grouping_autograd_matches_torch_indexing(dev):
    from toothgroupnetwork_amd import pointnet2_utils as U
    B, N, S, K, D = 2, 200, 16, 6, 5
    xyz = torch.randn(B, N, 3, device=dev, requires_grad=True)
    pts = torch.randn(B, N, D, device=dev, requires_grad=True)
    cidx = torch.randint(0, N, (B, S), device=dev)
    idx = torch.randint(0, N, (B, S, K), device=dev)
    g = torch.randn(B, S, K, 3 + D, device=dev)
    new_xyz = U.index_points(xyz, cidx)
    U.group_points(xyz, new_xyz, pts, idx, xyz_first=True).backward(g)
    x2, p2 = xyz.detach().clone().requires_grad_(), pts.detach().clone().requires_grad_()
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    nx2 = x2[torch.arange(B, device=dev).view(B, 1), cidx]
    ref = torch.cat([x2[bi, idx] - nx2.view(B, S, 1, 3), p2[bi, idx]], -1)
    ref.backward(g)
    torch.testing.assert_close(xyz.grad, x2.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(pts.grad, p2.grad, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------- three_nn / interpolate
def test_three_nn_and_interpolate_match_golden(dev, golden):
    from toothgroupnetwork_amd import pointnet2_utils as U
    d, i = U.three_nn(T(golden["tnn_xyz1"], dev), T(golden["tnn_xyz2"], dev))
    assert np.array_equal(d.cpu().numpy(), golden["tnn_dist"])
    assert np.array_equal(i.cpu().numpy(), golden["tnn_idx"].astype(np.int64))
    out = U.three_interpolate(T(golden["tnn_feat2"], dev), d, i)
    np.testing.assert_allclose(out.cpu().numpy(), golden["tnn_interp"], rtol=0, atol=1e-5)


def test_three_nn_ties_and_small_support(dev, oracle):
    from toothgroupnetwork_amd import pointnet2_utils as U
    xyz2 = synth.lattice_cloud(6, dup=30, seed=2)[None]          # S = 246 with duplicates
    xyz1 = synth.lattice_cloud(7, seed=3)[None]
    d, i = U.three_nn(T(xyz1, dev), T(xyz2, dev))
    od, oi = oracle.three_nn(xyz1, xyz2)
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(i.cpu().numpy(), oi)
    big = synth.uniform_cloud(2500, 4)[None]                      # S > one LDS tile
    d, i = U.three_nn(T(xyz1, dev), T(big, dev))
    od, oi = oracle.three_nn(xyz1, big)
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(i.cpu().numpy(), oi)


@pytest.mark.parametrize("B,N,S", [(1, 70, 17), (3, 333, 255), (2, 1000, 1023), (4, 512, 1024), (1, 64, 16), (2, 40000, 300)])
def test_three_nn_both_kernels_against_the_oracle_with_ties(dev, oracle, B, N, S):
    """few queries (B N <= 32 768, 16 <= S <= 1024): four waves share 64 queries and merge their top threes in wave order; many
    queries: one thread per query.  Both must give the index-ordered scan's answer on lattices with duplicated points (exact ties)."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    rng = np.random.default_rng(B * 1000 + S)
    xyz2 = (rng.integers(-4, 5, size=(B, S, 3)) * 0.125).astype(np.float32)          # coarse lattice: many equal distances, duplicates
    xyz1 = (rng.integers(-8, 9, size=(B, N, 3)) * 0.0625).astype(np.float32)
    d, i = U.three_nn(T(xyz1, dev), T(xyz2, dev))
    od, oi = oracle.three_nn(xyz1, xyz2)
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(i.cpu().numpy(), oi)


def test_three_interpolate_backward(dev):
    from toothgroupnetwork_amd import pointnet2_utils as U
    B, N, S, C = 2, 300, 50, 7
    xyz1, xyz2 = torch.randn(B, N, 3, device=dev), torch.randn(B, S, 3, device=dev)
    f = torch.randn(B, S, C, device=dev, requires_grad=True)
    d, i = U.three_nn(xyz1, xyz2)
    g = torch.randn(B, N, C, device=dev)
    U.three_interpolate(f, d, i).backward(g)
    f2 = f.detach().clone().requires_grad_()
    w = 1.0 / (d + 1e-8)
    w = w / w.sum(2, keepdim=True)
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    (f2[bi, i] * w.unsqueeze(-1)).sum(2).backward(g)
    torch.testing.assert_close(f.grad, f2.grad, rtol=1e-4, atol=1e-5)


# ----------------------------------------------------------------------------------------- kNN
@pytest.mark.parametrize("k", [1, 3, 16, 36, 64, 100])
def test_knn_vs_oracle_heap_order(dev, oracle, regression, k):
    from toothgroupnetwork_amd import pointops as P
    r = regression
    q = r["p_xyz"][r["p_fps_idx"].astype(np.int64)]
    idx, dist = P.knnquery(k, T(r["p_xyz"], dev), T(q, dev), T(r["p_offset"], dev), T(r["p_new_offset"], dev))
    oi, od = oracle.knnquery(k, r["p_xyz"], q, r["p_offset"], r["p_new_offset"])
    assert idx.dtype == torch.int32 and dist.dtype == torch.float32
    assert np.array_equal(idx.cpu().numpy(), oi)          # includes the lattice segment: exact heap tie order
    assert np.array_equal(dist.cpu().numpy(), od)
    if k == 16:
        assert np.array_equal(oi, r["p_knn_idx"])


def test_knn_self_query_none_and_tensor_k(dev, oracle, regression):
    from toothgroupnetwork_amd import pointops as P
    r = regression
    xyz, off = T(r["p_xyz"], dev), T(r["p_offset"], dev)
    idx, dist = P.knnquery(torch.tensor(8, device=dev), xyz, None, off, off)   # 0-d CUDA tensor as k (basic_operators.py:30)
    assert np.array_equal(idx.cpu().numpy(), r["p_knn_self_idx"])
    assert np.array_equal(dist.cpu().numpy(), r["p_knn_self_dist"])


def test_knn_memo_hits_and_invalidates(dev, oracle):
    from toothgroupnetwork_amd import pointops as P
    P.knn_cache_clear()
    xyz_np = synth.uniform_cloud(500, 1)
    xyz = T(xyz_np, dev)
    off = torch.tensor([500], dtype=torch.int32, device=dev)
    a, _ = P.knnquery(8, xyz, xyz, off, off)
    n0 = len(P._KNN_CACHE)
    b, _ = P.knnquery(8, xyz, None, off, off)      # same arguments -> memo hit (blocks.py:34-35 pattern)
    assert len(P._KNN_CACHE) == n0 and torch.equal(a, b)
    a[0, 0] = -7                                     # callers get private copies
    c, _ = P.knnquery(8, xyz, xyz, off, off)
    assert c[0, 0] != -7
    xyz[3] += 1.0                                    # in-place edit bumps the version -> recomputed
    d, _ = P.knnquery(8, xyz, xyz, off, off)
    xyz_np[3] += 1.0
    assert np.array_equal(d.cpu().numpy(), oracle.knnquery(8, xyz_np, xyz_np, [500], [500])[0])


def test_knn_vs_reference_kernel_on_this_gpu(dev, regression):
    from oracle import ref_gpu
    if not ref_gpu.available():
        pytest.skip("oracle/_ref not built")
    from toothgroupnetwork_amd import pointops as P
    r = regression
    xyz, off, noff = T(r["p_xyz"], dev), T(r["p_offset"], dev), T(r["p_new_offset"], dev)
    q = xyz[T(r["p_fps_idx"], dev).long()].contiguous()
    for k in (3, 24):
        ri, rd2 = ref_gpu.knnquery(k, xyz, q, off, noff)
        idx, d2 = P._knn_raw(k, xyz, q, off, noff)
        assert torch.equal(idx, ri) and torch.equal(d2, rd2)


# --------------------------------------------------------------------------- pointops gather ops
def _rand_case(dev, n=300, m=120, ns=9, c=20, wc=5, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    feat = torch.randn(n, c, generator=g)
    idx = torch.randint(0, n, (m, ns), generator=g, dtype=torch.int32)
    return feat.to(dev), idx.to(dev)


def test_grouping_fwd_bwd(dev, oracle):
    from toothgroupnetwork_amd import pointops as P
    feat, idx = _rand_case(dev)
    feat.requires_grad_()
    out = P.grouping(feat, idx)
    assert np.array_equal(out.detach().cpu().numpy(), oracle.grouping_forward(feat.detach().cpu().numpy(), idx.cpu().numpy()))
    g = torch.randn_like(out)
    out.backward(g)
    np.testing.assert_allclose(feat.grad.cpu().numpy(),
                               oracle.grouping_backward(g.cpu().numpy(), idx.cpu().numpy(), feat.shape[0]), atol=1e-5, rtol=1e-5)


def test_subtraction_aggregation_interpolation_fwd_bwd(dev, oracle):
    from toothgroupnetwork_amd import pointops as P
    n, ns, c, wc = 250, 8, 16, 4
    g = torch.Generator().manual_seed(3)
    a = torch.randn(n, c, generator=g).to(dev).requires_grad_()
    b = torch.randn(n, c, generator=g).to(dev).requires_grad_()
    idx = torch.randint(0, n, (n, ns), generator=g, dtype=torch.int32).to(dev)
    out = P.subtraction(a, b, idx)
    A, Bn, I = a.detach().cpu().numpy(), b.detach().cpu().numpy(), idx.cpu().numpy()
    assert np.array_equal(out.detach().cpu().numpy(), oracle.subtraction_forward(A, Bn, I))
    go = torch.randn_like(out)
    out.backward(go)
    g1, g2 = oracle.subtraction_backward(I, go.cpu().numpy())
    np.testing.assert_allclose(a.grad.cpu().numpy(), g1, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g2, atol=1e-5, rtol=1e-5)

    pos = torch.randn(n, ns, c, generator=g).to(dev).requires_grad_()
    w = torch.randn(n, ns, wc, generator=g).to(dev).requires_grad_()
    x = torch.randn(n, c, generator=g).to(dev).requires_grad_()
    out = P.aggregation(x, pos, w, idx)
    X, Pn, W = x.detach().cpu().numpy(), pos.detach().cpu().numpy(), w.detach().cpu().numpy()
    np.testing.assert_allclose(out.detach().cpu().numpy(), oracle.aggregation_forward(X, Pn, W, I), atol=1e-5, rtol=1e-5)
    go = torch.randn_like(out)
    out.backward(go)
    gi, gp, gw = oracle.aggregation_backward(X, Pn, W, I, go.cpu().numpy())
    np.testing.assert_allclose(x.grad.cpu().numpy(), gi, atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(pos.grad.cpu().numpy(), gp, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(w.grad.cpu().numpy(), gw, atol=1e-4, rtol=1e-4)


def test_interpolation_and_queryandgroup_vs_oracle(dev, oracle, regression):
    from toothgroupnetwork_amd import pointops as P
    r = regression
    xyz_np, off_np, noff_np = r["p_xyz"], r["p_offset"], r["p_new_offset"]
    q_np = xyz_np[r["p_fps_idx"].astype(np.int64)]
    rng = np.random.default_rng(5)
    feat_q = rng.normal(size=(q_np.shape[0], 12)).astype(np.float32)      # features on the coarse set
    xyz, off, noff, q = T(xyz_np, dev), T(off_np, dev), T(noff_np, dev), T(q_np, dev)
    for k in (1, 3):
        fq = T(feat_q, dev).requires_grad_()
        out = P.interpolation(q, xyz, fq, noff, off, k=k)                  # coarse -> fine (blocks.py:110)
        ref, oidx, ow = oracle.interpolation(q_np, xyz_np, feat_q, noff_np, off_np, k=k)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, atol=1e-5, rtol=1e-5)
        go = torch.randn_like(out)
        out.backward(go)
        np.testing.assert_allclose(fq.grad.cpu().numpy(), oracle.interpolation_backward(go.cpu().numpy(), oidx, ow, feat_q.shape[0]),
                                   atol=1e-4, rtol=1e-4)
        out2 = P.interpolation2(q, xyz, T(feat_q, dev), noff, off, k)
        np.testing.assert_allclose(out2.cpu().numpy(), ref, atol=1e-5, rtol=1e-5)
    feat = rng.normal(size=(xyz_np.shape[0], 10)).astype(np.float32)
    for use_xyz in (True, False):
        got = P.queryandgroup(16, xyz, q, T(feat, dev), None, off, noff, use_xyz=use_xyz)
        ref = oracle.queryandgroup(16, xyz_np, q_np, feat, None, off_np, noff_np, use_xyz=use_xyz)
        assert np.array_equal(got.cpu().numpy(), ref)
    # gradient of queryandgroup == torch fancy indexing (what the reference relies on, pointops.py:89-95)
    f = T(feat, dev).requires_grad_()
    idx, _ = P.knnquery(16, xyz, q, off, noff)
    go = torch.randn(q.shape[0], 16, 13, device=dev)
    P.queryandgroup(16, xyz, q, f, idx, off, noff).backward(go)
    f2 = T(feat, dev).requires_grad_()
    torch.cat([xyz[idx.long().view(-1)].view(-1, 16, 3) - q.unsqueeze(1), f2[idx.long().view(-1)].view(-1, 16, 10)], -1).backward(go)
    torch.testing.assert_close(f.grad, f2.grad, rtol=1e-5, atol=1e-5)


def test_gather_family_vs_reference_kernels_on_this_gpu(dev):
    from oracle import ref_gpu
    if not ref_gpu.available():
        pytest.skip("oracle/_ref not built")
    from toothgroupnetwork_amd import _lib
    L, S_, p = _lib.lib(), _lib.stream, _lib.ptr
    n, ns, c, wc = 400, 12, 32, 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, c, generator=g).to(dev)
    y = torch.randn(n, c, generator=g).to(dev)
    idx = torch.randint(0, n, (n, ns), generator=g, dtype=torch.int32).to(dev)
    pos = torch.randn(n, ns, c, generator=g).to(dev)
    w = torch.randn(n, ns, wc, generator=g).to(dev)
    go3 = torch.randn(n, ns, c, generator=g).to(dev)
    go2 = torch.randn(n, c, generator=g).to(dev)
    out = torch.empty(n, ns, c, device=dev)
    _lib.check(L.tgn_grouping_forward(n, ns, c, p(x), p(idx), p(out), S_()))
    assert torch.equal(out, ref_gpu.grouping_forward(x, idx))
    _lib.check(L.tgn_subtraction_forward(n, ns, c, p(x), p(y), p(idx), p(out), S_()))
    assert torch.equal(out, ref_gpu.subtraction_forward(x, y, idx))
    o2 = torch.zeros(n, c, device=dev)
    _lib.check(L.tgn_aggregation_forward(n, ns, c, wc, p(x), p(pos), p(w), p(idx), p(o2), S_()))
    torch.testing.assert_close(o2, ref_gpu.aggregation_forward(x, pos, w, idx), rtol=1e-5, atol=1e-5)
    gi = torch.zeros(n, c, device=dev)
    _lib.check(L.tgn_grouping_backward(n, ns, c, p(go3), p(idx), p(gi), S_()))
    torch.testing.assert_close(gi, ref_gpu.grouping_backward(go3, idx, n), rtol=1e-5, atol=1e-5)
    g1, g2 = torch.zeros(n, c, device=dev), torch.zeros(n, c, device=dev)
    _lib.check(L.tgn_subtraction_backward(n, ns, c, p(idx), p(go3), p(g1), p(g2), S_()))
    r1, r2 = ref_gpu.subtraction_backward(idx, go3)
    torch.testing.assert_close(g1, r1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g2, r2, rtol=1e-5, atol=1e-5)
    a, b_, c_ = torch.zeros_like(x), torch.zeros_like(pos), torch.zeros_like(w)
    _lib.check(L.tgn_aggregation_backward(n, ns, c, wc, p(x), p(pos), p(w), p(idx), p(go2), p(a), p(b_), p(c_), S_()))
    ra, rb, rc = ref_gpu.aggregation_backward(x, pos, w, idx, go2)
    torch.testing.assert_close(a, ra, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b_, rb, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(c_, rc, rtol=1e-4, atol=1e-4)
    k = 3
    ik = idx[:, :k].contiguous()
    wk = torch.rand(n, k, generator=g).to(dev)
    o3 = torch.zeros(n, c, device=dev)
    _lib.check(L.tgn_interpolation_forward(n, c, k, p(x), p(ik), p(wk), p(o3), S_()))
    torch.testing.assert_close(o3, ref_gpu.interpolation_forward(x, ik, wk), rtol=1e-5, atol=1e-5)
    g3 = torch.zeros(n, c, device=dev)
    _lib.check(L.tgn_interpolation_backward(n, c, k, p(go2), p(ik), p(wk), p(g3), S_()))
    torch.testing.assert_close(g3, ref_gpu.interpolation_backward(go2, ik, wk, n), rtol=1e-5, atol=1e-5)


def test_all_ten_verbatim_launchers_on_a_side_stream(dev, oracle):
    """pointops_api.cpp:12-23, all ten names, through `pointops_cuda` (= the ten reference-signature `*_cuda_launcher` symbols of
    include/tgn_pointops.h section 1) on a NON-default torch stream: the inputs are produced on that stream behind a long sleep and
    nothing is synchronised before the call, so a launcher that did not pick the stream up from `tgn_set_default_stream` (or
    forwarded an argument to the wrong slot) reads zeros / writes the wrong buffer.  Checked against the reference's own kernels
    (oracle/_ref) where built, and always against the stream forms (`tgn_*`)."""
    import pointops_cuda as PC
    from oracle import ref_gpu
    from toothgroupnetwork_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    have_ref = ref_gpu.available()
    n, ns, c, wc, k = 700, 9, 24, 6, 3
    g = torch.Generator().manual_seed(29)
    host = dict(x=torch.randn(n, c, generator=g), y=torch.randn(n, c, generator=g), pos=torch.randn(n, ns, c, generator=g),
                w=torch.randn(n, ns, wc, generator=g), go3=torch.randn(n, ns, c, generator=g), go2=torch.randn(n, c, generator=g),
                wk=torch.rand(n, k, generator=g), xyz=torch.from_numpy(synth.uniform_cloud(n, 4)))
    idx_h = torch.randint(0, n, (n, ns), generator=g, dtype=torch.int32)
    real = {k_: v.to(dev) for k_, v in host.items()}
    idx_real = idx_h.to(dev)
    off = torch.tensor([n], dtype=torch.int32, device=dev)
    noff = torch.tensor([64], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    seen = []

    def staged(fn):
        """run fn(inputs) on the side stream with the inputs filled in BEHIND a sleep on that stream"""
        t = {k_: torch.zeros_like(v) for k_, v in real.items()}
        idx = torch.zeros_like(idx_real)
        with torch.cuda.stream(side):
            torch.cuda._sleep(40_000_000)
            for k_ in t:
                t[k_].copy_(real[k_], non_blocking=True)
            idx.copy_(idx_real, non_blocking=True)
            out = fn(t, idx)
        side.synchronize()
        return out

    def z(*shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=dev)

    def via_stream_form(name, *args):
        _lib.check(getattr(L, name)(*args, _lib.stream()), name)
        torch.cuda.synchronize()

    # 1. furthestsampling_cuda ------------------------------------------------------------------------------------------
    def fps(t, idx):
        out, tmp = z(64, dtype=torch.int32), torch.full((n,), 1e10, device=dev)
        PC.furthestsampling_cuda(1, n, t["xyz"], off, noff, tmp, out)
        return out
    got = staged(fps)
    seen.append("furthestsampling_cuda")
    assert np.array_equal(got.cpu().numpy(), oracle.furthestsampling(host["xyz"].numpy(), [n], [64]))
    # 2. knnquery_cuda --------------------------------------------------------------------------------------------------
    def knn(t, idx):
        ki, kd = z(n, 5, dtype=torch.int32), z(n, 5)
        PC.knnquery_cuda(n, 5, t["xyz"], t["xyz"], off, off, ki, kd)
        return ki, kd
    ki, kd = staged(knn)
    seen.append("knnquery_cuda")
    oi, od = oracle.knnquery(5, host["xyz"].numpy(), host["xyz"].numpy(), [n], [n])
    assert np.array_equal(ki.cpu().numpy(), oi) and np.array_equal(torch.sqrt(kd).cpu().numpy(), od)
    if have_ref:
        ri, rd = ref_gpu.knnquery(5, real["xyz"], real["xyz"], off, off)
        assert torch.equal(ki, ri) and torch.equal(kd, rd)
    # 3./4. grouping ----------------------------------------------------------------------------------------------------
    def grp_f(t, idx):
        out = z(n, ns, c)
        PC.grouping_forward_cuda(n, ns, c, t["x"], idx, out)
        return out
    got = staged(grp_f)
    seen.append("grouping_forward_cuda")
    want = z(n, ns, c)
    via_stream_form("tgn_grouping_forward", n, ns, c, p(real["x"]), p(idx_real), p(want))
    assert torch.equal(got, want) and torch.equal(got, real["x"][idx_real.long()])
    if have_ref:
        assert torch.equal(got, ref_gpu.grouping_forward(real["x"], idx_real))

    def grp_b(t, idx):
        gi = z(n, c)
        PC.grouping_backward_cuda(n, ns, c, t["go3"], idx, gi)
        return gi
    got = staged(grp_b)
    seen.append("grouping_backward_cuda")
    want = z(n, c)
    via_stream_form("tgn_grouping_backward", n, ns, c, p(real["go3"]), p(idx_real), p(want))
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    if have_ref:
        torch.testing.assert_close(got, ref_gpu.grouping_backward(real["go3"], idx_real, n), rtol=1e-5, atol=1e-5)
    # 5./6. interpolation -----------------------------------------------------------------------------------------------
    ik_real = idx_real[:, :k].contiguous()

    def itp_f(t, idx):
        out = z(n, c)
        PC.interpolation_forward_cuda(n, c, k, t["x"], idx[:, :k].contiguous(), t["wk"], out)
        return out
    got = staged(itp_f)
    seen.append("interpolation_forward_cuda")
    want = z(n, c)
    via_stream_form("tgn_interpolation_forward", n, c, k, p(real["x"]), p(ik_real), p(real["wk"]), p(want))
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    if have_ref:
        torch.testing.assert_close(got, ref_gpu.interpolation_forward(real["x"], ik_real, real["wk"]), rtol=1e-5, atol=1e-5)

    def itp_b(t, idx):
        gi = z(n, c)
        PC.interpolation_backward_cuda(n, c, k, t["go2"], idx[:, :k].contiguous(), t["wk"], gi)
        return gi
    got = staged(itp_b)
    seen.append("interpolation_backward_cuda")
    want = z(n, c)
    via_stream_form("tgn_interpolation_backward", n, c, k, p(real["go2"]), p(ik_real), p(real["wk"]), p(want))
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    if have_ref:
        torch.testing.assert_close(got, ref_gpu.interpolation_backward(real["go2"], ik_real, real["wk"], n), rtol=1e-5, atol=1e-5)
    # 7./8. subtraction -------------------------------------------------------------------------------------------------
    def sub_f(t, idx):
        out = z(n, ns, c)
        PC.subtraction_forward_cuda(n, ns, c, t["x"], t["y"], idx, out)
        return out
    got = staged(sub_f)
    seen.append("subtraction_forward_cuda")
    assert torch.equal(got, real["x"][:, None, :] - real["y"][idx_real.long()])
    if have_ref:
        assert torch.equal(got, ref_gpu.subtraction_forward(real["x"], real["y"], idx_real))

    def sub_b(t, idx):
        g1, g2 = z(n, c), z(n, c)
        PC.subtraction_backward_cuda(n, ns, c, idx, t["go3"], g1, g2)
        return g1, g2
    g1, g2 = staged(sub_b)
    seen.append("subtraction_backward_cuda")
    w1, w2 = z(n, c), z(n, c)
    via_stream_form("tgn_subtraction_backward", n, ns, c, p(idx_real), p(real["go3"]), p(w1), p(w2))
    torch.testing.assert_close(g1, w1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g2, w2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g1, real["go3"].sum(1), rtol=1e-5, atol=1e-5)
    if have_ref:
        r1, r2 = ref_gpu.subtraction_backward(idx_real, real["go3"])
        torch.testing.assert_close(g1, r1, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(g2, r2, rtol=1e-5, atol=1e-5)
    # 9./10. aggregation ------------------------------------------------------------------------------------------------
    def agg_f(t, idx):
        out = z(n, c)
        PC.aggregation_forward_cuda(n, ns, c, wc, t["x"], t["pos"], t["w"], idx, out)
        return out
    got = staged(agg_f)
    seen.append("aggregation_forward_cuda")
    want = z(n, c)
    via_stream_form("tgn_aggregation_forward", n, ns, c, wc, p(real["x"]), p(real["pos"]), p(real["w"]), p(idx_real), p(want))
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    if have_ref:
        torch.testing.assert_close(got, ref_gpu.aggregation_forward(real["x"], real["pos"], real["w"], idx_real), rtol=1e-5, atol=1e-5)

    def agg_b(t, idx):
        a, b_, c_ = z(n, c), z(n, ns, c), z(n, ns, wc)
        PC.aggregation_backward_cuda(n, ns, c, wc, t["x"], t["pos"], t["w"], idx, t["go2"], a, b_, c_)
        return a, b_, c_
    a, b_, c_ = staged(agg_b)
    seen.append("aggregation_backward_cuda")
    wa, wb, wcg = z(n, c), z(n, ns, c), z(n, ns, wc)
    via_stream_form("tgn_aggregation_backward", n, ns, c, wc, p(real["x"]), p(real["pos"]), p(real["w"]), p(idx_real), p(real["go2"]),
                    p(wa), p(wb), p(wcg))
    torch.testing.assert_close(a, wa, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b_, wb, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(c_, wcg, rtol=1e-4, atol=1e-4)
    if have_ref:
        ra, rb, rc = ref_gpu.aggregation_backward(real["x"], real["pos"], real["w"], idx_real, real["go2"])
        torch.testing.assert_close(a, ra, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(b_, rb, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(c_, rc, rtol=1e-4, atol=1e-4)
    # every name of pointops_api.cpp:13-22 went through the shim
    assert sorted(seen) == sorted(n_ for n_ in dir(PC) if n_.endswith("ward_cuda") or n_ in ("furthestsampling_cuda", "knnquery_cuda"))
    assert len(seen) == 10


def test_pointops_cuda_shim_runs_reference_style_code(dev, oracle):
    """The legacy native-module API (pointops_api.cpp:13-22): caller-allocated, pre-initialised buffers."""
    import pointops_cuda
    xyz_np = synth.uniform_cloud(900, 2)
    xyz = T(xyz_np, dev)
    off = torch.tensor([900], dtype=torch.int32, device=dev)
    noff = torch.tensor([100], dtype=torch.int32, device=dev)
    idx = torch.zeros(100, dtype=torch.int32, device=dev)
    tmp = torch.full((900,), 1e10, device=dev)
    pointops_cuda.furthestsampling_cuda(1, 900, xyz, off, noff, tmp, idx)
    assert np.array_equal(idx.cpu().numpy(), oracle.furthestsampling(xyz_np, [900], [100]))
    kidx = torch.zeros(900, 5, dtype=torch.int32, device=dev)
    kd2 = torch.zeros(900, 5, device=dev)
    pointops_cuda.knnquery_cuda(900, 5, xyz, xyz, off, off, kidx, kd2)
    oi, od = oracle.knnquery(5, xyz_np, xyz_np, [900], [900])
    assert np.array_equal(kidx.cpu().numpy(), oi) and np.array_equal(torch.sqrt(kd2).cpu().numpy(), od)
    out = torch.empty(900, 5, 3, device=dev)
    pointops_cuda.grouping_forward_cuda(900, 5, 3, xyz, kidx, out)
    assert np.array_equal(out.cpu().numpy(), xyz_np[oi])


@pytest.mark.parametrize("variant", [0, 3, 5])
@pytest.mark.parametrize("shape", [(600, 12, 32, 4), (257, 7, 64, 8), (300, 5, 16, 16), (200, 9, 128, 16), (150, 6, 10, 5), (90, 4, 36, 12)])
def test_gather_family_every_kernel_variant_matches_a_float64_evaluation(dev, variant, shape):
    """The gather family's kernel variants (tgn_set_tuning "gather_v4": 0 = dword lanes and atomics everywhere, 3 = 16-byte lanes
    forward and backward, 5 = the default: 16-byte lanes forward, owner-side sums backward) at channel counts that take the fast
    paths (c % 4 == 0, powers of two) and that do not, against the operators' definitions (grouping_cuda_kernel.cu:5-25,
    subtraction_cuda_kernel.cu:5-30, aggregation_cuda_kernel.cu:5-39, interpolation_cuda_kernel.cu:5-33) evaluated in float64.
    Outputs the reference accumulates into are handed over pre-filled: what was there must still be added to."""
    from toothgroupnetwork_amd import _lib
    L, p = _lib.lib(), _lib.ptr
    n, ns, c, wc = shape
    g = torch.Generator().manual_seed(1000 + n)
    R = lambda *sh: torch.randn(*sh, generator=g).to(dev)   # noqa: E731
    x, y, pos, w, go3, go2 = R(n, c), R(n, c), R(n, ns, c), R(n, ns, wc), R(n, ns, c), R(n, c)
    idx = torch.randint(0, n, (n, ns), generator=g, dtype=torch.int32).to(dev)
    il = idx.long()
    k = 3
    ik, wk = idx[:, :k].contiguous(), torch.rand(n, k, generator=g).to(dev)
    d = lambda t: t.double()   # noqa: E731
    wfull = d(w)[:, :, torch.arange(c, device=dev) % wc]                       # weight[n, j, ci % w_c]
    base2, base3w = R(n, c), R(n, ns, wc)
    close = lambda a, b, tol=2e-5: torch.testing.assert_close(d(a), b, rtol=tol, atol=tol)   # noqa: E731
    with _lib.tuning(gather_v4=variant):
        st = _lib.stream
        out = torch.empty(n, ns, c, device=dev)
        _lib.check(L.tgn_grouping_forward(n, ns, c, p(x), p(idx), p(out), st()))
        assert torch.equal(out, x[il])
        _lib.check(L.tgn_subtraction_forward(n, ns, c, p(x), p(y), p(idx), p(out), st()))
        assert torch.equal(out, x[:, None, :] - y[il])
        o = base2.clone()
        _lib.check(L.tgn_aggregation_forward(n, ns, c, wc, p(x), p(pos), p(w), p(idx), p(o), st()))
        close(o, d(base2) + ((d(x)[il] + d(pos)) * wfull).sum(1))
        o = base2.clone()
        _lib.check(L.tgn_interpolation_forward(n, c, k, p(x), p(ik), p(wk), p(o), st()))
        close(o, d(base2) + (d(x)[ik.long()] * d(wk)[:, :, None]).sum(1))
        # backward: scatters accumulate into what the caller hands over
        gi = base2.clone()
        _lib.check(L.tgn_grouping_backward(n, ns, c, p(go3), p(idx), p(gi), st()))
        close(gi, d(base2).index_put((il.reshape(-1),), d(go3).reshape(-1, c), accumulate=True))
        gi = base2.clone()
        _lib.check(L.tgn_interpolation_backward(n, c, k, p(go2), p(ik), p(wk), p(gi), st()))
        close(gi, d(base2).index_put((ik.long().reshape(-1),), (d(go2)[:, None, :] * d(wk)[:, :, None]).reshape(-1, c), accumulate=True))
        g1, g2 = base2.clone(), torch.zeros(n, c, device=dev)
        _lib.check(L.tgn_subtraction_backward(n, ns, c, p(idx), p(go3), p(g1), p(g2), st()))
        close(g1, d(base2) + d(go3).sum(1))
        close(g2, torch.zeros(n, c, device=dev, dtype=torch.float64).index_put((il.reshape(-1),), -d(go3).reshape(-1, c), accumulate=True))
        ga, gp, gw = torch.zeros(n, c, device=dev), torch.full((n, ns, c), 7.0, device=dev), base3w.clone()
        _lib.check(L.tgn_aggregation_backward(n, ns, c, wc, p(x), p(pos), p(w), p(idx), p(go2), p(ga), p(gp), p(gw), st()))
        gwfull = d(go2)[:, None, :] * wfull
        close(ga, torch.zeros(n, c, device=dev, dtype=torch.float64).index_put((il.reshape(-1),), gwfull.reshape(-1, c), accumulate=True), 1e-4)
        close(gp, gwfull)                                                        # assigned, not accumulated (aggregation_cuda_kernel.cu:35)
        t = d(go2)[:, None, :] * (d(x)[il] + d(pos))                             # (n, ns, c) -> summed over the channels that share a weight
        want_w = torch.zeros(n, ns, wc, device=dev, dtype=torch.float64).index_add_(2, torch.arange(c, device=dev) % wc, t)
        close(gw, d(base3w) + want_w, 1e-4)
    torch.cuda.synchronize()


@pytest.mark.parametrize("C", [256, 64, 30, 7])
def test_three_interpolate_with_fused_epilogue_equals_the_separate_passes(dev, C):
    """tgn_three_interpolate_ex: [relu](interpolation (+ add)) in one kernel -- bit for bit what three_interpolate followed by
    torch's add and relu give (same operation order), with 16-byte lanes (C % 4 == 0) and with dword lanes; `add` is consumed
    in place."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    B, N, S = 3, 1500, 200
    g = torch.Generator().manual_seed(C)
    xyz1 = T(np.stack([synth.arch_cloud(N, 10 + b, False) for b in range(B)]), dev)
    xyz2 = xyz1[:, torch.randperm(N, generator=g)[:S].to(dev)].contiguous()
    f2 = torch.randn(B, S, C, generator=g).to(dev)
    add = torch.randn(B, N, C, generator=g).to(dev)
    dist, idx = U.three_nn(xyz1, xyz2)
    plain = U.three_interpolate(f2, dist, idx)
    assert torch.equal(U.three_interpolate_add_relu(f2, dist, idx), plain)
    assert torch.equal(U.three_interpolate_add_relu(f2, dist, idx, relu=True), torch.relu(plain))
    buf = add.clone()
    got = U.three_interpolate_add_relu(f2, dist, idx, add=buf, relu=True)
    assert got.data_ptr() == buf.data_ptr() and torch.equal(got, torch.relu(plain + add))
    assert torch.equal(U.three_interpolate_add_relu(f2, dist, idx.int(), add=add.clone()), plain + add)


def test_linear_relu_epilogue_is_bit_identical_to_the_two_calls(dev):
    """pointnet2_utils.linear_relu (feature-propagation tails and heads in eval mode): the GEMM with the ReLU in its epilogue equals
    F.linear + relu_ bit for bit; non-contiguous rows and inputs that need grad take the two calls."""
    import torch.nn.functional as F

    from toothgroupnetwork_amd import pointnet2_utils as U
    g = torch.Generator().manual_seed(3)
    for (B, N, K, C) in ((2, 3000, 256, 128), (1, 24000, 128, 32), (3, 77, 35, 17)):
        x, W, b = torch.randn(B, N, K, generator=g).to(dev), torch.randn(C, K, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
        want = torch.relu(F.linear(x, W, b))
        with torch.no_grad():
            assert torch.equal(U.linear_relu(x, W, b), want)
            xt = x.permute(0, 2, 1).contiguous().permute(0, 2, 1)            # same values, rows not contiguous: the two calls
            assert torch.equal(U.linear_relu(xt, W, b), torch.relu(F.linear(xt, W, b)))   # (torch picks another GEMM for this layout)
            torch.testing.assert_close(U.linear_relu(xt, W, b), want, rtol=1e-4, atol=1e-4)
        xg = x.clone().requires_grad_()
        y = U.linear_relu(xg, W, b)
        y.sum().backward()
        assert torch.equal(y.detach(), want) and xg.grad is not None
