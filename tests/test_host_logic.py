"""CPU: host-side logic -- sharding arithmetic, drop-in import layout, module/state_dict compatibility."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO
from toothgroupnetwork_amd import sharding


@pytest.mark.parametrize("n,world", [(0, 1), (1, 8), (7, 8), (8, 8), (9, 8), (1800, 8), (1801, 4), (5, 2)])
def test_shard_partition(n, world):
    for mode in ("contiguous", "round_robin"):
        parts = [sharding.shard_indices(n, r, world, mode) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 4, 4)
    with pytest.raises(ValueError):
        sharding.shard_indices(4, 0, 2, "nope")


def test_gather_metrics_single_process():
    m = sharding.gather_metrics([1.0, 2.0, 3.0])
    assert m.shape == (1, 3) and m.dtype == torch.float64
    res = sharding.run_sharded(list(range(10)), lambda i: {"sum": i, "sq": i * i}, 0, 1)
    assert res["count"] == 10 and res["sum"] == 45 and res["sq"] == 285


def test_dropin_modules_share_parameter_names_with_reference_state_dict():
    """state_dicts written by the reference's modules load into ours (same names and shapes)."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    sd = torch.load(os.path.join(GOLDEN, "module_weights.pt"))
    sa = U.PointNetSetAbstractionMsg(128, [0.1, 0.2], [8, 16], 6, [[16, 24], [16, 32]])
    ssg = U.PointNetSetAbstraction(64, 0.2, 16, 6 + 3, [16, 32], False)
    fp = U.PointNetFeaturePropagation(56 + 6, [32, 16])
    sa.load_state_dict(sd["sa"], strict=True)
    ssg.load_state_dict(sd["ssg"], strict=True)
    fp.load_state_dict(sd["fp"], strict=True)


def test_public_api_surface():
    from external_libs.pointnet2_utils import pointnet2_utils as U
    from external_libs.pointops.functions import pointops as P
    import pointops_cuda
    for n in ["furthestsampling", "knnquery", "grouping", "queryandgroup", "subtraction", "aggregation",
              "interpolation", "interpolation2", "FurthestSampling", "KNNQuery", "Grouping", "Subtraction",
              "Aggregation", "Interpolation"]:
        assert hasattr(P, n), n
    for n in ["timeit", "pc_normalize", "square_distance", "index_points", "farthest_point_sample",
              "farthest_point_sample_np", "query_ball_point", "sample_and_group", "sample_and_group_all",
              "PointNetSetAbstraction", "PointNetSetAbstractionMsg", "PointNetFeaturePropagation"]:
        assert hasattr(U, n), n
    for n in ["knnquery_cuda", "furthestsampling_cuda", "grouping_forward_cuda", "grouping_backward_cuda",
              "interpolation_forward_cuda", "interpolation_backward_cuda", "subtraction_forward_cuda",
              "subtraction_backward_cuda", "aggregation_forward_cuda", "aggregation_backward_cuda"]:
        assert hasattr(pointops_cuda, n), n  # pointops_api.cpp:13-22


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference checkout not present")
def test_reference_models_import_our_operators_unchanged():
    """sys.path = [repo, reference]: the reference's model files resolve external_libs.* to THIS repo's ops."""
    code = (
        "import sys; sys.dont_write_bytecode=True; sys.path[:0]=[%r, '/root/reference']\n"
        "import models.modules.pointnet_pp as m\n"
        "import toothgroupnetwork_amd.pointnet2_utils as U\n"
        "assert m.PointNetSetAbstractionMsg is U.PointNetSetAbstractionMsg\n"
        "net = m.get_model()\n"
        "assert type(net.sa1) is U.PointNetSetAbstractionMsg and type(net.fp1) is U.PointNetFeaturePropagation\n"
        "import external_libs.scheduler.scheduler_factory as sf\n"
        "assert sf.__file__.startswith('/root/reference')\n"
        "from external_libs.pointops.functions import pointops\n"
        "import toothgroupnetwork_amd.pointops as P\n"
        "assert pointops.queryandgroup is P.queryandgroup\n"
        # the tsegnet consumers north_star names (models/tsegnet_model.py -> modules/tsegnet.py -> tsg_centroid_module / tsg_seg_module)
        # and the loss / utility modules that import square_distance etc. (tgn_loss.py:4, tsg_loss.py:2, ops_utils.py:5)
        # (open3d / trimesh / wandb -- mesh I/O and logging, absent from this image and outside the path -- are stubbed)
        "import types\n"
        "for n in ('open3d', 'trimesh', 'wandb'): sys.modules[n] = types.ModuleType(n)\n"
        "import models.modules.tsg_centroid_module as cm, models.modules.tsg_seg_module as sm, models.modules.tsegnet as tn\n"
        "import models.tgn_loss as tl, models.tsg_loss as sl, ops_utils as ou\n"
        "assert tl.square_distance is U.square_distance and sl.square_distance is U.square_distance\n"
        "assert ou.square_distance is U.square_distance\n"
        "c = cm.get_model(); s = sm.get_model()\n"
        "assert type(c.sa1) is U.PointNetSetAbstractionMsg and type(c.fp1) is U.PointNetFeaturePropagation\n"
        "assert type(s.flatten_sa) is U.PointNetSetAbstraction and s.flatten_sa.group_all is True\n"
        "assert [m.out_channels for m in s.flatten_sa.mlp_convs] == [256, 512] and s.flatten_sa.mlp_convs[0].in_channels == 515\n"
        "assert tn.get_centroid_module is cm.get_model and tn.get_seg_module is sm.get_model and tn.square_distance is U.square_distance\n"
        # the three model wrappers north_star names import unchanged, and the preprocess resampler calls this repo's FPS
        "import models.pointnet_pp_model, models.transformer_model, models.tsegnet_model as tm, models.fps_grouping_network_model\n"
        "assert tm.square_distance is U.square_distance\n"
        "import gen_utils as gu\n"
        "assert gu.pointops is pointops and gu.pointops.furthestsampling is P.furthestsampling\n"
        "print('ok')\n" % REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp",
                         env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference checkout not present")
def test_reference_point_transformer_builds_on_our_operators_and_mirrors_interchange_weights():
    """BASELINE.json config 4: the reference's PointTransformerSeg (cbl_point_transformer_module.py:219-235, tgnet_fps stage
    sizes) is constructed with sys.path = [repo, reference] -- its blocks / heads / basic_operators import THIS repo's
    pointops -- and the host-side mirrors of its building blocks (toothgroupnetwork_amd.point_transformer) carry the same
    parameter names and shapes as the reference classes, so trained weights move between them."""
    code = (
        "import sys; sys.dont_write_bytecode=True; sys.path[:0]=[%r, '/root/reference']\n"
        "import models.modules.cbl_point_transformer.cbl_point_transformer_module as M\n"
        "import models.modules.cbl_point_transformer.blocks as RB\n"
        "import toothgroupnetwork_amd.pointops as P\n"
        "from toothgroupnetwork_amd import point_transformer as PT\n"
        "assert RB.pointops.queryandgroup is P.queryandgroup and RB.pointops.furthestsampling is P.furthestsampling\n"
        "net = M.get_model(c=6, k=17, planes=[32,64,128,256,512], stride=[1,4,4,4,4], nsample=[36,24,24,24,24], "
        "blocks=[2,3,4,6,3], block_num=5)\n"
        "assert type(net.enc1[1].transformer2).__name__ == 'PointTransformerLayer'\n"
        "for name, args in (('PointTransformerLayer', (32, 32, 8, 36)), ('TransitionDown', (32, 64, 4, 24)), "
        "('TransitionDown', (6, 32, 1, 36)), ('TransitionUp', (512, None)), ('TransitionUp', (256, 128)), "
        "('PointTransformerBlock', (64, 64, 8, 24))):\n"
        "    a = getattr(RB, name)(*args).state_dict(); b = getattr(PT, name)(*args)\n"
        "    assert list(a.keys()) == list(b.state_dict().keys()), name\n"
        "    b.load_state_dict(a)\n"
        "u = PT.PointTransformerUNet()\n"
        "ref = {k for k in net.state_dict() if k.startswith(('enc', 'dec'))}\n"
        "assert len(ref) == len(u.state_dict()), (len(ref), len(u.state_dict()))\n"
        "print('ok')\n" % REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp",
                         env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_synthetic_scans_have_the_documented_density():
    from toothgroupnetwork_amd import synth
    pts = synth.arch_cloud(24000, seed=0)
    assert pts.shape == (24000, 6) and pts.dtype == np.float32
    assert np.abs(pts[:, :3]).max() <= 1.2
    np.testing.assert_allclose(np.linalg.norm(pts[:, 3:], axis=1), 1.0, atol=1e-4)
    d = ((pts[:200, None, :3] - pts[None, :, :3]) ** 2).sum(-1)
    per_ball = (d <= 0.05 ** 2).sum(1).mean()
    assert 20 <= per_ball <= 90, per_ball  # r=0.05 balls hold a few dozen points


def test_algorithmic_bytes_match_the_survey():
    """compulsory HBM bytes per scan of the two benchmark shapes"""
    from toothgroupnetwork_amd import hotpath
    assert hotpath.algorithmic_bytes(**hotpath.SHAPE_A)[0] == 46109952
    assert hotpath.algorithmic_bytes(**hotpath.SHAPE_B)[0] == 165863680
    assert hotpath.algorithmic_bytes(**hotpath.SHAPE_A, fused=True)[0] == 12588288


def test_known_offsets_spare_the_device_to_host_copy():
    """pointops.register_offsets / offsets_host: host values a caller already knows are used instead of a device->host copy
    (what makes a dense-batch Point-Transformer forward free of host round trips); an in-place write drops the entry."""
    import torch

    from toothgroupnetwork_amd import pointops as P

    o = torch.tensor([5, 9], dtype=torch.int32)
    q = torch.tensor([2, 4], dtype=torch.int32)
    assert P.offsets_host(o, q) == [[5, 9], [2, 4]]          # unknown: read from the tensors (one copy for both)
    P.register_offsets(o, [5, 9])
    calls = []
    orig = P._offsets_host
    P._offsets_host = lambda t: calls.append(t.numel()) or orig(t)
    try:
        assert P.offsets_host(o) == [[5, 9]] and calls == []   # known: no copy
        assert P.offsets_host(o, q) == [[5, 9], [2, 4]] and calls == [2]   # only the unknown one is fetched
        o.add_(1)                                             # written in place: the registered values are stale
        assert P.offsets_host(o) == [[6, 10]] and calls == [2, 2]
    finally:
        P._offsets_host = orig
    with pytest.raises(AssertionError):
        P.register_offsets(q, [1, 2, 3])
    with torch.inference_mode():                              # no version counter there: still usable
        r = torch.tensor([7], dtype=torch.int32)
        P.register_offsets(r, [7])
        assert P.offsets_host(r) == [[7]]


def test_the_phased_plan_follows_its_measurements():
    """hotpath.plan_schedule: the last level's query moves in front of the groupings only where those leave room beside
    the FPS level-1 launch -- the headline shape at a full batch (measured: FPS level 1 3.33 ms, groupings 2.95 ms;
    4.69 -> 4.62 ms per step), not the multi-scale shape whose groupings take ten times the FPS launch (measured: 2 %
    slower); the spacer follows the FPS set-up time."""
    from toothgroupnetwork_amd import hotpath as H

    a = H.plan_schedule(3.33, 2.95, 0.17, 3)
    assert a["last_query_early"] and 90 <= a["spacer_us"] <= 110
    assert not H.plan_schedule(0.9, 9.0, 0.17, 3)["last_query_early"]          # Shape B
    assert not H.plan_schedule(3.33, 3.25, 0.17, 3)["last_query_early"]        # no room left
    assert H.plan_schedule(3.33, 3.05, 0.15, 3)["last_query_early"]            # what the box measures for Shape A
    assert not H.plan_schedule(3.33, 0.1, 0.17, 1)["last_query_early"]         # a single level has nothing to move
    assert H.plan_schedule(1.0, 0.5, 0.0, 2)["spacer_us"] == 0 and H.plan_schedule(9.0, 0.5, 2.0, 2)["spacer_us"] == 600
    assert H.plan_schedule(3.33, 2.95, 0.15, 3, grid_ms=0.16)["spacer_us"] == 250              # the grid build between phase 2 and FPS level 1


def test_cpulist_parsing_and_numa_pinning_are_best_effort():
    """sharding.parse_cpulist reads sysfs cpulists; pin_to_gpu_numa gives up quietly where there is no GPU / no sysfs entry."""
    from toothgroupnetwork_amd import sharding as S

    assert S.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert S.parse_cpulist("") == [] and S.parse_cpulist("5") == [5]
    before = sorted(os.sched_getaffinity(0))
    assert S.pin_to_gpu_numa(0, sysfs="/nonexistent") is None
    assert sorted(os.sched_getaffinity(0)) == before


def test_fused_plan_shapes_and_algorithmic_bytes():
    """hotpath: per-branch shared-MLP widths of a level (one list for all branches or one per branch) and the fused byte model"""
    from toothgroupnetwork_amd import hotpath as H

    assert H._branch_mlps([64, 128], 2) == [[64, 128], [64, 128]]
    assert H._branch_mlps([[8, 16], [24, 32]], 2) == [[8, 16], [24, 32]]
    total, per = H.algorithmic_bytes(**H.SHAPE_B, fused=True)
    # level 3, both branches: indices + inputs + centres + the (S, 1024) output each
    assert per[2]["group"] == sum(4 * 256 * k + 4 * 512 * 1027 + 12 * 256 + 4 * 256 * 1024 for k in (32, 64))
    assert total == 15687936


def test_derived_operand_memo_follows_the_parameters():
    """_derived.cached: the folded operands of the fused eval paths are rebuilt exactly when a source tensor is written in place
    (optimiser step, load_state_dict, BatchNorm statistics), replaced, or the shape parameters of the derivation change."""
    import torch
    import torch.nn as nn
    from toothgroupnetwork_amd import _derived, point_transformer as PT
    lin, bn = nn.Linear(8, 5, bias=False), nn.BatchNorm1d(5).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_()
        bn.bias.normal_()
        x = torch.randn(11, 8)
        W, b = PT.folded_linear(lin, bn)
        assert torch.allclose(torch.nn.functional.linear(x, W, b), bn(lin(x)), atol=1e-6)
        assert PT.folded_linear(lin, bn)[0] is W                               # hit
        bn.running_mean.add_(1.0)                                              # in-place write: version counter
        W2, b2 = PT.folded_linear(lin, bn)
        assert W2 is not W and torch.allclose(torch.nn.functional.linear(x, W2, b2), bn(lin(x)), atol=1e-6)
        lin.load_state_dict({"weight": torch.randn(5, 8)})                     # copy_ into the parameter
        W3, b3 = PT.folded_linear(lin, bn)
        assert W3 is not W2 and torch.allclose(torch.nn.functional.linear(x, W3, b3), bn(lin(x)), atol=1e-6)
        lin.weight = nn.Parameter(lin.weight.detach().clone() * 2)             # replaced: identity
        W4, b4 = PT.folded_linear(lin, bn)
        assert W4 is not W3 and torch.allclose(torch.nn.functional.linear(x, W4, b4), bn(lin(x)), atol=1e-5)
    calls = []
    build = lambda: calls.append(1) or len(calls)
    assert _derived.cached(bn, "t", [bn.weight], 3, build) == 1 and _derived.cached(bn, "t", [bn.weight], 3, build) == 1
    assert _derived.cached(bn, "t", [bn.weight], 4, build) == 2                # other shape parameters
    with torch.inference_mode():
        t = torch.ones(3)                                                      # no version counter: not memoised
        assert _derived.cached(bn, "u", [t], None, build) == 3 and _derived.cached(bn, "u", [t], None, build) == 4
    seq = nn.Sequential(nn.Linear(8, 8), nn.BatchNorm1d(8), nn.ReLU(inplace=True), nn.Linear(8, 4)).eval()
    with torch.no_grad():
        assert torch.allclose(PT.mlp_eval(seq, x.clone()), seq(x.clone()), atol=1e-6)


def test_derived_operand_memo_follows_storage_swaps_and_copies():
    """toothgroupnetwork_amd/_derived.py: a memoised operand must be rebuilt when a source parameter is written in place (version),
    when its storage is swapped under the same Parameter object (`module.double()`, `.to(device)`, `p.data = ...`: same object, same
    version -- the storage address / device / dtype part of the key), and must not travel with a deep copy of the module."""
    import copy

    import torch

    from toothgroupnetwork_amd import _derived
    lin = torch.nn.Linear(4, 3)
    builds = []

    def build():
        builds.append(1)
        return lin.weight.detach().clone() * 2

    def get(m=lin):
        return _derived.cached(m, "w2", _derived.sources(m), None, build)
    a = get()
    assert get() is a and len(builds) == 1                                   # memo hit
    with torch.no_grad():
        lin.weight.mul_(3.0)                                                 # in-place write: version counter
    b = get()
    assert len(builds) == 2 and torch.equal(b, lin.weight.detach() * 2)
    w, v = lin.weight, lin.weight._version
    lin.double()                                                             # same Parameter object, same version, new storage
    assert lin.weight is w and lin.weight._version == v
    c = get()
    assert len(builds) == 3 and c.dtype == torch.float64
    lin.weight.data = torch.ones(3, 4, dtype=torch.float64)                  # storage swapped by hand
    d = get()
    assert len(builds) == 4 and torch.equal(d, torch.full((3, 4), 2.0, dtype=torch.float64))
    twin = copy.deepcopy(lin)
    assert not twin.__dict__.get("_tgn_derived")                             # the copy starts with an empty memo
    _derived.invalidate(lin)
    get()
    assert len(builds) == 5


def test_effective_cpus_honours_the_cgroup_quota(tmp_path):
    """sharding.effective_cpus: the affinity mask cut down by the container's CPU quota (cgroup v2 cpu.max / v1 cfs files)."""
    import os

    from toothgroupnetwork_amd import sharding
    have = len(os.sched_getaffinity(0))
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert sharding.effective_cpus(str(tmp_path)) == have
    (tmp_path / "cpu.max").write_text("150000 100000\n")                       # 1.5 cores' worth of time
    assert sharding.effective_cpus(str(tmp_path)) == min(have, 2)
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert sharding.effective_cpus(str(tmp_path)) == min(have, 16)
    (tmp_path / "cpu.max").unlink()
    os.makedirs(tmp_path / "cpu")
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("300000\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert sharding.effective_cpus(str(tmp_path)) == min(have, 3)
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert sharding.effective_cpus(str(tmp_path)) == have


def test_cpus_for_this_rank_divides_mask_and_quota_once_each(tmp_path, monkeypatch):
    """sharding.cpus_for_this_rank: the quota is shared by all ranks of the node, the NUMA-narrowed affinity mask only by the ranks
    on that node (round 4 divided the narrowed mask by all ranks: 2 x 64 cores and 8 ranks gave 8 loader threads, not 16)."""
    from toothgroupnetwork_amd import sharding
    (tmp_path / "cpu.max").write_text("max 100000\n")
    monkeypatch.setattr(sharding, "_affinity_count", lambda: 64)
    monkeypatch.setattr(sharding, "_PIN", {"before": 128, "after": 64})
    assert sharding.cpus_for_this_rank(8, str(tmp_path)) == 16          # 4 of the 8 ranks share this node's 64 cores
    assert sharding.cpus_for_this_rank(1, str(tmp_path)) == 64
    monkeypatch.setattr(sharding, "_affinity_count", lambda: 32)
    monkeypatch.setattr(sharding, "_PIN", {"before": 128, "after": 32})
    assert sharding.cpus_for_this_rank(8, str(tmp_path)) == 16          # NPS4: 2 ranks per 32-core node
    monkeypatch.setattr(sharding, "_PIN", {})
    monkeypatch.setattr(sharding, "_affinity_count", lambda: 128)
    assert sharding.cpus_for_this_rank(8, str(tmp_path)) == 16          # not pinned: the whole mask over all ranks
    (tmp_path / "cpu.max").write_text("1600000 100000\n")               # a 16-core quota for the whole node
    assert sharding.cpus_for_this_rank(8, str(tmp_path)) == 2
    assert sharding.cpus_for_this_rank(1, str(tmp_path)) == 16


def test_config_object_is_the_one_place_for_python_side_switches():
    """toothgroupnetwork_amd.config: one object, seeded from the environment once; override() for a block; the module-level names of
    rounds 1-4 (U.FUSED_SA, P.KNN_GRID, _lib.INDEX_CHECK, ...) are live aliases of its fields, reads AND writes."""
    from toothgroupnetwork_amd import _lib, config, pointnet2_utils as U, pointops as P
    c = config.cfg
    assert (U.FUSED_SA, U.COMMUTE_FP, U.SA_BF16X3) == (c.fused_sa, c.commute_fp, c.sa_bf16x3)
    assert (P.KNN_GRID, P.KNN_GRID_MIN_POINTS, P._KNN_CACHE_SIZE, _lib.INDEX_CHECK) == (c.knn_grid, c.knn_grid_min_points, c.knn_cache_size, c.index_check)
    with config.override(fused_sa=False, knn_cache_size=0, index_check="off"):
        assert U.FUSED_SA is False and P._KNN_CACHE_SIZE == 0 and _lib.INDEX_CHECK == "off" and not _lib._checking()
    assert U.FUSED_SA is c.fused_sa and _lib.INDEX_CHECK == c.index_check
    keep = c.commute_fp
    U.COMMUTE_FP = not keep                      # a legacy write lands in the config object
    assert c.commute_fp is (not keep)
    U.COMMUTE_FP = keep
    with pytest.raises(AttributeError):
        config.override(no_such_switch=1)
    fresh = config.Config()
    assert fresh.fused_sa and fresh.sa_bf16x3 and fresh.knn_grid_min_points == 3000 and fresh.index_check == "sync"


def test_tuning_table_round_trip_and_unknown_key():
    """tgn_set_tuning / tgn_get_tuning (include/tgn_pointops.h): host-side table, no GPU needed.  Known keys round-trip, (nt, p) pairs
    pack as nt * 256 + p, the context manager restores, an unknown key is an error (not silently a no-op)."""
    from toothgroupnetwork_amd import _lib
    L = _lib.lib()
    for key, default in (("fps_plain", 0), ("fps_bucket_min", -1), ("fps_cell_bits", 4), ("ball_bitmap", 2), ("sa_tile", 0), ("knn_grid_scale", 1000), ("gather_v4", 5)):
        assert L.tgn_get_tuning(key.encode(), -12345) == default, key
    with _lib.tuning(fps_bucket_config=(512, 48), fps_plain=1):
        assert L.tgn_get_tuning(b"fps_bucket_config", 0) == 512 * 256 + 48
        assert L.tgn_get_tuning(b"fps_plain", 0) == 1
    assert L.tgn_get_tuning(b"fps_bucket_config", -1) == 0 and L.tgn_get_tuning(b"fps_plain", -1) == 0
    assert L.tgn_set_tuning(b"no_such_switch", 1) != 0 and b"no_such_switch" in L.tgn_last_error()
    assert L.tgn_get_tuning(b"no_such_switch", 77) == 77
    with pytest.raises(RuntimeError):
        _lib.set_tuning("no_such_switch", 1)


def test_secondary_bench_loads_and_prices_the_fused_levels():
    """bench.py attaches tools/secondary_bench.py's measurements to its JSON line; on CPU: the module loads the way bench.py loads it
    and its flop count for the fused levels is the documented one (20.3 GFLOP per Shape-A scan, 57.7 per Shape-B scan)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sec = bench._secondary()
    from toothgroupnetwork_amd import hotpath
    assert sec.fused_flops(hotpath.SHAPE_A) == 20293091328
    assert sec.fused_flops(hotpath.SHAPE_B) == 57713197056
    r = sec._roof("hbm", 4000.0, 8000.0, "GB/s")
    assert r["frac"] == 0.5 and sec._roof("latency", 0.81, None, "us")["frac"] is None
