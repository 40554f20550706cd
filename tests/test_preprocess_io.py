"""Preprocess I/O either side of the FPS and the sharded runner (BASELINE.json config 5).

CPU part: the native OBJ reader and vertex normals (libtgn_pointops.so section 4 runs on the host) against what the
REFERENCE's gen_utils.read_txt_obj_ls parsed from the same synthetic files (tests/golden/make_golden_r2_io.py) and the
oracle; the runner's control flow on 2 gloo ranks with the oracle's FPS plugged in.
GPU part (-m gpu): the whole pipeline -- native reader, normals, label remap, scaling, batched GPU FPS, np.save -- must
write byte-identical .npy files to the ones the reference's preprocess_data.py wrote (sha256 in the golden file)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
from make_golden_r2_io import SCANS, SMALL, write_dataset  # noqa: E402  (pure python, no reference import at module level)


def test_native_obj_reader_matches_the_reference_parser(tmp_path, golden_r2):
    from oracle import meshio as OM
    from toothgroupnetwork_amd import preprocess, synth
    for style, nu, nv, seed in SMALL:
        path = tmp_path / f"{style}.obj"
        path.write_text(synth.obj_text(nu, nv, seed, style))
        v, f = preprocess.read_obj(str(path))
        assert v.dtype == np.float64 and f.dtype == np.int64
        assert np.array_equal(v, golden_r2[f"obj_{style}_vertices"])            # what the reference handed to open3d
        assert np.array_equal(f - 1, golden_r2[f"obj_{style}_triangles"])
        ov, of = OM.read_obj(str(path))
        assert np.array_equal(v, ov) and np.array_equal(f, of)
        vn = preprocess.read_txt_obj_ls(str(path))[0]
        assert np.array_equal(vn[:, :3], v)
        assert np.array_equal(vn, golden_r2[f"obj_{style}_vn"])                 # normals: oracle restatement (parity unpinned)
        assert np.array_equal(vn[:, 3:], OM.vertex_normals(v, f - 1))


def test_obj_reader_quirks_and_errors(tmp_path):
    from toothgroupnetwork_amd import preprocess
    p = tmp_path / "q.obj"
    p.write_text("# c\nv 1 2 3 0.5 0.5 0.5\nvn 0 0 1\n  v\t-1e-3  +2.5 .5\r\nf 1//7 2//8 1//9\nf 2 1 2 9\n   \nv 9 9 9\n")
    v, f = preprocess.read_obj(str(p))
    assert np.array_equal(v, [[1, 2, 3], [-1e-3, 2.5, 0.5]])                    # extra tokens ignored, blank line ends the file
    assert np.array_equal(f, [[1, 2, 1], [2, 1, 2]])
    for bad in ("v 1 2\n", "v 1 2 x\n", "f 1/2/3 4/5/6 7/8/9\n", "f 1 2\n"):     # what float() / int() / the array build reject
        p.write_text(bad)
        with pytest.raises(ValueError):
            preprocess.read_obj(str(p))
    with pytest.raises(ValueError):
        preprocess.read_obj(str(tmp_path / "missing.obj"))
    # float() spellings: both parsers behind parse_double (from_chars for plain decimals, strtod for the rest)
    toks = ["+1.5", "-.5e+2", "5.", "1e400", "-1e400", "1e-400", "inf", "-Infinity", "nan", "0.1", "123456789.123456789e-5",
            "2.2250738585072014e-308", "4.9e-324", "17976931348623157e292"]
    p.write_text("".join(f"v {t} 0 1\n" for t in toks))
    v, _ = preprocess.read_obj(str(p))
    want = np.array([float(t) for t in toks])
    assert np.array_equal(v[:, 0].view(np.uint64)[~np.isnan(want)], want.view(np.uint64)[~np.isnan(want)]) and np.isnan(v[8, 0])
    for bad in ("0x10", "+-1", "1e", "--1", "1.2.3", "e5", "+"):
        p.write_text(f"v {bad} 0 1\n")
        with pytest.raises(ValueError):
            preprocess.read_obj(str(p))
    # a vertex no triangle touches keeps its zero sum (Eigen's normalize() leaves a zero vector alone; open3d's (0,0,1)
    # is for NaN only); a triangle index out of range is an error
    n = preprocess.vertex_normals(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], float), np.array([[0, 1, 2]]))
    assert np.array_equal(n, [[0, 0, 1]] * 3 + [[0, 0, 0]])
    n = preprocess.vertex_normals(np.array([[0, 0, 0], [1, 0, 0], [0, 1, np.nan]], float), np.array([[0, 1, 2]]))
    assert np.array_equal(n, [[0, 0, 1]] * 3)                                  # NaN coordinates: the substitution
    with pytest.raises(ValueError):
        preprocess.vertex_normals(np.zeros((3, 3)), np.array([[0, 1, 3]]))
    lab = preprocess.remap_fdi_labels([0, 11, 18, 21, 28, 31, 38, 41, 48], "upper").reshape(-1)
    assert lab.tolist()[:5] == [0, 1, 8, 9, 16]
    assert preprocess.remap_fdi_labels([0, 31, 38, 41, 48], "lower").reshape(-1).tolist() == [0, 1, 8, 9, 16]


def test_native_scan_loader_equals_the_python_path(tmp_path):
    """tgn_scan_open / tgn_scan_take (the GIL-free load of the sharded runner) against load_scan, which follows
    preprocess_data.py:37-52 step by step in numpy: identical bits, on json files in several legal layouts; json shapes the
    strict reader does not take are handed back (None) so that Python's json decides."""
    from toothgroupnetwork_amd import preprocess, synth
    obj = tmp_path / "AB12_lower.obj"
    obj.write_text(synth.obj_text(60, 45, seed=5, style="slashes"))
    n = 60 * 45
    rng = np.random.default_rng(3)
    for jaw in ("lower", "upper"):
        labels = synth.fdi_labels(n, jaw, seed=9)
        labels[:12] = [0, 5, 9, 10, 19, 20, 29, 30, 39, 48, 55, -3]            # values outside the jaw's own range too
        texts = [json.dumps({"id_patient": "AB12", "jaw": jaw, "labels": labels, "instances": rng.integers(0, 9, n).tolist()}),
                 json.dumps({"labels": labels, "nested": {"labels": [1.5, "x", {"jaw": "no"}], "s": "a\\\"]}"}, "f": -1.5e-3,
                             "t": True, "z": None, "jaw": jaw}, indent=2),
                 '{"jaw":"%s","labels":[%s]}' % (jaw, ",".join(map(str, labels))),
                 '\n {\t"labels" : [ %s ] ,\r\n "jaw"\n:\n"%s" , "labels2": [] }\n' % (" , ".join(map(str, labels)), jaw)]
        for text in texts:
            jp = tmp_path / "gt.json"
            jp.write_text(text)
            want, name, wjaw = preprocess.load_scan(str(obj), str(jp))
            got = preprocess.load_scan_native(str(obj), str(jp), with_xyz32=True)
            assert got is not None, text[:60]
            assert got[1] == name == "AB12_lower" and got[2] == wjaw == jaw and got[3] is None
            assert got[0].dtype == want.dtype and np.array_equal(got[0].view(np.uint64), want.view(np.uint64))
    # more than N_SAMPLED vertices: the float32 copy of the coordinates comes along
    obj.write_text(synth.obj_text(250, 100, seed=6))
    labels = synth.fdi_labels(25000, "upper", seed=1)
    (tmp_path / "gt.json").write_text(json.dumps({"jaw": "upper", "labels": labels}))
    want = preprocess.load_scan(str(obj), str(tmp_path / "gt.json"))[0]
    got = preprocess.load_scan_native(str(obj), str(tmp_path / "gt.json"), with_xyz32=True)
    assert np.array_equal(got[0].view(np.uint64), want.view(np.uint64))
    assert got[3].dtype == np.float32 and np.array_equal(got[3], want[:, :3].astype(np.float32))
    assert preprocess.load_scan_native(str(obj), str(tmp_path / "gt.json"))[3] is None
    # not the plain shape -> None (Python's json module decides what happens); the reference's errors stay errors
    jp = tmp_path / "odd.json"
    for text in ('{"jaw": "upper", "labels": [11.0, 12]}', '{"jaw": "upper"}', '{"labels": [1]}', '{"jaw": "up\\u0070er", "labels": [1]}',
                 '{"jaw": "upper", "labels": [1] ', '[1, 2]', '{"jaw": "upper", "labels": [01]}', '{"jaw": "upper", "labels": [1], "x": tru}',
                 '{"jaw": "upper", "labels": [1,]}', '{"jaw": "upper", "labels": [123456789012345678901]}', ''):
        jp.write_text(text)
        assert preprocess.load_scan_native(str(obj), str(jp)) is None, text
    jp.write_text(json.dumps({"jaw": "upper", "labels": labels[:-1]}))
    with pytest.raises(ValueError):
        preprocess.load_scan_native(str(obj), str(jp))                           # np.concatenate raises in the reference
    with pytest.raises(ValueError):
        preprocess.load_scan(str(obj), str(jp))
    with pytest.raises(ValueError):
        preprocess.load_scan_native(str(tmp_path / "missing.obj"), str(tmp_path / "gt.json"))
    with pytest.raises(ValueError):
        preprocess.load_scan_native(str(obj), str(tmp_path / "missing.json"))


def _oracle_fps_batch(xyz_list, npoint):
    from oracle import cpu as O
    return [O.furthestsampling(np.ascontiguousarray(x, dtype=np.float32), [x.shape[0]], [npoint]).reshape(-1) for x in xyz_list]


def _check_outputs(save, golden_r2):
    for name, jaw, nu, nv, seed, style in SCANS:
        raw = open(os.path.join(save, f"{name}_{jaw}_sampled_points.npy"), "rb").read()
        assert np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), golden_r2[f"pre_{name}_sha256"]), name


def test_pipeline_with_oracle_fps_writes_the_reference_files(tmp_path, golden_r2):
    """host logic only (CPU): reader + normals + remap + scaling + save with the ORACLE's FPS plugged in reproduce the
    files the reference's preprocess_data.py wrote, byte for byte."""
    from toothgroupnetwork_amd import preprocess
    write_dataset(str(tmp_path))
    pairs = preprocess.list_scans(str(tmp_path / "obj"), str(tmp_path / "json"))
    assert len(pairs) == 2
    st = preprocess.preprocess_scans(pairs, str(tmp_path / "out"), batch=2, fps_batch=_oracle_fps_batch)
    assert st["scans"] == 2 and st["sampled"] == 1 and st["points_in"] == 27000 + 9000
    _check_outputs(str(tmp_path / "out"), golden_r2)


def test_sharded_runner_two_ranks_gloo(tmp_path, golden_r2):
    """tools/preprocess_sharded.py's main() under torch.distributed.run with 2 gloo ranks; there is no GPU here, so the
    test-side launcher (tests/sharded_launcher.py) injects the oracle's FPS as the sampler: each rank writes its shard,
    one all_gather combines the counters, the files are the reference's."""
    write_dataset(str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(REPO, "tests", "sharded_launcher.py"), "--source_obj_data_path",
           str(tmp_path / "obj"), "--source_json_data_path", str(tmp_path / "json"), "--save_data_path", str(tmp_path / "out"),
           "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scans"] == 2 and res["per_rank_scans"] == [1, 1] and res["sampled"] == 1
    _check_outputs(str(tmp_path / "out"), golden_r2)


def test_sharded_runner_starts_its_own_ranks_and_refuses_a_mismatch(tmp_path, golden_r2):
    """`--gpus 2` without torchrun: the runner replaces itself by two ranks (toothgroupnetwork_amd.launch) and says who they were;
    under torchrun with another rank count it exits non-zero instead of reporting a smaller job as the one asked for."""
    write_dataset(str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    args = [os.path.join(REPO, "tests", "sharded_launcher.py"), "--source_obj_data_path", str(tmp_path / "obj"), "--source_json_data_path",
            str(tmp_path / "json"), "--save_data_path", str(tmp_path / "out"), "--backend", "gloo"]
    out = subprocess.run([sys.executable] + args + ["--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["self_spawned"] is True and [r["rank"] for r in res["ranks"]] == [0, 1] and res["backend"] == "gloo"
    _check_outputs(str(tmp_path / "out"), golden_r2)
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29633"] + args + ["--gpus", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert bad.returncode != 0 and "refusing to report" in bad.stderr


@pytest.mark.gpu
def test_gpu_pipeline_writes_the_reference_files(dev, tmp_path, golden_r2, oracle):
    """the product path end to end on the GPU (batched FPS kernel): byte-identical .npy files; and the label transfer
    (inference_pipeline_sem.py:37-39) against a brute-force nearest neighbour."""
    from toothgroupnetwork_amd import preprocess, sharding
    write_dataset(str(tmp_path))
    pairs = preprocess.list_scans(str(tmp_path / "obj"), str(tmp_path / "json"))
    res = preprocess.preprocess_sharded(pairs, str(tmp_path / "out"), 0, 1, batch=4)
    assert res["scans"] == 2 and res["sampled"] == 1 and res["per_rank_scans"] == [2]
    _check_outputs(str(tmp_path / "out"), golden_r2)
    arr = np.load(os.path.join(str(tmp_path / "out"), "SYNTHA_upper_upper_sampled_points.npy"))
    assert arr.shape == (24000, 7) and np.array_equal(arr[:8], golden_r2["pre_SYNTHA_upper_head"])
    full, _, _ = preprocess.load_scan(*pairs[0])
    got = preprocess.transfer_labels(arr[:, :3], arr[:, 6], full[:, :3])
    from oracle import cpu as O
    idx_of_samples = O.furthestsampling(np.ascontiguousarray(full[:, :3], dtype=np.float32), [full.shape[0]], [24000]).astype(np.int64)
    sub = np.arange(0, full.shape[0], 37)
    d = ((full[sub, None, :3].astype(np.float32) - arr[None, :, :3].astype(np.float32)) ** 2).sum(-1)
    assert np.array_equal(got[sub], arr[d.argmin(1), 6])
    assert np.array_equal(got[idx_of_samples], arr[:, 6])      # a vertex that IS a sample gets its own label back


def test_array_pool_recycles_row_buffers():
    """preprocess.ArrayPool: take() hands out a view of a kept buffer of enough capacity (or a new one rounded up to 16 384 rows),
    give() takes the base back whatever view it is handed, and at most `keep` buffers are retained."""
    import numpy as np

    from toothgroupnetwork_amd import preprocess
    pool = preprocess.ArrayPool(7, np.float64, keep=2)
    a = pool.take(100000)
    assert a.shape == (100000, 7) and a.dtype == np.float64 and a.base.shape[0] == 114688     # 7 x 16 384 rows of capacity
    base = a.base
    pool.give(a)
    b = pool.take(90000)                                                                         # fits the kept buffer: recycled
    assert b.base is base and b.shape == (90000, 7)
    c = pool.take(120000)                                                                        # too large for anything kept: new
    assert c.base is not base and c.base.shape[0] == 131072
    pool.give(b[:10])                                                                            # a view of a view: still the base
    pool.give(c)
    pool.give(np.empty((5, 7)))                                                                  # beyond `keep`: dropped
    assert len(pool.free) == 2 and pool.free[0] is base
