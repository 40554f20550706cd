"""CPU: libtgn_pointops.so loads and exports every function declared in include/*.h, the ctypes
signature table covers them, and the product path refuses CPU tensors (no fallback).  No compute."""
import ctypes
import glob
import os
import re

import pytest
import torch

from conftest import REPO


def _declared_functions():
    names = []
    for h in glob.glob(os.path.join(REPO, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)  # preprocessor lines
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src):
            if m.group(1) not in ("defined",):
                names.append(m.group(1))
    return sorted(set(names))


def test_header_declares_the_reference_launchers():
    names = _declared_functions()
    for ref in ["furthestsampling_cuda_launcher", "knnquery_cuda_launcher", "grouping_forward_cuda_launcher",
                "grouping_backward_cuda_launcher", "interpolation_forward_cuda_launcher",
                "interpolation_backward_cuda_launcher", "subtraction_forward_cuda_launcher",
                "subtraction_backward_cuda_launcher", "aggregation_forward_cuda_launcher",
                "aggregation_backward_cuda_launcher"]:
        assert ref in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    from toothgroupnetwork_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(handle, name), f"{name} declared in include/ but not exported"


def _declared_arity():
    out = {}
    for h in glob.glob(os.path.join(REPO, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{}()]*)\)\s*;", src):
            args = m.group(2).strip()
            out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_ctypes_table_matches_header():
    from toothgroupnetwork_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_functions()
    arity = _declared_arity()
    for name, (_, args) in _lib.SIGNATURES.items():     # a wrong count is a TypeError (or garbage arguments) at call time
        assert len(args) == arity[name], (name, len(args), arity[name])
    L = _lib.lib()
    assert b"gfx950" in L.tgn_version()
    assert L.tgn_fps_resident_capacity() >= 24000  # a whole 24 000-point scan stays in registers
    assert L.tgn_ball_query_workspace_bytes(1, 24000, 4096) >= 0


def test_ops_refuse_cpu_tensors_loudly():
    from toothgroupnetwork_amd import pointnet2_utils as U, pointops as P
    xyz = torch.rand(1, 64, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        U.farthest_point_sample(xyz, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        U.query_ball_point(0.2, 4, xyz, xyz[:, :4])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.furthestsampling(xyz[0], torch.tensor([64], dtype=torch.int32), torch.tensor([8], dtype=torch.int32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.knnquery(4, xyz[0], xyz[0], torch.tensor([64], dtype=torch.int32), torch.tensor([64], dtype=torch.int32))


def test_missing_library_fails_loudly(monkeypatch):
    from toothgroupnetwork_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtgn_pointops.so")
    with pytest.raises(_lib.TgnLibraryError, match="not been built"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    bad = []
    for root in ("toothgroupnetwork_amd", "external_libs", "tools"):    # shipped runners and benches included
        for dirpath, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith(".py"):
                    txt = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    txt = open(os.path.join(REPO, "pointops_cuda.py")).read()
    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M)
    assert bad == []
