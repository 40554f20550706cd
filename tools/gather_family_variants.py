#!/usr/bin/env python3
"""secondary_bench.gather_family under each "gather_v4" variant (0: dword lanes + atomics, 3: 16-byte lanes, 5: default)."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from toothgroupnetwork_amd import _lib  # noqa: E402

spec = importlib.util.spec_from_file_location("secondary_bench", os.path.join(REPO, "tools", "secondary_bench.py"))
S = importlib.util.module_from_spec(spec)
spec.loader.exec_module(S)
dev = torch.device("cuda", 0)
variants = [int(v) for v in (sys.argv[1:] or ["0", "3", "5"])]
res = {}
for v in variants:
    with _lib.tuning(gather_v4=v):
        res[v] = S.gather_family(dev)
ops = [k for k, val in res[variants[0]].items() if isinstance(val, dict)]
print(f"{'us per launch (frac of 8 TB/s)':32s}" + "".join(f"{'gather_v4=' + str(v):>22s}" for v in variants))
for op in ops:
    print(f"{op:32s}" + "".join(f"{res[v][op]['us']:13.1f} ({res[v][op]['roofline']['frac']:.3f})" for v in variants))
