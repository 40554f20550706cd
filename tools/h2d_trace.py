#!/usr/bin/env python3
"""The H2D-fed headline step alone (tools/secondary_bench.py: hot_path_with_h2d), for a rocprofv3 --kernel-trace --memory-copy-trace run."""
import importlib.util
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402

spec = importlib.util.spec_from_file_location("secondary_bench", os.path.join(REPO, "tools", "secondary_bench.py"))
sb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sb)
which = sys.argv[1] if len(sys.argv) > 1 else "h2d"
fn = sb.hot_path_with_h2d if which == "h2d" else sb.hot_path_tree_ties
print(json.dumps(fn(bench.make_inputs, torch.device("cuda", 0))))
