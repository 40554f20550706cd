#!/usr/bin/env python3
"""Summarise the two PMC passes of tools/gpu_pmc.sh into profiles/<tag>_pmc_traffic.json (read by bench.py for
roofline.traffic) and copy the raw counter CSVs next to it.

    python tools/pmc_summary.py [gpurun_out] [profiles]

Counter values are the TCC derived counters FETCH_SIZE / WRITE_SIZE in KiB (x 1024 -> bytes), one pass per
counter (MI355X_MICROARCH.md, HBM section).  Dispatches are mapped to hot-path kernels by launch order inside
a step: the i-th FPS / ball-query / group dispatch of a step is level i+1.
"""
import collections
import csv
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
tag = sys.argv[3] if len(sys.argv) > 3 else "r05"
LEVELS = 3


def read(counter):
    path = os.path.join(src, f"pmc_{counter}", "pmc_counter_collection.csv")
    rows = []
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") != counter:
            continue
        rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0], float(row["Counter_Value"]) * 1024.0))
    rows.sort()
    shutil.copy(path, os.path.join(dst, f"{tag}_pmc_{counter}_counter_collection.csv"))
    return rows


def classify(name):
    if "fps_" in name:
        return "fps"
    if "ball_grid_query" in name or "ball_query_scan" in name:
        return "ball"
    if "group_points" in name:
        return "group"
    return None


out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py "
                 "--pipeline 0 --group-max-blocks 256 --steps 2 --warmup 1 --cpu-meshes 0 --no-kernel-timing (one stream: the dispatch order identifies the level; the groupings at the "
                 "256-block grid they have in the phased schedule -- rounds 1-4 measured them at the unbounded grid, where 32 scans per XCD thrash its L2: group_l1 fetched "
                 "977 MB there and 176 MB here); 256 scans per launch; bytes = counter x 1024, "
                 "NOT doubled (MI355X_MICROARCH.md notes FETCH_SIZE under-reports wide streaming reads by 2x on gfx950; "
                 "these kernels read 4 B per lane).  Mean over the dispatches of each kernel class and level.",
       "per_kernel": {}}
acc = collections.defaultdict(lambda: {"fetch": [], "write": []})
for counter, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    seen = collections.Counter()
    for _, name, val in read(counter):
        pk = out["per_kernel"].setdefault(name, {"dispatches": 0, "fetch_bytes_per_dispatch": [], "write_bytes_per_dispatch": []})
        pk[f"{key}_bytes_per_dispatch"].append(val)
        kind = classify(name)
        if kind:
            level = seen[kind] % LEVELS + 1
            seen[kind] += 1
            acc[f"{kind}_l{level}"][key].append(val)
for name, pk in out["per_kernel"].items():
    pk["dispatches"] = max(len(pk["fetch_bytes_per_dispatch"]), len(pk["write_bytes_per_dispatch"]))
for k, v in sorted(acc.items()):
    out[k] = {"fetch": sum(v["fetch"]) / max(len(v["fetch"]), 1), "write": sum(v["write"]) / max(len(v["write"]), 1)}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
for k in sorted(acc):
    print(f"{k:10s} fetch {out[k]['fetch'] / 1e6:10.1f} MB  write {out[k]['write'] / 1e6:10.1f} MB per launch")
