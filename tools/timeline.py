#!/usr/bin/env python3
"""Kernel timeline of the pipelined bench from a rocprofv3 --kernel-trace CSV: start / end / duration (ms) and HIP stream of
every hot-path kernel over the last few steps, plus the steady-state period.

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py --steps 6 --warmup 2 --cpu-meshes 0 --no-alt --no-kernel-timing
    python tools/timeline.py out/t_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    n = r["Kernel_Name"]
    short = ("fps_l1" if "fps_bucket_kernel<512, 48" in n else "fps_l2" if "fps_bucket" in n else "fps_l3" if ("fps_resident" in n or "fps_lean" in n)
             else "grid" if "grid_build" in n else "query" if ("ball_grid_query" in n or "ball_query_scan" in n)
             else "group_l1" if "pairs" in n else "group_l23" if "rows_kernel" in n else "spacer" if "delay" in n
             else "group" if "group_points" in n else None)
    if short:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", "?")))
ks.sort()
t0 = ks[0][0]
sel = [k for k in ks if k[2] == "fps_l1"]
print("# start_ms end_ms duration_ms kernel stream  (last two steps)")
for s, e, n, st in ks:
    if sel[-3][0] - 100000 <= s < sel[-1][0] + 100000:
        print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:7.3f} {n:9s} s{st}")
print("# steady-state periods (ms):", " ".join(f"{(sel[i + 1][0] - sel[i][0]) / 1e6:.3f}" for i in range(2, len(sel) - 1)))
