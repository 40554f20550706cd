#!/usr/bin/env python3
"""One forward of a network as the ordered list of its kernels, from a rocprofv3 --kernel-trace CSV of a run that repeats
the same forward R times (tools/pnpp_forward_run.py: 3 warm-ups + REPS): the dispatches of the LAST forward, in start order,
with duration, grid and workgroup size, and the gap to the previous kernel's end.

    python tools/forward_sequence.py trace.csv FORWARDS"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
forwards = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // forwards
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
print(f"# {len(rows)} dispatches / {forwards} forwards = {per} per forward; wall of the last forward "
      f"{(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, sum of kernel times {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e3:.1f} us")
print("# start_us   dur_us   gap_us  grid x wg            kernel")
prev = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    for a, b in (("void ", ""), ("at::native::", ""), ("(anonymous namespace)::", "")):
        name = name.replace(a, b)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:8.1f}  {r['Grid_Size_X']:>9s} x {r['Workgroup_Size_X']:<5s}  {name[:120]}")
    prev = e
