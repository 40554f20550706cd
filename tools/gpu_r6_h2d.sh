#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_h2d
mkdir -p $O
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o h -- python $GRAFT_REPO_ROOT/tools/h2d_trace.py h2d > $GRAFT_REPO_ROOT/$O/h2d.log 2>&1)
tail -1 $O/h2d.log | cut -c1-600
python - <<'PY'
import csv, glob
k = glob.glob("gpurun_out/r6_h2d/tr/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/r6_h2d/tr/**/*memory_copy_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(k)):
    n = r["Kernel_Name"]
    short = ("fps_l1" if "fps_bucket_kernel<512, 48" in n else "fps_l23" if "fps_" in n else "grid" if "grid_build" in n else "query" if "ball_" in n
             else "group" if "group_points" in n else "spacer" if "delay" in n else n[:30])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, "s" + r.get("Stream_Id", "?")))
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?"))[:24], "-"))
rows.sort()
t0 = rows[0][0]
sel = [x for x in rows if x[2] == "fps_l1"]
print("fps_l1 launches:", len(sel), "copy rows:", sum(1 for x in rows if x[2].startswith("COPY")), "copy csv:", m)
lo, hi = sel[9][0], sel[11][1]
for s, e, n, st in rows:
    if lo - 200000 <= s <= hi:
        print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:8.3f} {n:28s} {st}")
PY
rm -rf $O/tr
