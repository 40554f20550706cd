#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/group_bench.py rows 2>&1 | tee gpurun_out/group_bench_rows2.log | tail -18
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcg4 -o pmc -- \
      python $GRAFT_REPO_ROOT/tools/group_bench.py rows > $GRAFT_REPO_ROOT/gpurun_out/pmcg4.log 2>&1)
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcg4/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "group_points" not in k: continue
        agg[(k, row["Grid_Size"], row.get("LDS_Block_Size", row.get("LDS_Block_Size_v", "")))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g, l), cs in sorted(agg.items()):
    print(k, "grid", g, "lds", l, " ".join(f"{c}={sum(v)/len(v)/1e6:.2f}M(n={len(v)})" for c, v in sorted(cs.items())))
PY
