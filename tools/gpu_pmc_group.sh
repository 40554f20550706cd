#!/bin/bash
# L2 / TLB counters for the grouping kernels (tools/group_bench.py), a few counters per pass (kernel-trace only).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcg_$i -o pmc -- \
      python $GRAFT_REPO_ROOT/tools/group_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmcg_$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcg_*/pmc_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "group" not in k: continue
        agg[(k, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    print(k, "grid", g)
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
PY
