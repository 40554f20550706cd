#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb_$i -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-meshes 0 --no-alt --no-kernel-timing --pipeline 0 > $GRAFT_REPO_ROOT/gpurun_out/pmcb_$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcb_*/pmc_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "ball" not in k and "pairs" not in k: continue
        agg[(k, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    print(k, "grid", g)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):2d} min={min(v):16.1f} max={max(v):16.1f}")
PY
