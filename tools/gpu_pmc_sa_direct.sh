#!/bin/bash
# SQ counters of the direct-form chained set-abstraction kernel (tools/sa_direct_time.py), a few per pass, kernel-trace only.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_sad
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sad/p$i -o pmc -- \
      TGN_SA_TIME_ONLY=bf16x3 python $GRAFT_REPO_ROOT/tools/sa_direct_time.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_sad/p$i.log 2>&1)
  tail -1 gpurun_out/pmc_sad/p$i.log
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_sad/p*/**/pmc_counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "sa_mlp2_max_split" not in k: continue
        agg[(k[:80], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    print(k, "grid", g)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
PY
