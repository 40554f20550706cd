#!/bin/bash
# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE counter passes (no tracing domains besides
# --kernel-trace), as MI355X_MICROARCH.md prescribes.  Summaries land in gpurun_out/pmc_*.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --pipeline 0 --steps 2 --warmup 1 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1)
  tail -1 gpurun_out/pmc_$C.log | cut -c1-200
  ls gpurun_out/pmc_$C
done
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv")
    if not fs:
        print("no counter csv for", c); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(fs[0])):
        if row.get("Counter_Name") != c: continue
        k = row["Kernel_Name"].split("(")[0][:70]
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    print("==", c, "(sum over dispatches; units as reported: KiB on gfx9 TCC derived counters)")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{k:72s} dispatches={n:3d} total={v:14.1f} per_dispatch={v / n:14.1f}")
PY
