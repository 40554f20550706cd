#!/bin/bash
# round-6 counter passes (rocprofv3 --pmc, a few counters per pass, --kernel-trace only):
#   HBM traffic of the hot-path kernels (FETCH_SIZE / WRITE_SIZE), the instruction mix of the ball queries per level, of the kNN grid kernel,
#   and the measured vector-issue ceiling (tools/valu_bench)
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_pmc
mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --pipeline 0 --group-max-blocks 256 --steps 2 --warmup 1 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1)
  tail -1 gpurun_out/pmc_$C.log | cut -c1-120
done
mkdir -p $O/prof
python tools/pmc_summary.py gpurun_out $O/prof r06 2>&1 | tail -12
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/sq_$i -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-meshes 0 --no-alt --no-kernel-timing --pipeline 0 --secondary 0 > $GRAFT_REPO_ROOT/$O/sq_$i.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/knn_$i -o pmc -- \
      python $GRAFT_REPO_ROOT/tools/knn_run.py > $GRAFT_REPO_ROOT/$O/knn_$i.log 2>&1)
done
python - <<'PY' | tee gpurun_out/r6_pmc/sq_summary.txt
import csv, glob, collections, json
def collect(pattern, want):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(pattern, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if not any(w in k for w in want): continue
            agg[k][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return agg
out = {}
a = collect("gpurun_out/r6_pmc/sq_*/**/pmc_counter_collection.csv", ("ball_grid_query", "ball_query_scan", "ball_grid_build", "fps_"))
for k, cs in sorted(a.items()):
    print(k)
    for c, v in sorted(cs.items()):
        v.sort()
        vals = [x[1] for x in v]
        print(f"   {c:24s} per dispatch in launch order: " + " ".join(f"{x:.4g}" for x in vals[:12]))
        out.setdefault(k, {})[c] = vals
a = collect("gpurun_out/r6_pmc/knn_*/**/pmc_counter_collection.csv", ("knn",))
for k, cs in sorted(a.items()):
    print(k)
    for c, v in sorted(cs.items()):
        vals = [x[1] for x in sorted(v)]
        print(f"   {c:24s} " + " ".join(f"{x:.4g}" for x in vals[:8]))
        out.setdefault(k, {})[c] = vals
json.dump(out, open("gpurun_out/r6_pmc/sq_counters.json", "w"))
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o tools/valu_bench 2>/dev/null; ./tools/valu_bench > $O/valu_issue_rate.txt 2>&1; grep "waves/SIMD=[48]" $O/valu_issue_rate.txt | head -20
rm -rf $O/sq_*/ $O/knn_*/ 2>/dev/null
