#!/bin/bash
# round 2, call C: correctness of all grouping kernels, sweep, counters of the ring kernel, pipelined bench variants
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (group + seams first)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group or index or tree_tie" 2>&1 | tail -15
echo "== pytest gpu all"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25
echo "== group bench"; timeout 600 python tools/group_bench.py > gpurun_out/group_bench.log 2>&1; cat gpurun_out/group_bench.log
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcg3_$i -o pmc -- \
      python $GRAFT_REPO_ROOT/tools/group_bench.py quick > $GRAFT_REPO_ROOT/gpurun_out/pmcg3_$i.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcg3_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "group_points" not in k: continue
        agg[(k, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    print(k, "grid", g, " ".join(f"{c}={sum(v)/len(v)/1e6:.2f}M(n={len(v)})" for c, v in sorted(cs.items())))
PY
for v in "--group-impl 3 --group-policy 16" "--group-impl 3 --group-policy 16 --group-max-blocks 256" "--group-impl 3 --group-policy 16 --group-max-blocks 1024" "--group-impl 3 --group-policy 0" "--group-impl 3 --group-policy 2" "--group-impl 4 --group-policy 16"; do
  echo "== bench $v"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))"
done
echo "== bench one stream"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt --pipeline 0 --group-max-blocks 1024 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
