#!/bin/bash
# round 3 evidence on the final build: all GPU tests, headline bench + rocprofv3 kernel stats of the same command, Shape B,
# fused lines, PMC traffic passes, extra benches (PT forward, train step, preprocess runner incl. two gloo ranks on one GPU)
set -u
mkdir -p gpurun_out/r3z
export TMPDIR=/tmp
O=gpurun_out/r3z
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tee $O/pytest.log | tail -4
echo "== bench"; timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3z/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["kernel_ms_per_step"], d["roofline_group"]["frac"], d.get("cpu_baseline",{}).get("value"))
PY
echo "== rocprof of the bench command"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --cpu-meshes 0 --no-alt > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1); tail -c 300 $O/rocprof.log | head -3; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -14
echo "== bench shape B"; timeout 600 python bench.py --shape B --steps 20 --warmup 3 --cpu-meshes 0 --no-alt > $O/bench_shapeB.json 2>/dev/null; tail -c 300 $O/bench_shapeB.json
echo "== PMC passes"; bash tools/gpu_pmc.sh 2>&1 | tail -30
python tools/pmc_summary.py gpurun_out $O r03 | tail -12
echo "== preprocess runner"; timeout 600 python tools/preprocess_sharded.py --synthetic 64 --save_data_path /tmp/pre_out 2>&1 | grep -v amdgpu | tail -1 | tee $O/preprocess.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/preprocess_sharded.py --synthetic 64 --save_data_path /tmp/pre_out2 --backend gloo 2>&1 | grep "^{" | tail -1 | tee -a $O/preprocess.txt
