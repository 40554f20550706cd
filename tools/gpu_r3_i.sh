#!/bin/bash
# round 3: ball-grid build unrolled: parity, bench (+ rocprof stats for the build kernel)
set -u
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
O=gpurun_out/r3i
echo "== parity"; timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_random_sweep.py -m gpu -q -x -k "ball or query or sample or pipelined or shape_b or hotpath" > $O/parity.log 2>&1; rc=$?; tail -3 $O/parity.log
if [ $rc -ne 0 ]; then echo "PARITY FAILED"; exit 0; fi
for r in 1 2; do timeout 300 python bench.py --steps 40 --warmup 5 --cpu-meshes 0 --no-alt > $O/bench_$r.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench_$r.json").read().strip().splitlines()[-1])
print({k:round(d[k],3) for k in ("value","ms_per_step")}, d["kernel_ms_per_step"])
PY
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt > /dev/null 2>&1); f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i "build\|Name" $f | cut -c1-140
