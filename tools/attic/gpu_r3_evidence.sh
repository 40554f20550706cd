#!/bin/bash
# final build: smoke, the driver's bench command (+ 20-step variant), rocprofv3 kernel stats of it
set -u
mkdir -p gpurun_out/evidence
export TMPDIR=/tmp
O=gpurun_out/evidence
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench","bench_20"):
    d=json.loads(open(f"gpurun_out/evidence/{f}.json").read().strip().splitlines()[-1])
    print(f, {k:d[k] for k in ("value","ms_per_step","steps")}, d["config"]["schedule"][-120:], d["roofline"]["frac"], d["roofline_group"]["frac"], d["cpu_baseline"]["value"])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r3 -- python $GRAFT_REPO_ROOT/bench.py --cpu-meshes 0 --no-alt > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1); f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cut -c1-120 $f | head -6
