#!/bin/bash
# One gpurun call: smoke, GPU tests, microbenchmarks, bench + rocprof kernel stats.  Logs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT=${1:-all}
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
if [ "$WHAT" = "all" ] || [ "$WHAT" = "micro" ]; then
  echo "== fps bucket sweep"; timeout 300 python tools/fps_sweep.py --batch 256 --bucket > gpurun_out/fps_bucket_sweep.log 2>&1; cat gpurun_out/fps_bucket_sweep.log
fi
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-meshes 0 --no-alt > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1); tail -1 gpurun_out/rocprof.log; find gpurun_out/prof -name "*.csv" | head; cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -20
