#!/bin/bash
# One gpurun call: smoke, GPU tests, FPS sweep, bench + rocprof kernel stats.  Logs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== fps sweep"; timeout 300 python tools/fps_sweep.py --batch 256 > gpurun_out/fps_sweep.log 2>&1; tail -40 gpurun_out/fps_sweep.log
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1; tail -3 gpurun_out/bench.log
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-meshes 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1); tail -2 gpurun_out/rocprof.log; find gpurun_out/prof -name "*stats*" | head
