#!/bin/bash
# new defaults: phased schedule, pairs / row-piece grouping beside the LDS-light FPS, early level-1 grid
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 12 --warmup 4 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))
except Exception as e: print('FAILED', e)"
}
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
run
run --early-grid 0
run --group-gate 0
run --ball-stream 0 --group-gate 0
run --group-max-blocks 256,320,320
run --group-policy 2,16,16
echo "== fps sweep cb4"; timeout 300 python tools/fps_sweep.py --batch 256 --bucket 2>&1 | tail -12
echo "== fps sweep cb5"; TGN_FPS_CELL_BITS=5 timeout 300 python tools/fps_sweep.py --batch 256 --bucket 2>&1 | tail -12
