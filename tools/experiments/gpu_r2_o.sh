#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d.get('kernel_ms_per_step'), 'group frac', round(d.get('roofline_group',{}).get('frac',0),3))
except Exception as e: print('FAILED', e)"
}
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "hotpath or pipelined" 2>&1 | tail -2
run --no-kernel-timing
run --timing-stride 1
run --timing-stride 4
run --ball-split 0 --no-kernel-timing
run --ball-split 0 --early-grid 0 --no-kernel-timing
run --batch 512 --no-kernel-timing
