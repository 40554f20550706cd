#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step'); print(round(d['value']), round(d['ms_per_step'],3), k)
except Exception as e: print('FAILED', e)"
}
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x -k "hotpath or pipelined" 2>&1 | tail -2
run
run --group-order 2,1,0
run --group-order 1,2,0
run --group-order 1,0,2
