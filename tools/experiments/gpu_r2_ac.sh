#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d.get('kernel_ms_per_step'), 'group frac', round(d.get('roofline_group',{}).get('frac',0),3))
except Exception as e: print('FAILED', e)"
}
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fps_prefix.py tests/test_gpu_sa_fused.py -m gpu -q -x -k "hotpath or pipelined or full_size" 2>&1 | tail -2
run
run --grid-stream 1
run --ball-split 0
run --no-kernel-timing
