#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step'); print(round(d['value']), round(d['ms_per_step'],3), k, round(d['roofline_group']['frac'],3))
except Exception as e: print('FAILED', e)"
}
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_fps_prefix.py tests/test_gpu_sa_fused.py -m gpu -q -x -k "hotpath or pipelined or full_size" 2>&1 | tail -2
run
run --shape B
run --group-delay-us 0 --shape B
run --fused 1
