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
run
run --group-delay-us 100
run --group-delay-us 200
run --group-delay-us 300
run --group-delay-us 400
