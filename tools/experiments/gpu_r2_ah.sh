#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 16 --warmup 4 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step'); print(round(d['value']), round(d['ms_per_step'],3), k)
except Exception as e: print('FAILED', e)"
}
run
run --batch 512
run --batch 512 --group-max-blocks 256,256,256
run --batch 128
run --batch 192
