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
run
run --group-max-blocks 192,192,192
run --group-max-blocks 128,192,192
run --group-max-blocks 256,128,128
run --group-policy 2,2,2
run --group-policy 0,0,0
