#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 30 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d.get('kernel_ms_per_step'), 'group frac', round(d.get('roofline_group',{}).get('frac',0),3))
except Exception as e: print('FAILED', e)"
}
run
TGN_FPS_CONFIG=256,4 run
TGN_FPS_CONFIG=256,8 run
run --group-policy 2,2,2
run --group-policy 2,16,16
run
