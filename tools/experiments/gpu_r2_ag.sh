#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step'); print(round(d['value']), round(d['ms_per_step'],3), 'fps', k['fps_l1'], 'g', k['group_l1'], k['group_l2'], k['group_l3'])
except Exception as e: print('FAILED', e)"
}
for f in 2176; do
  echo "== image floats $f"; TGN_GROUP_IMAGE_FLOATS=$f run
  TGN_GROUP_IMAGE_FLOATS=$f timeout 200 python tools/group_bench.py rows 2>&1 | grep -E "^ +7 +16 +(0|256) "
done
