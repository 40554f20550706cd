#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 30 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); k=d.get('kernel_ms_per_step'); print(round(d['value']), round(d['ms_per_step'],3), k['fps_l1'], k['group_l1'], k['group_l2'], k['group_l3'], k['ball_l1'])
except Exception as e: print('FAILED', e)"
}
for rep in 1 2; do
echo "== default"; run
echo "== policy 2,2,2"; run --group-policy 2,2,2
echo "== policy 2,16,2"; run --group-policy 2,16,2
echo "== delay 100"; run --group-delay-us 100
echo "== delay 220"; run --group-delay-us 220
done
