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
timeout 600 python -m pytest tests/test_gpu_train_step.py -m gpu -q -x 2>&1 | tail -15
echo "== train step bench"; timeout 600 python tools/train_step_bench.py 2>&1 | tail -3
run --ball-split 0 --group-policy 16,16,2
run --ball-split 0 --group-policy 2,16,2
run --ball-split 0 --group-policy 2,2,2
