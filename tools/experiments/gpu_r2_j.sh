#!/bin/bash
# schedule experiments: ball queries on their own stream, fenced off from the next step's FPS level 1 ("phased")
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 12 --warmup 4 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))
except Exception as e: print('FAILED', e)"
}
run
run --ball-stream 2
run --ball-stream 2 --group-gate 1
run --ball-stream 1
run --ball-stream 2 --group-impl 2,7,7 --group-max-blocks 512,128,128
run --ball-stream 2 --group-impl 2,7,7 --group-max-blocks 512,256,256
run --ball-stream 2 --group-gate 1 --group-impl 2,7,7 --group-max-blocks 512,128,128
run --batch 512
run --batch 512 --ball-stream 2
run --batch 1024 --ball-stream 2
