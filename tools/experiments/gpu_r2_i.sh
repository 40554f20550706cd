#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_parity.py tests/test_gpu_fps_prefix.py tests/test_gpu_random_sweep.py -m gpu -q -x -k "ball or hotpath or pipelined or full_size or sample_and_group" 2>&1 | tail -4
for v in "--ball-stream 1" "--ball-stream 0" "--ball-stream 1 --group-impl 1,2,2 --group-max-blocks 0,512,512" "--ball-stream 1 --group-max-blocks 1024,512,512" "--ball-stream 1 --group-impl 2,7,7 --group-max-blocks 512,128,128" "--ball-stream 1 --pipeline 0"; do
  echo "== bench $v"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))"
done
