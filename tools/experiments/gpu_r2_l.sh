#!/bin/bash
# pairs kernel (group level 1): parity, standalone rate, pipelined bench with the LDS-light FPS
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
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_sa_fused.py -m gpu -q -x -k "group or hotpath or pipelined or full_size or sample_and_group or modules or sweep" 2>&1 | tail -4
echo "== group_bench pairs"; timeout 300 python tools/group_bench.py pairs 2>&1 | tail -20
export TGN_FPS_CELL_BITS=4
run --group-impl 10,7,7 --group-max-blocks 256,256,256
run --group-impl 10,7,7 --group-max-blocks 512,256,256
run --group-impl 10,7,7 --group-max-blocks 256,256,256 --ball-stream 2
run --group-impl 10,7,7 --group-max-blocks 256,256,256 --group-gate 1
run --group-impl 10,7,7 --group-max-blocks 256,256,256 --group-gate 1 --ball-stream 2
run --group-impl 10,7,7 --group-max-blocks 256,256,256 --pipeline 0
