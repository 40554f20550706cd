#!/bin/bash
# LDS-light FPS level 1 (12-bit cell codes: 63 KiB instead of 116) so that the row-piece grouping kernel fits 4 waves per CU beside it
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
export TGN_FPS_CELL_BITS=4
echo "== parity with 12-bit cells"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fps_prefix.py -m gpu -q -x -k "fps" 2>&1 | tail -3
run --pipeline 0
run
run --group-impl 2,7,7 --group-max-blocks 512,256,256
run --group-impl 2,7,7 --group-max-blocks 512,320,320
run --group-impl 2,7,7 --group-max-blocks 512,256,256 --group-gate 1
run --group-impl 2,7,7 --group-max-blocks 512,256,256 --ball-stream 2
run --group-impl 2,7,7 --group-max-blocks 512,256,256 --ball-stream 2 --group-gate 1
run --group-impl 2,7,7 --group-max-blocks 256,256,256 --group-gate 1
export TGN_FPS_CELL_BITS=5
run --pipeline 0
