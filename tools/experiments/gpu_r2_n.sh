#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_n -o t -- python $R/bench.py --steps 6 --warmup 2 --cpu-meshes 0 --no-alt --no-kernel-timing > $R/gpurun_out/rocprof_n.log 2>&1); tail -1 gpurun_out/rocprof_n.log | cut -c1-200
ls gpurun_out/prof_n
run() {
  echo "== bench $*"
  timeout 300 python bench.py --steps 12 --warmup 4 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))
except Exception as e: print('FAILED', e)"
}
run --batch 512
run --batch 512 --group-max-blocks 256,256,256
run --no-kernel-timing 2>/dev/null
