#!/bin/bash
set -u
mkdir -p gpurun_out/r3x
export TMPDIR=/tmp
O=gpurun_out/r3x
timeout 900 python -m pytest tests/test_gpu_pt_attention.py tests/test_gpu_train_step.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python tools/train_step_bench.py --graph --two-stage 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['ms_per_step'],2) if isinstance(v,dict) else v) for k,v in d.items() if k!='workload'})" | tee $O/train_step.txt
python tools/experiments/train_gemm_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | head -12
