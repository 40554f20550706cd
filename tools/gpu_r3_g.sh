#!/bin/bash
# round 3: training step: static losses, whole-step HIP graph
set -u
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
O=gpurun_out/r3g
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_train_step.py -m gpu -q -x 2>&1 | tail -15
echo "== train step"; timeout 900 python tools/train_step_bench.py --steps 8 --graph 2>&1 | grep -v amdgpu.ids | tee $O/train_step3.txt | tail -5
