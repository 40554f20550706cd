#!/bin/bash
# round 3: all GPU tests after the training-path changes; the two-stage step of BASELINE config 3
set -u
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
O=gpurun_out/r3g
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
echo "== train step, two stages"; timeout 900 python tools/train_step_bench.py --steps 6 --two-stage 2>&1 | grep -v amdgpu.ids | tee $O/train_step4.txt | tail -3
