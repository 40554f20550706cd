#!/bin/bash
# round 3: HIP-graph fix (kNN redo counter cleared by a kernel): regression test, PT forward as a graph, train-step profile
set -u
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
O=gpurun_out/r3g
echo "== pytest pt_attention + parity knn"; timeout 600 python -m pytest tests/test_gpu_pt_attention.py tests/test_gpu_parity.py -m gpu -q -x -k "graph or knn or pt or unet or transition" 2>&1 | tail -4
echo "== pt_forward_bench"; timeout 600 python tools/pt_forward_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/pt_forward.txt | tail -8
echo "== train step profile"; timeout 600 python tools/train_step_bench.py --steps 6 --profile 2>&1 | grep -v amdgpu.ids | tee $O/train_step.txt | tail -45
