#!/bin/bash
# round 3, call 2: the chained two-layer set-abstraction kernel: parity tests, sa_bench, PointNet++ forward bench
set -u
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
O=gpurun_out/r3b
echo "== pytest sa_fused + whole nets"; timeout 900 python -m pytest tests/test_gpu_sa_fused.py tests/test_gpu_whole_nets.py tests/test_gpu_fps_prefix.py -m gpu -q -x -s 2>&1 | tee $O/pytest.log | grep -v "^\s*$" | tail -25
echo "== sa_bench"; timeout 600 python tools/sa_bench.py 2>&1 | tee $O/sa_bench.txt | tail -20
echo "== pointnet++ forward"; timeout 600 python tools/pointnetpp_forward_bench.py 2>&1 | tee $O/pnpp_forward.txt | tail -12
