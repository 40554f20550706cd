#!/bin/bash
# round 3: multi-scale fused levels (out_stride), bench --shape B --fused 1 = the reference net's own SA stack
set -u
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
O=gpurun_out/r3h
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_sa_fused.py tests/test_gpu_whole_nets.py tests/test_gpu_modules.py tests/test_gpu_fps_prefix.py -m gpu -q -x 2>&1 | tail -4
echo "== bench shape B fused"; timeout 600 python bench.py --shape B --fused 1 --steps 6 --warmup 2 --cpu-meshes 0 > $O/bench_shapeB_fused.json 2> $O/bench_shapeB_fused.err; tail -c 1400 $O/bench_shapeB_fused.json; tail -2 $O/bench_shapeB_fused.err
echo "== pointnet++ forward"; timeout 600 python tools/pointnetpp_forward_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/pnpp_forward.txt | tail -7
